#!/bin/bash
# Round 5, session 21: the handed-over tree once more after the B8 schedule table: whole gpu suite, smoke.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s21
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $OUT/suite.log | tail -2
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
