#!/bin/bash
# Round 4, session 37: the 1-D tables (F(2,3) / F(4,3) / F(4,7) / direct) once more after the shorter generated transform chains - logs only.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s37
mkdir -p $OUT
timeout 600 python tools/bench_wino1d.py --no-upconv > $OUT/w1d_c2.log 2>&1; echo "c2 rc=$?"; tail -2 $OUT/w1d_c2.log | cut -c1-300
timeout 600 python tools/bench_wino1d.py --no-upconv --batch 8 --frames 4 --depths 64 > $OUT/w1d_c3.log 2>&1; echo "c3 rc=$?"; tail -2 $OUT/w1d_c3.log | cut -c1-300
