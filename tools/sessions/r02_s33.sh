#!/bin/bash
# Last check of the round: the whole gpu suite and smoke() with the final library (ABI 7).
OUT=gpurun_out/s33
mkdir -p $OUT
S=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
