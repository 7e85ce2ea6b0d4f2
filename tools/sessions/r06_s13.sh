#!/bin/bash
# Round 6, session 13: whole gpu suite + smoke on the current library (ABI 19), c2 / c3 lines.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s13
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -25 $OUT/suite.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "c2 200 steps:"
timeout 400 python bench.py --batch 8 --frames 4 --depths 64 --steps 30 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "c3:"
