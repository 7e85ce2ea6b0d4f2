#!/bin/bash
# Round 6, session 28: B8 schedule tables (bf16 mode) for the two KITTI shapes of BASELINE configs[1] / configs[2] - tuned_b8.json held configs[4] signatures only, so
# `hip_bf16=True` at 256x512 ran on the rule.  tools/tune_all.py --bf16, then the bf16 lines on the rule against the new entries.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s28
mkdir -p $OUT
timeout 1500 python tools/tune_all.py --shape 1 256 512 2 32 --bf16 --out-dir $OUT/tables 2>&1 | tee $OUT/tune_all_c2.txt | cut -c1-250
timeout 1500 python tools/tune_all.py --shape 8 256 512 4 64 --bf16 --out-dir $OUT/tables 2>&1 | tee $OUT/tune_all_c3.txt | cut -c1-250
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
Q="--no-primer --no-forward-api --no-secondary --no-cpu-baseline"
for rep in 1 2; do
  timeout 400 python bench.py --bf16 --steps 200 $Q 2>/dev/null | line "c2 bf16 rule:"
  MR_TUNED_B8=$OUT/tables/tuned_b8.json timeout 400 python bench.py --bf16 --steps 200 $Q 2>/dev/null | line "c2 bf16 table:"
  timeout 400 python bench.py --bf16 --batch 8 --frames 4 --depths 64 --steps 30 $Q 2>/dev/null | line "c3 bf16 rule:"
  MR_TUNED_B8=$OUT/tables/tuned_b8.json timeout 400 python bench.py --bf16 --batch 8 --frames 4 --depths 64 --steps 30 $Q 2>/dev/null | line "c3 bf16 table:"
done
MR_TUNED_B8=$OUT/tables/tuned_b8.json timeout 400 python bench.py --bf16 --steps 100 --no-primer --no-forward-api --no-secondary 2>/dev/null | line "c2 bf16 table, with the CPU oracle:"
