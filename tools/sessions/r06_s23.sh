#!/bin/bash
# Round 6, session 23: tools/tune_all.py end to end on a shape no table has seen - 480x640, 2 source frames, 32 depth bins, batch 1 (TUM RGB-D class,
# data_loader/tum_rgbd_dataset.py) - then the line on the nearest-signature rules (committed tables) against the line on the tables the tool wrote.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s23
mkdir -p $OUT
timeout 3000 python tools/tune_all.py --shape 1 480 640 2 32 --out-dir $OUT/tables 2>&1 | tee $OUT/tune_all.txt | cut -c1-250
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
Q="--no-primer --no-forward-api --no-secondary"
S="--height 480 --width 640 --frames 2 --depths 32 --steps 100"
for rep in 1 2; do
  timeout 400 python bench.py $S $Q --no-cpu-baseline 2>/dev/null | line "480x640 rules:"
  MR_TUNED_SCHEDULES=$OUT/tables/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/tables/tuned_winograd.json timeout 400 python bench.py $S $Q --no-cpu-baseline 2>/dev/null | line "480x640 tables of tune_all:"
done
MR_TUNED_SCHEDULES=$OUT/tables/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/tables/tuned_winograd.json timeout 600 python bench.py $S $Q 2>/dev/null | line "480x640 tables of tune_all, with the CPU oracle:"
