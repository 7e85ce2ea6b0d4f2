#!/bin/bash
# Round 6, session 14: (1) hip_skip_dead_layer4 parity + line; (2) the untuned path: 320x640 batch 4 / batch 1 with the committed tables vs
# with every entry of that shape REMOVED (tools/holdout_tables.py: nearest-signature rules); (3) a TUM-class shape no table has seen, 480x640 F2 D32;
# (4) c3: direct-kernel schedules re-measured on the new sweep with ring candidates, A/B.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s14
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "skip_dead_layer4" -p no:cacheprovider 2>&1 | tail -2
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
Q="--no-primer --no-cpu-baseline --no-forward-api --no-secondary"
timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "c2 200 steps:"
timeout 300 python bench.py --steps 200 $Q --skip-layer4 2>/dev/null | line "c2 200 steps, hip_skip_dead_layer4:"
timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line "c2 driver-style:"
timeout 300 python bench.py --steps 20 --warmup 5 $Q --skip-layer4 2>/dev/null | line "c2 driver-style, hip_skip_dead_layer4:"
for k in 4 1; do
  python tools/holdout_tables.py --shape $k 320 640 2 32 --out-dir $OUT/holdout_b$k > /dev/null 2>&1
  S="--height 320 --width 640 --batch $k"
  for rep in 1 2; do
    timeout 300 python bench.py --steps 100 $S $Q 2>/dev/null | line "320x640 batch $k, committed tables:"
    MR_TUNED_SCHEDULES=$OUT/holdout_b$k/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/holdout_b$k/tuned_winograd.json timeout 300 python bench.py --steps 100 $S $Q 2>/dev/null | line "320x640 batch $k, entries of the shape removed (nearest-signature rules):"
  done
done
MR_TUNED_SCHEDULES=$OUT/holdout_b4/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/holdout_b4/tuned_winograd.json timeout 400 python bench.py --steps 40 --height 320 --width 640 --batch 4 --no-primer --no-forward-api --no-secondary 2>/dev/null | line "320x640 batch 4, rules, with the CPU oracle:"
timeout 400 python bench.py --steps 100 --height 480 --width 640 --no-primer --no-forward-api --no-secondary 2>/dev/null | line "480x640 F2 D32 (TUM-class, no table entries), with the CPU oracle:"
# (4) c3
cp monorec_amd/tuned_schedules.json $OUT/tuned_c3.json
timeout 1200 python tools/tune_conv.py --batch 8 --frames 4 --depths 64 --merge --ring 3,4 --out $OUT/tuned_c3.json --report $OUT/tune_c3.json > $OUT/tune_c3.log 2>&1; echo "tune c3 rc=$?"; tail -1 $OUT/tune_c3.log
C3="--batch 8 --frames 4 --depths 64 --steps 30"
for rep in 1 2; do
  timeout 400 python bench.py $C3 $Q 2>/dev/null | line "c3, committed table:"
  MR_TUNED_SCHEDULES=$OUT/tuned_c3.json timeout 400 python bench.py $C3 $Q 2>/dev/null | line "c3, re-tuned direct kernel:"
done
