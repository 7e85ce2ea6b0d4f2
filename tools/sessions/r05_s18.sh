#!/bin/bash
# Round 5, session 18: measured tables for the reference's second evaluation shape (configs/evaluate/eval_monorec_oxrc.json:26: batch 4; the Oxford RobotCar loader
# feeds 320x640, data_loader/oxford_robotcar_dataset.py:53) - VERDICT r4 "missing" #6: direct-kernel schedules, 3x3 forms, transposed, 1-D forms, stride-2 pairs for
# batch 4 and batch 1 at 320x640, into COPIES of the tables; then tree tables against candidate tables on that shape, with the depth against the CPU oracle.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s18
mkdir -p $OUT
cp monorec_amd/tuned_schedules.json $OUT/tuned_schedules.json
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd.json
S="--height 320 --width 640"
for k in 4 1; do
  timeout 600 python tools/tune_conv.py --batch $k $S --out $OUT/tuned_schedules.json --merge --missing --report $OUT/tune_b$k.json > $OUT/tune_b$k.log 2>&1; echo "tune batch $k rc=$?"; tail -1 $OUT/tune_b$k.log
  export MR_TUNED_SCHEDULES=$OUT/tuned_schedules.json
  timeout 600 python tools/bench_wino.py --batch $k $S --emit $OUT/tuned_winograd.json > $OUT/wino_b$k.log 2>&1; echo "wino batch $k rc=$?"; tail -1 $OUT/wino_b$k.log | cut -c1-200
  timeout 600 python tools/bench_wino_t.py --batch $k $S --emit $OUT/tuned_winograd.json > $OUT/wino_t_b$k.log 2>&1; echo "wino_t batch $k rc=$?"; tail -1 $OUT/wino_t_b$k.log | cut -c1-200
  timeout 600 python tools/bench_wino1d.py --batch $k $S --emit $OUT/tuned_winograd.json > $OUT/wino1d_b$k.log 2>&1; echo "wino1d batch $k rc=$?"; tail -1 $OUT/wino1d_b$k.log | cut -c1-200
  timeout 600 python tools/bench_stride2.py --batch $k $S --emit $OUT/tuned_winograd.json > $OUT/stride2_b$k.log 2>&1; echo "stride2 batch $k rc=$?"; tail -1 $OUT/stride2_b$k.log | cut -c1-200
  unset MR_TUNED_SCHEDULES
done
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth vs cpu', d.get('depth_max_abs_err_vs_cpu'))"; }
for rep in 1 2; do
  for k in 4 1; do
    timeout 300 python bench.py --steps 100 --batch $k $S --no-primer --no-forward-api --no-cpu-baseline 2>/dev/null | line "320x640 batch $k, tables of the tree"
    MR_TUNED_SCHEDULES=$OUT/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/tuned_winograd.json timeout 300 python bench.py --steps 100 --batch $k $S --no-primer --no-forward-api --no-cpu-baseline 2>/dev/null | line "320x640 batch $k, candidate tables"
  done
done
MR_TUNED_SCHEDULES=$OUT/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/tuned_winograd.json timeout 400 python bench.py --steps 40 --batch 4 $S --no-primer --no-forward-api 2>/dev/null | line "320x640 batch 4, candidate tables, with the CPU oracle"
MR_TUNED_SCHEDULES=$OUT/tuned_schedules.json MR_TUNED_WINOGRAD=$OUT/tuned_winograd.json timeout 400 python bench.py --steps 40 --batch 1 $S --no-primer --no-forward-api 2>/dev/null | line "320x640 batch 1, candidate tables, with the CPU oracle"
python - <<'PY'
import json
for f in ("tuned_schedules", "tuned_winograd"):
    old = json.load(open(f"monorec_amd/{f}.json")); new = json.load(open(f"gpurun_out/r05_s18/{f}.json"))
    changed = [k for k in old if new.get(k) != old[k]]
    print(f, "entries", len(old), "->", len(new), "changed old entries:", changed[:5])
PY
