#!/bin/bash
# Round 6, session 8: conv_b8_kernel with the tap loop peeled (no branch around the operand reads: the LDS waits now cover the PREVIOUS tap's reads)
# and the bias in the accumulator initialisation.  Parity, ablation of the new sweep, re-tune with ring depths, configs[4] lines.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s8
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_b8.py -m gpu -q -x -p no:cacheprovider > $OUT/b8_tests.log 2>&1; echo "b8 tests rc=$?"; tail -4 $OUT/b8_tests.log | cut -c1-300
for L in "enc0.1 3,4,8" "dec3.1 3,4,8"; do
  set -- $L
  for DBG in 0 1 10 11 62; do
    MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so MR_B8_DBG=$DBG timeout 120 python tools/bench_b8.py --layer $1 --scheds $2 2>/dev/null | grep sched
  done
done | tee $OUT/b8_sweep_ablation.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16"
timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_c5_old_table.json 2>/dev/null | line "c5 bf16, round-5 table:"
cp monorec_amd/tuned_b8.json $OUT/tuned_b8_new.json
timeout 1500 python tools/tune_b8.py --stages 0,3,4 --emit $OUT/tuned_b8_new.json > $OUT/tune_b8.log 2>&1; echo "tune_b8 rc=$?"; tail -1 $OUT/tune_b8.log | cut -c1-300
for rep in 1 2; do
  timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "c5 bf16, round-5 table:"
  MR_TUNED_B8=$OUT/tuned_b8_new.json timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_c5_new.json 2>/dev/null | line "c5 bf16, new table:"
done
MR_TUNED_B8=$OUT/tuned_b8_new.json timeout 400 python bench.py $C5 --steps 20 --warmup 5 --no-primer 2>/dev/null | line "c5 bf16, new table, driver-style with cpu check:"
