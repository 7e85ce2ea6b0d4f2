#!/bin/bash
# Round 6, session 20: the direct kernel back on the two-buffer chunk loop (ring removed, r06_s19): conv parity, re-tune of the direct-kernel layers of c2 on
# the final kernel, A/B of the tables; forward() with nothing between the input wait and the first launch (MR_DIAG_FORWARD: 1 = round-5 order, 2 = host-side
# wait for the result instead of a wait packet on the caller's stream).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s20
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv and not cost_volume" -p no:cacheprovider > $OUT/conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -3 $OUT/conv_tests.log | cut -c1-300
for rep in 1 2; do
  for D in 1 0 2 3; do MR_DIAG_FORWARD=$D timeout 300 python tools/forward_rate.py 2>/dev/null | tail -1; done
done | tee $OUT/forward_rate.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
Q="--no-primer --no-cpu-baseline --no-forward-api --no-secondary"
DIRECT=resnet.,mask.enc2,mask.enc3,mask.enc4,mask.dec0,mask.dec1,depth.enc2.0.conv_x,depth.enc3,depth.enc4,depth.dec0,depth.dec1.0
cp monorec_amd/tuned_schedules.json $OUT/tuned_new.json
timeout 1500 python tools/tune_conv.py --merge --only $DIRECT --out $OUT/tuned_new.json --report $OUT/tune_report.json > $OUT/tune.log 2>&1; echo "tune rc=$?"; tail -1 $OUT/tune.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "installed table:"
  MR_TUNED_SCHEDULES=$OUT/tuned_new.json timeout 300 python bench.py --steps 200 $Q --dump-layers $OUT/layers_new_table.json 2>/dev/null | line "new table:"
done
