#!/bin/bash
# Round 6, session 1: FIRST hardware run of (a) the software-pipelined fp32 sweep (register double buffer: the LDS reads of k-step group g + 1
# are requested before the MFMAs of group g) and (b) the LDS ring pipeline (mr_conv_desc.pipeline_buffers, ABI 19: all chunks of a workgroup
# requested up front, partial vmcnt waits).  Parity first, then the c2 line on the round-5 table (sweep only), then a re-tune of the direct-kernel
# layers with ring candidates and the A/B of the two tables.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv and not cost_volume" -p no:cacheprovider > $OUT/conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -5 $OUT/conv_tests.log | cut -c1-300
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_old_table.json 2>$OUT/bench1.err | tee $OUT/bench_old_table.json | line "r5 table, new sweep, 200 steps:"
DIRECT=resnet.,mask.enc2,mask.enc3,mask.enc4,mask.dec0,mask.dec1,depth.enc2.0.conv_x,depth.enc3,depth.enc4,depth.dec0,depth.dec1.0
cp monorec_amd/tuned_schedules.json $OUT/tuned_ring.json
timeout 1500 python tools/tune_conv.py --merge --only $DIRECT --ring 3,4,8 --out $OUT/tuned_ring.json --report $OUT/tune_report.json > $OUT/tune.log 2>&1; echo "tune rc=$?"; tail -50 $OUT/tune.log | cut -c1-260
for rep in 1 2; do
  timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "r5 table:"
  MR_TUNED_SCHEDULES=$OUT/tuned_ring.json timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_ring_table.json 2>/dev/null | line "ring table:"
done
