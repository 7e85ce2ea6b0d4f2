#!/bin/bash
# Round 6, session 31: direct-kernel schedules of c2 chosen by the PIPELINED rate (tools/tune_pipeline.py: the runners-up of an isolated sweep, kept where the
# four-in-flight loop itself gets faster), then the installed table against the result through bench.py.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s31
mkdir -p $OUT
DIRECT=resnet.,mask.enc2,mask.enc3,mask.enc4,mask.dec0,mask.dec1,depth.enc2.0.conv_x,depth.enc3,depth.enc4,depth.dec0,depth.dec1.0
cp monorec_amd/tuned_schedules.json $OUT/isolated.json
timeout 600 python tools/tune_conv.py --merge --only $DIRECT --out $OUT/isolated.json --report $OUT/tune_report.json > $OUT/tune.log 2>&1; echo "tune_conv rc=$?"; tail -1 $OUT/tune.log
timeout 2400 python tools/tune_pipeline.py --report $OUT/tune_report.json --out $OUT/pipelined.json --budget-seconds 1500 > $OUT/tune_pipeline.log 2>&1; echo "tune_pipeline rc=$?"
grep -c candidate $OUT/tune_pipeline.log; grep '"kept": true' $OUT/tune_pipeline.log | cut -c1-300; tail -1 $OUT/tune_pipeline.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
Q="--no-primer --no-cpu-baseline --no-forward-api --no-secondary"
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "installed table, 200:"
  MR_TUNED_SCHEDULES=$OUT/pipelined.json timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "pipeline-tuned, 200:"
  timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line "installed table, 20:"
  MR_TUNED_SCHEDULES=$OUT/pipelined.json timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line "pipeline-tuned, 20:"
done
