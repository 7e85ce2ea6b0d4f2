#!/bin/bash
# Round 6, session 19: (a) does the LDS ring bookkeeping of the DIRECT fp32 kernel cost anything at two buffers?  ab_b8/libmonorec_hip_oldloop.so = this tree's
# conv_mfma.hip with the round-5 two-buffer chunk loop put back (new sweep kept; ring depths of the table are ignored by it).  (b) tuned_b8.json re-measured on the
# final conv_b8 kernel, A/B against the installed table.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s19
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
Q="--no-primer --no-cpu-baseline --no-forward-api --no-secondary"
for v in oldloop product oldloop product; do
  if [ $v = product ]; then unset MR_HIP_LIBRARY; else export MR_HIP_LIBRARY=$REPO/ab_b8/libmonorec_hip_$v.so; fi
  timeout 400 python bench.py --steps 100 $Q --dump-layers $OUT/layers_c2_$v.json 2>/dev/null | line "c2 $v:"
done
unset MR_HIP_LIBRARY
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16"
cp monorec_amd/tuned_b8.json $OUT/tuned_b8_new.json
timeout 1500 python tools/tune_b8.py --emit $OUT/tuned_b8_new.json > $OUT/tune_b8.log 2>&1; echo "tune_b8 rc=$?"; tail -1 $OUT/tune_b8.log | cut -c1-300
for rep in 1 2; do
  timeout 400 python bench.py $C5 --steps 100 $Q 2>/dev/null | line "c5 bf16, installed table:"
  MR_TUNED_B8=$OUT/tuned_b8_new.json timeout 400 python bench.py $C5 --steps 100 $Q 2>/dev/null | line "c5 bf16, new table:"
done
