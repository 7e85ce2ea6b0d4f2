#!/bin/bash
# Round 6, session 18: which of this round's conv_b8_kernel changes cost the configs[4] line 3 %?  Variant libraries (ab_b8/): v0 = round-5 kernel, va = + input ring,
# vb = + peeled tap loop, product = + bias in the accumulators + buffer-descriptor stores / permlane swap epilogue.  One box, interleaved.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
for rep in 1 2; do
for L in "enc0.1 3,2,4 3,4,8" "dec3.1 3,4,8" "enc1.0 3,2,4"; do
  set -- $L; layer=$1; shift
  for v in v0 product; do
    if [ $v = product ]; then unset MR_HIP_LIBRARY; else export MR_HIP_LIBRARY=$REPO/ab_b8/libmonorec_hip_$v.so; fi
    timeout 200 python tools/bench_b8.py --layer $layer --scheds "$@" 2>/dev/null | grep sched | sed "s/^/$v /"
  done
done
done
unset MR_HIP_LIBRARY
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
Q="--no-primer --no-cpu-baseline --no-forward-api --no-secondary"
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16 --steps 100"
for v in v0 product v0 product; do
  if [ $v = product ]; then unset MR_HIP_LIBRARY; else export MR_HIP_LIBRARY=$REPO/ab_b8/libmonorec_hip_$v.so; fi
  timeout 400 python bench.py $C5 $Q 2>/dev/null | line "c5 bf16 $v:"
done
