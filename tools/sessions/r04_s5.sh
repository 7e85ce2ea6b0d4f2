#!/bin/bash
# Round 4, session 5: where the time of a full-resolution B8 layer goes - ablations of conv_b8_kernel on the diagnostic library.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s5
mkdir -p $OUT
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for L in enc0.1 enc0.0 enc0.1x; do
  for DBG in 0 1 2 4 8 3 9 11 15; do
    MR_B8_DBG=$DBG timeout 60 python tools/bench_b8.py --layer $L --scheds 3,4,8 3,2,8 2>&1 | grep "sched"
  done
done | tee $OUT/ablate.log
