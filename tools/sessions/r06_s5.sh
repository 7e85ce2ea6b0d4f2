#!/bin/bash
# Round 6, session 5: where does conv_b8_kernel (bf16 path, BASELINE configs[4]) spend its time?  MR_B8_DBG ablations on the diagnostic library:
# 1 no sweep, 2 no input staging, 4 no weight DMA, 8 no stores.
cd "$(dirname "$0")/../.." || exit 1
export MR_HIP_LIBRARY=$(pwd)/monorec_amd/libmonorec_hip_timeline.so
mkdir -p gpurun_out/r06_s5
for L in "enc0.1 3,2,4" "dec3.1 3,4,8" "dec3 3,2,8" "enc0.0 3,2,4"; do
  set -- $L
  for DBG in 0 1 2 8 9 3 10 11; do
    MR_B8_DBG=$DBG timeout 120 python tools/bench_b8.py --layer $1 --scheds $2 2>/dev/null | grep sched
  done
done | tee gpurun_out/r06_s5/b8_ablation.txt
