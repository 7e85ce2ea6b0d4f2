#!/bin/bash
# Round 6, session 7: conv_b8_kernel sweep ablation (diagnostic library): MR_B8_DBG 1 no sweep, 2 no input staging, 4 no weight DMA, 8 no stores,
# 16 no A (weight) LDS reads, 32 no B (input) LDS reads.
cd "$(dirname "$0")/../.." || exit 1
export MR_HIP_LIBRARY=$(pwd)/monorec_amd/libmonorec_hip_timeline.so
mkdir -p gpurun_out/r06_s7
for L in "enc0.1 3,4,8" "enc0.1 3,2,4" "dec3.1 3,4,8"; do
  set -- $L
  for DBG in 0 1 10 26 42 58 11 14 62; do
    MR_B8_DBG=$DBG timeout 120 python tools/bench_b8.py --layer $1 --scheds $2 2>/dev/null | grep sched
  done
done | tee gpurun_out/r06_s7/b8_sweep_ablation.txt
