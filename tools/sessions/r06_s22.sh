#!/bin/bash
# Round 6, session 22 (= r06_s17 on the FINAL tree): round-5 tree (ab_r5/: `git archive 57d0845`, its own library) against this tree on ONE box, interleaved: c2 (20 / 200 steps), c3, configs[4] bf16,
# and the kernel-only sum of a c2 keyframe by HIP events (device_ms_per_step_sum_of_kernels).  Box-to-box spread this round was +-3 %: only same-box pairs count.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06_s22
{
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
Q="--no-primer --no-cpu-baseline --no-forward-api --no-secondary"
for rep in 1 2 3; do
  (cd ab_r5 && timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "r5 tree c2 200:")
  timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "r6 tree c2 200:"
  (cd ab_r5 && timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line "r5 tree c2 20:")
  timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line "r6 tree c2 20:"
done
C3="--batch 8 --frames 4 --depths 64 --steps 30"
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16 --steps 100"
for rep in 1 2; do
  (cd ab_r5 && timeout 400 python bench.py $C3 $Q 2>/dev/null | line "r5 tree c3:")
  timeout 400 python bench.py $C3 $Q 2>/dev/null | line "r6 tree c3:"
  (cd ab_r5 && timeout 400 python bench.py $C5 $Q 2>/dev/null | line "r5 tree c5 bf16:")
  timeout 400 python bench.py $C5 $Q 2>/dev/null | line "r6 tree c5 bf16:"
done
} | tee gpurun_out/r06_s22/ab.txt
# the issue-rate probe again, with the bf16 + VALU placements (kinds 22-25)
timeout 200 tools/probes/mfma_rates > gpurun_out/r06_s22/mfma_rates.txt; grep -n 'bf16' gpurun_out/r06_s22/mfma_rates.txt | cut -c1-200
