#!/bin/bash
# Round 4, session 23: where the time of conv3x3_wino44_kernel goes at the c3 shape - ablations on the diagnostic library (MR_W44_DBG bits:
# 1 no input transform, 2 no MFMAs, 4 no patch reads, 8 no A reads, 16 no DMA).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s23
mkdir -p $OUT
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for dbg in 0 1 2 4 8 16 5 13 29; do
  MR_W44_DBG=$dbg timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{') and 'wino31_us' in l:
        r = json.loads(l); print('dbg $dbg: wino44', r['wino31_us'], 'us   (F(2x2) best', min(r[k] for k in r if k.startswith('wino') and k.endswith('_us') and k != 'wino31_us'), ')')
"
done
