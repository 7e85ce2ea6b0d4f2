#!/bin/bash
# Round 4, session 34: conv3x3_wino44_kernel with the DMA instructions of the next chunk spread over the MFMAs of the first channel quad (two per
# transform column) instead of one burst behind the chunk barrier: parity cases, c3 / c2 layer times against the burst (diagnostic library, MR_W44_DBG=32).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s34
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd44 or wino44" > $OUT/k.log 2>&1; echo "kernel tests rc=$?"; tail -2 $OUT/k.log | cut -c1-300
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for dbg in 0 32 0 32; do
  echo "== MR_W44_DBG=$dbg (0: spread, 32: burst)"
  for shape in "--batch 8 --frames 4 --depths 64" ""; do
    MR_W44_DBG=$dbg timeout 300 python tools/bench_wino.py $shape --min-pixels 30000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{') and 'wino31_us' in l:
        r = json.loads(l); print('  ', r['name'], r['hw'], 'n', r['n'], 'cin', r['cin'], 'cout', r['cout'], 'wino44', r['wino31_us'], 'maxdiff', round(r['wino31_maxdiff'], 7))
"
  done
done
