#!/bin/bash
# Round 4, session 25: conv3x3_wino44_kernel "ping-pong" - the two waves of a SIMD one phase apart (load + transform | 36 MFMAs from registers),
# a barrier per phase - against the lock-step loop (diagnostic library, MR_W44_DBG=100 = the other structure): parity cases, c3 / c2 layer times.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s25
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd44 or wino44 or cooktoom or winograd_1d" > $OUT/k.log 2>&1; echo "kernel tests rc=$?"; tail -2 $OUT/k.log | cut -c1-300
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for dbg in 0 100; do
  echo "== MR_W44_DBG=$dbg (0: ping-pong, 100: lock step)"
  for shape in "--batch 8 --frames 4 --depths 64" ""; do
    MR_W44_DBG=$dbg timeout 300 python tools/bench_wino.py $shape --min-pixels 30000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{') and 'wino31_us' in l:
        r = json.loads(l); print('  ', r['name'], r['hw'], 'n', r['n'], 'cin', r['cin'], 'cout', r['cout'], 'wino44', r['wino31_us'], 'maxdiff', round(r['wino31_maxdiff'], 7), ' F(2x2) best', min(r[k] for k in r if k.startswith('wino') and k.endswith('_us') and k != 'wino31_us'))
"
  done
done
