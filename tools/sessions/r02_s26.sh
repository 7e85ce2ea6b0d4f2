#!/bin/bash
# Which waves of a 512-thread workgroup share a SIMD?  Ping-pong Winograd kernel with the two candidate pairings.
for p in 0 1; do
echo "--- pairing $p"; MR_WINO_PAIRING=$p timeout 200 python tools/bench_wino.py --only mask.enc 2>/dev/null | grep -E "enc0.0|enc1.1" | cut -c1-260
MR_WINO_PAIRING=$p timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 2>/dev/null | grep -E "enc0.0" | cut -c1-260
done
MR_WINO_PAIRING=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd" 2>&1 | tail -1
