#!/bin/bash
# 16-wave Winograd kernel (transform and sweep overlapped, two waves per SIMD in each role): parity, then times next to the 8-wave kernel.
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "winograd" 2>&1 | tail -3
timeout 200 python tools/bench_wino.py --only mask.enc 2>/dev/null | grep -E "enc0.0|enc1.1|enc2.1" | cut -c1-330
timeout 200 python tools/bench_wino.py --only mask.dec3 2>/dev/null | grep -E "dec3" | cut -c1-330
timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 2>/dev/null | grep -E "enc0.0" | cut -c1-330
