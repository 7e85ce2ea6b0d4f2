#!/bin/bash
# First hardware run of the F(2x2,2x2) transposed-convolution kernel: parity, then times next to the direct kernel (c2, c3 Refine layers).
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "transposed" 2>&1 | tail -3
timeout 200 python tools/bench_wino_t.py 2>&1 | tail -6 | cut -c1-330
timeout 200 python tools/bench_wino_t.py --batch 8 --frames 4 --depths 64 2>&1 | tail -6 | cut -c1-330
