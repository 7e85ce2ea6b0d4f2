#!/bin/bash
# Ping-pong Winograd kernel, U issued by the transforming group only.
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd" 2>&1 | tail -1
timeout 200 python tools/bench_wino.py --only mask.enc 2>/dev/null | grep -E "enc0.0|enc1.1" | cut -c1-260
timeout 200 python tools/bench_wino.py --only mask.dec3 2>/dev/null | grep -E "dec3" | cut -c1-260
timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 2>/dev/null | grep -E "enc0.0" | cut -c1-260
