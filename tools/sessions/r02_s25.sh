#!/bin/bash
# Ping-pong Winograd kernel (two wave groups alternating sweep / transform): parity, then A/B against the lock-step kernel.
OUT=gpurun_out/s25
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "winograd kernel tests rc=$?"; tail -3 $OUT/pytest_wino.log
echo "--- ping-pong"; timeout 200 python tools/bench_wino.py --only mask 2>/dev/null | grep -E "enc0.0|enc1.1|dec2.1|dec3.1|dec3.2|layers" | cut -c1-300
echo "--- lock-step"; MR_WINO_V1=1 timeout 200 python tools/bench_wino.py --only mask 2>/dev/null | grep -E "enc0.0|enc1.1|dec2.1|dec3.1|dec3.2|layers" | cut -c1-300
echo "--- c3 ping-pong"; timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask 2>/dev/null | grep -E "enc0.0|enc2.1|dec1.1|dec2.1|dec3.1|layers" | cut -c1-300
