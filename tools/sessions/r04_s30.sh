#!/bin/bash
# Round 4, session 30: the bf16 mode with depth counts off the fast paths (7, 20 hypotheses) and the other bf16 model / kernel tests on the final library.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r04_s30
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_b8.py -x -q -k "bf16 or b8" > gpurun_out/r04_s30/t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04_s30/t.log | cut -c1-300
