timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd" 2>&1 | grep -E "^E|FAILED|Error|assert" | head -12
