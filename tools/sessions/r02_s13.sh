#!/bin/bash
# bf16 MFMA mode on the K = 32 instruction: kernel + model tests, bench lines at c2 and the configs[4] shape.
OUT=gpurun_out/s13
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -k "bf16" > $OUT/pytest.log 2>&1; echo "bf16 tests rc=$?"; tail -3 $OUT/pytest.log; grep "bf16 mode\|c5 shape" $OUT/pytest.log | head -4
timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer --bf16 > $OUT/c2--bf16.json 2>/dev/null; cut -c1-200 $OUT/c2--bf16.json
timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-primer --height 512 --width 1024 --frames 4 --depths 48 --bf16 > $OUT/c5--bf16.json 2>/dev/null; cut -c1-200 $OUT/c5--bf16.json
