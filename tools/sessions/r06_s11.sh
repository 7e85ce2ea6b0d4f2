#!/bin/bash
# Round 6, session 11: FIRST hardware run of conv3x3_wino44w_kernel (F(4x4,3x3), one wave per SIMD, both cout blocks per wave, quad pipeline in
# registers, ring of four single-quad stages): parity (bit-identical to conv_wino44.hip), then times next to codes 31 / 41 at c2 and c3.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s11
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd44" -p no:cacheprovider > $OUT/w44_tests.log 2>&1; echo "w44 tests rc=$?"; tail -5 $OUT/w44_tests.log | cut -c1-300
timeout 300 python tools/bench_wino.py --codes 31,51 --min-pixels 32768 > $OUT/wino_c2.log 2>&1; echo "wino c2 rc=$?"; grep -o '"name": "[a-z0-9.]*"\|"direct_us": [0-9.]*\|"wino[0-9]*_us": [0-9.]*\|"wino51_maxdiff": [0-9.e-]*' $OUT/wino_c2.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
timeout 600 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --codes 31,51 --min-pixels 32768 > $OUT/wino_c3.log 2>&1; echo "wino c3 rc=$?"; grep -o '"name": "[a-z0-9.]*"\|"direct_us": [0-9.]*\|"wino[0-9]*_us": [0-9.]*\|"wino51_maxdiff": [0-9.e-]*' $OUT/wino_c3.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
