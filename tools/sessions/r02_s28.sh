#!/bin/bash
# Winograd table for the remaining shapes: every 3x3 stride-1 layer of c3 and of the configs[4] shape (direct / 32 / 64 couts per workgroup).
OUT=gpurun_out/s28
mkdir -p $OUT
timeout 400 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --emit $OUT/wino_c3.json > $OUT/wino_c3.log 2>&1; tail -1 $OUT/wino_c3.log
timeout 400 python tools/bench_wino.py --height 512 --width 1024 --frames 4 --depths 48 --emit $OUT/wino_c5.json > $OUT/wino_c5.log 2>&1; tail -1 $OUT/wino_c5.log
timeout 400 python tools/bench_wino.py --emit $OUT/wino_c2.json > $OUT/wino_c2.log 2>&1; tail -1 $OUT/wino_c2.log
