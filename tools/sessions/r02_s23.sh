#!/bin/bash
# First hardware run of the Winograd F(2x2,3x3) kernel: difference to the direct MFMA kernel and times, every 3x3 stride-1 layer of c2, the mask net of c3.
OUT=gpurun_out/s23
mkdir -p $OUT
timeout 300 python tools/bench_wino.py > $OUT/wino_c2.log 2>&1; echo "c2 rc=$?"; cat $OUT/wino_c2.log | cut -c1-330
timeout 300 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask > $OUT/wino_c3.log 2>&1; echo "c3 rc=$?"; cat $OUT/wino_c3.log | cut -c1-330
