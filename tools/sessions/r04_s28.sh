#!/bin/bash
# Round 4, session 28: the 3x3 tables of the batch-2 / batch-4 KITTI shapes (coalesced requests; measured in r04_s15 on a library without F(4x4,3x3))
# and of c3 / configs[4] fp32 once more on the round-4 F(4x4,3x3) kernel.  Logs only - the entries are merged by hand with a 3 % margin.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s28
mkdir -p $OUT
timeout 600 python tools/bench_wino.py --batch 2 --min-pixels 8192 > $OUT/wino_b2.log 2>&1; echo "b2 rc=$?"
timeout 600 python tools/bench_wino.py --batch 4 --min-pixels 8192 > $OUT/wino_b4.log 2>&1; echo "b4 rc=$?"
timeout 600 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --min-pixels 8192 > $OUT/wino_c3.log 2>&1; echo "c3 rc=$?"
timeout 600 python tools/bench_wino.py --height 512 --width 1024 --frames 4 --depths 48 --min-pixels 8192 > $OUT/wino_c5.log 2>&1; echo "c5 rc=$?"
tail -1 $OUT/wino_b2.log $OUT/wino_b4.log $OUT/wino_c3.log $OUT/wino_c5.log | cut -c1-200
