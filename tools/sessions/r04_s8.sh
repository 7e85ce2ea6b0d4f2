#!/bin/bash
# Round 4, session 8: schedules of the B8 kernel on the full-resolution layers (two workgroups per CU vs one).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s8
mkdir -p $OUT
{
timeout 100 python tools/bench_b8.py --layer enc0.1 --scheds 3,4,8 3,2,8 3,1,8 3,2,4 3,4,4 3,1,4
timeout 100 python tools/bench_b8.py --layer enc0.0 --scheds 3,4,8 3,2,8 3,2,4
timeout 100 python tools/bench_b8.py --layer dec3.1 --scheds 3,4,8 3,2,8 3,2,4 3,1,8
timeout 100 python tools/bench_b8.py --layer dec3 --height 256 --width 512 --scheds 3,2,8 3,4,8 3,2,4 3,1,8
timeout 100 python tools/bench_b8.py --layer dec2.1 --height 256 --width 512 --scheds 4,2,8 4,2,4 4,1,8 2,2,8 2,4,8
timeout 100 python tools/bench_b8.py --layer enc1.0 --height 256 --width 512 --scheds 3,4,8 3,2,8 3,2,4
timeout 100 python tools/bench_b8.py --layer enc0.1x --scheds 3,2,8 3,2,4 3,1,8 3,1,4
} 2>&1 | grep sched | tee $OUT/scheds.log
