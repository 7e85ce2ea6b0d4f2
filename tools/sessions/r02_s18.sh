#!/bin/bash
# Per-head timing of mr_depth_heads_f32 (quad / pixel mode, weights from LDS or global) at the c2 and c3 decoder sizes.
t() { echo "$1: $(env $2 timeout 120 python tools/bench_heads.py $3 2>/dev/null | tail -1)"; }
t "c2 default" "A=1" ""
t "c2 pixel mode, global weights" "MR_HEADS_W_LDS=0" ""
t "c2 all quad" "MR_HEADS_QUAD_MIN=1" ""
t "c2 all pixel" "MR_HEADS_QUAD_MIN=1000000000" ""
t "c2 quad from 16k pixels" "MR_HEADS_QUAD_MIN=16384" ""
t "c3 default" "A=1" "--batch 8 --depths 64"
t "c3 all quad" "MR_HEADS_QUAD_MIN=1" "--batch 8 --depths 64"
t "c5 shape default" "A=1" "--height 512 --width 1024 --depths 48"
