#!/bin/bash
# profiles of the round-3 state: c2 and c3 (kernel traces: pipelined and one-keyframe-single-stream; six PMC passes each)
OUT=gpurun_out/r03_s20; mkdir -p $OUT
bash tools/profile_round.sh r03_c2 > $OUT/prof_c2.log 2>&1; tail -3 $OUT/prof_c2.log | cut -c1-200
bash tools/profile_round.sh r03_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12 > $OUT/prof_c3.log 2>&1; tail -3 $OUT/prof_c3.log | cut -c1-200
