#!/bin/bash
# final profiles (kernel-name groups now include the Upconv kernel) and the bench lines that quote them
OUT=gpurun_out/r03_s23; mkdir -p $OUT
bash tools/profile_round.sh r03_c2 > $OUT/prof_c2.log 2>&1
bash tools/profile_round.sh r03_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12 > $OUT/prof_c3.log 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-160 $OUT/bench_driver.json
timeout 400 python bench.py --steps 200 --no-primer > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-160 $OUT/bench_c2.json
timeout 300 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer > $OUT/bench_c3.json 2>/dev/null; cut -c1-160 $OUT/bench_c3.json
cp profiles/r03_* $OUT/
