#!/bin/bash
# Last session of round 2: the multi-rank branch of bench.py with two ranks on one device (gloo), then the committed bench lines of the
# final code (they quote the committed rocprofv3 tables for the kernel-only fractions, so they are taken after those were committed).
OUT=gpurun_out/s30
mkdir -p $OUT
MR_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "2-rank rc=$?"; tail -1 $OUT/bench_2rank.json | cut -c1-300
timeout 400 python bench.py --steps 200 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -1 $OUT/bench_c2.json | cut -c1-200
timeout 300 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline > $OUT/bench_c3.json 2>/dev/null; tail -1 $OUT/bench_c3.json | cut -c1-200
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>/dev/null; tail -1 $OUT/bench_driver.json | cut -c1-200
