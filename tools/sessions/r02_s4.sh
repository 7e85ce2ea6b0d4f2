#!/bin/bash
OUT=gpurun_out/s4
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log; grep "kitti example\|fused volume\|masked fused" $OUT/pytest.log | head
for SHAPE in "1 256 512 2 32" "8 256 512 4 64" "1 512 1024 4 48"; do
  set -- $SHAPE
  timeout 120 python tools/bench_cv.py --batch $1 --height $2 --width $3 --frames $4 --depths $5 --iters 50 2>&1 | tail -1
  MR_CV_NO_KF_PREPASS=1 timeout 120 python tools/bench_cv.py --impl march --batch $1 --height $2 --width $3 --frames $4 --depths $5 --iters 50 2>&1 | tail -1
done
for TY in 16 24 30 37 43; do
  MR_CV_MARCH_TY=$TY timeout 60 python tools/bench_cv.py --impl march --iters 100 2>&1 | tail -1
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-330 $OUT/bench_driver.json
timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer 2>/dev/null | cut -c1-300
