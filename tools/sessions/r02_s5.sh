#!/bin/bash
OUT=gpurun_out/s5
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "marching or cost_volume" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for SHAPE in "1 256 512 2 32" "8 256 512 4 64"; do
  set -- $SHAPE
  for rep in 1 2; do
    timeout 120 python tools/bench_cv.py --impl march --batch $1 --height $2 --width $3 --frames $4 --depths $5 --iters 50 2>&1 | tail -1
    MR_HIP_LIBRARY=$(pwd)/monorec_amd/libmonorec_hip_nosched.so timeout 120 python tools/bench_cv.py --impl march --batch $1 --height $2 --width $3 --frames $4 --depths $5 --iters 50 2>&1 | tail -1 | sed 's/^/nosched /'
  done
done
timeout 300 python tools/tune_conv.py --bf16 --merge --missing > $OUT/tune_bf16_c2.log 2>&1; tail -2 $OUT/tune_bf16_c2.log
timeout 300 python tools/tune_conv.py --bf16 --merge --missing --height 512 --width 1024 --frames 4 --depths 48 > $OUT/tune_bf16_c5.log 2>&1; tail -2 $OUT/tune_bf16_c5.log
timeout 300 python tools/tune_conv.py --bf16x3 --merge --missing --height 512 --width 1024 --frames 4 --depths 48 > $OUT/tune_bf16x3_c5.log 2>&1; tail -2 $OUT/tune_bf16x3_c5.log
cp monorec_amd/tuned_schedules.json $OUT/tuned_schedules.json
for MODE in "--bf16" "--bf16x3"; do
  timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-primer --height 512 --width 1024 --frames 4 --depths 48 $MODE > $OUT/c5$MODE.json 2>/dev/null; cut -c1-200 $OUT/c5$MODE.json
done
timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer --bf16 > $OUT/c2--bf16.json 2>/dev/null; cut -c1-200 $OUT/c2--bf16.json
