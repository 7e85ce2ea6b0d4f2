#!/bin/bash
# Round-2 first hardware session: parity of everything, marching cost volume vs the tiled kernels, driver-style bench.
OUT=gpurun_out/s1
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for SHAPE in "1 256 512 2 32" "8 256 512 4 64" "1 512 1024 4 48"; do
  set -- $SHAPE
  timeout 120 python tools/bench_cv.py --batch $1 --height $2 --width $3 --frames $4 --depths $5 --iters 50 2>&1 | tail -1
done
for TY in 16 24 32 37 48 64; do
  MR_CV_MARCH_TY=$TY timeout 60 python tools/bench_cv.py --impl march --iters 100 2>&1 | tail -1
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 1500 $OUT/bench_driver.json | cut -c1-900
timeout 300 python bench.py --steps 300 --no-cpu-baseline --dump-layers $OUT/layers.json > $OUT/bench300.json 2> $OUT/bench300.err; cut -c1-420 $OUT/bench300.json
for IF in 1 3 4; do
  timeout 200 python bench.py --steps 300 --no-cpu-baseline --no-primer --in-flight $IF 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_flight', d['config']['keyframes_in_flight'], round(d['value'],1), 'kf/s')"
done
export MR_TEST_EXPERIMENTAL=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3" > $OUT/bf16x3_kernels.log 2>&1; echo "bf16x3 kernel tests rc=$?"; tail -3 $OUT/bf16x3_kernels.log
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "bf16x3" -s > $OUT/bf16x3_model.log 2>&1; echo "bf16x3 model test rc=$?"; tail -3 $OUT/bf16x3_model.log
timeout 200 python bench.py --steps 200 --no-cpu-baseline --no-primer --bf16x3 2>/dev/null | cut -c1-300
