#!/bin/bash
# Dynamic batching of a request stream (hip_batch_keyframes): test, driver-style line with both secondary keys, full suite.
OUT=gpurun_out/s12
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "dynamic_batching or owned or in_place" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -3 $OUT/pytest_new.log
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step']); print(d.get('secondary_bf16x3')); print(d.get('secondary_dynamic_batching'))"
tail -3 $OUT/bench_driver.err
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
