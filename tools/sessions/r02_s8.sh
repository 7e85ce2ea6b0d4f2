#!/bin/bash
# K split across the waves of a workgroup: kernel tests, full gpu suite, re-tune c2 with the new candidates, A/B against the table without them.
OUT=gpurun_out/s8
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" > $OUT/pytest_conv.log 2>&1; echo "conv tests rc=$?"; tail -4 $OUT/pytest_conv.log
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cp monorec_amd/tuned_schedules.json $OUT/table_A.json
b() { timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s', 'sum-of-kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3), 'conv ms', round(d['roofline']['conv_ms_per_step'],3))"; }
b "table A (no K-split-wave entries)"
timeout 900 python tools/tune_conv.py --merge --report $OUT/tune_report.json > $OUT/tune_kws.log 2>&1; tail -3 $OUT/tune_kws.log; grep -c ", 1)" $OUT/tune_kws.log
cp monorec_amd/tuned_schedules.json $OUT/table_K.json
b "table K (K-split-wave candidates allowed)"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-330 $OUT/bench_driver.json
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --dump-layers $OUT/layers.json > /dev/null 2>&1
