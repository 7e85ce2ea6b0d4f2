#!/bin/bash
# Winograd kernel behind the plan: kernel tests, the model suite (parity bars unchanged), interleaved A/B at c2, c3 and the configs[4] shape.
OUT=gpurun_out/s24
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "winograd kernel tests rc=$?"; tail -3 $OUT/pytest_wino.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_evaluate_loop.py tests/test_pointcloud.py -m gpu -q > $OUT/pytest_model.log 2>&1; echo "model tests rc=$?"; tail -5 $OUT/pytest_model.log
b() { echo "$1: $(env $2 timeout 300 python bench.py --no-cpu-baseline --no-primer $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'kf/s', 'sum-of-kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3), 'conv ms', round(d['roofline']['conv_ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))")"; }
timeout 300 python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1     # primer + warm box
for r in 1 2; do
  b "c2 direct kernel only" "MR_WINOGRAD=0" "--steps 300"
  b "c2 with the Winograd kernel" "A=1" "--steps 300"
done
b "c3 direct kernel only" "MR_WINOGRAD=0" "--steps 40 --batch 8 --frames 4 --depths 64"
b "c3 with the Winograd kernel" "A=1" "--steps 40 --batch 8 --frames 4 --depths 64"
b "c5 shape direct kernel only" "MR_WINOGRAD=0" "--steps 60 --height 512 --width 1024 --frames 4 --depths 48"
b "c5 shape with the Winograd kernel (heuristic, no table entries)" "A=1" "--steps 60 --height 512 --width 1024 --frames 4 --depths 48"
timeout 200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); print('driver-style', round(d['value'],1), 'depth err', d['depth_max_abs_err_vs_cpu'], 'abs_rel', d['abs_rel_sparse_metric'])"
