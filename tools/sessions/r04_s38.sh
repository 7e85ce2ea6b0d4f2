#!/bin/bash
# Round 4, session 38: the bf16 configuration's cost-volume entry point (mr_cost_volume_b8_f32) with separable window sums and x * fp32(1/9):
# B8 kernel tests (the side-copy case), the bf16 model tests, cost-volume kernel tests (the exact path must be untouched), lines at configs[4] bf16 / c2 bf16.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s38
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_b8.py tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -k "b8 or bf16 or cost_volume or c5_shape" > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t.log | cut -c1-300
for i in 1 2; do
  timeout 300 python bench.py --no-primer --no-cpu-baseline --no-forward-api --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 > $OUT/c5b_$i.json 2> $OUT/c5b_$i.err
done
timeout 200 python bench.py --steps 200 --bf16 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c2_bf16.json 2> /dev/null
python - <<'PY'
import json
for f in ("c5b_1", "c5b_2", "c2_bf16"):
    d = json.loads(open(f"gpurun_out/r04_s38/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), "kf/s; sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3), "cv us", round(d["cost_volume_kernel"]["us"], 1))
PY
