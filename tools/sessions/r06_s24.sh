#!/bin/bash
# Round 6, session 24: forward() on the caller's stream (hip_forward_on_callers_stream=True, the new default): the whole model test file, then
# `model(data)` both ways against the in-flight-1 loop (tools/forward_rate.py, interleaved rounds), then the forward_api leg of the bench line.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s24
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_evaluate_loop.py -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/model_tests.log 2>&1; echo "model tests rc=$?"; tail -8 $OUT/model_tests.log | cut -c1-300
for rep in 1 2 3; do timeout 300 python tools/forward_rate.py 2>/dev/null | tail -1; done | tee $OUT/forward_rate.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $OUT/driver_style_c2.json 2>/dev/null; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_s24/driver_style_c2.json").read().strip().splitlines()[-1])
print("driver-style", round(d["value"], 1), "200:", round(d.get("value_200_steps", 0), 1), "forward_api", round(d["forward_api"]["value"], 1), d["forward_api"].get("outputs_owned_by_caller"))
PY
