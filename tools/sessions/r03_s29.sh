#!/bin/bash
# Round 3, session 29 (the last 2 GPU-minutes): forward() with the host waiting for the keyframe before the output copy is enqueued
# (hip_forward_host_wait) next to the stream wait: parity / ownership tests and the forward_api figure both ways.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r03_s29
mkdir -p $OUT
# (the variant and its test were removed after this run: no gain)
timeout 70 python -m pytest tests/test_gpu_model.py -x -q -k "host_waiting or owned" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 60 python bench.py --steps 100 --no-cpu-baseline --no-primer --forward-ab > $OUT/forward_ab.json 2> $OUT/forward_ab.err
python - <<PY
import json
d = json.loads(open("$OUT/forward_ab.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "forward_api", round(d["forward_api"]["value"], 1), "host wait", round(d["forward_api_host_wait"]["value"], 1))
PY
