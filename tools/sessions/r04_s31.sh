#!/bin/bash
# Round 4, session 31: what the first timed steps of a 20-step line cost (prepare / wait / submit per step).
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s31
mkdir -p $OUT
for i in 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api --step-times > $OUT/c2_20_$i.json 2> $OUT/c2_20_$i.err
done
python - <<'PY'
import json
for f in ("c2_20_1", "c2_20_2"):
    d = json.loads(open(f"gpurun_out/r04_s31/{f}.json").read().strip().splitlines()[-1])
    m = d["step_marks_ms"]
    print(f, round(d["value"], 1), "elapsed", d["elapsed_ms"])
    print("  marks", [round(b - a, 2) for a, b in zip([0.0] + m[:-1], m)])
    print("  parts", d["step_parts_ms_prepare_wait_submit"])
PY
