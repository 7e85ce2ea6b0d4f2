#!/bin/bash
# Round 4, session 14: where do the 20-step lines lose their 7 % against the 200-step lines?  Host time stamps after every timed step.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s14
mkdir -p $OUT
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api --step-times > $OUT/c2_20_$i.json 2> $OUT/c2_20_$i.err
done
timeout 200 python bench.py --steps 60 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api --step-times > $OUT/c2_60.json 2> $OUT/c2_60.err
python - <<'PY'
import json
for f in ("c2_20_1", "c2_20_2", "c2_20_3", "c2_60"):
    d = json.loads(open(f"gpurun_out/r04_s14/{f}.json").read().strip().splitlines()[-1])
    m = d["step_marks_ms"]
    print(f, round(d["value"], 1), "elapsed", d["elapsed_ms"], "marks", [round(b - a, 2) for a, b in zip([0.0] + m[:-1], m)])
PY
