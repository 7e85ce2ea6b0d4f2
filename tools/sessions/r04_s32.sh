#!/bin/bash
# Round 4, session 32: is the slow first timed step a cold host thread?  20-step lines with a host-only busy loop (prepare() calls) of 0 / 3 ms before t0.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s32
mkdir -p $OUT
for i in 1 2 3; do
  for p in 0 3; do
    timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api --step-times --host-prime-ms $p > $OUT/c2_p${p}_$i.json 2> $OUT/c2_p${p}_$i.err
  done
done
python - <<'PY'
import json
for i in (1, 2, 3):
    for p in (0, 3):
        d = json.loads(open(f"gpurun_out/r04_s32/c2_p{p}_{i}.json").read().strip().splitlines()[-1])
        print("prime", p, "run", i, round(d["value"], 1), "elapsed", d["elapsed_ms"], "first steps (prepare, wait, submit):", d["step_parts_ms_prepare_wait_submit"][:3])
PY
