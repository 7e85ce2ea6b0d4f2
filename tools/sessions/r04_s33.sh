#!/bin/bash
# Round 4, session 33: the committed c2 line once more with the final bench.py (driver command), and the in-flight-1 line of the same session.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s33
mkdir -p $OUT
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --in-flight 1 > $OUT/c2_inflight1.json 2> /dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_s33/driver_style.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver_style", round(d["value"], 1), "200:", round(d["value_200_steps"], 1), "forward_api", round(d["forward_api"]["value"], 1), "frac", round(r["frac"], 3), "kernel_only", r.get("frac_kernel_only"), "pipelined", round(r["frac_pipelined"], 3), "stale" if "stale_profile" in r else "current", d["config"].get("host_prime_ms"), d.get("secondary_dynamic_batching"))
d1 = json.loads(open("gpurun_out/r04_s33/c2_inflight1.json").read().strip().splitlines()[-1])
print("in flight 1", round(d1["value"], 1))
PY
