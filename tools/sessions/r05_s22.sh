#!/bin/bash
# Round 5, session 22: the driver command on the handed-over tree (HEAD), for the record.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s22
mkdir -p $OUT
timeout 330 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_s22/driver_style.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("driver-style", round(d["value"], 1), "kf/s; 200:", round(d["value_200_steps"], 1), "primed", round(d["value_host_primed"], 1), "forward_api", round(d["forward_api"]["value"], 1),
      "frac", round(r["frac"], 3), r["frac_source"], "stale" if "stale_profile" in r else "current", "cpu", round(d["cpu_baseline"]["value"], 3), "depth vs cpu", d.get("depth_max_abs_err_vs_cpu"))
PY
