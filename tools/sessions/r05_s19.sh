#!/bin/bash
# Round 5, session 19: the tree as it will be handed over - the new 320x640 parity test, the whole gpu suite, smoke, the driver command once more.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s19
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "oxford" > $OUT/oxrc.log 2>&1; echo "320x640 tests rc=$?"; grep -E "320x640|passed|failed" $OUT/oxrc.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $OUT/suite.log | tail -2
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_s19/driver_style.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver-style", round(d["value"], 1), "kf/s; 200:", round(d["value_200_steps"], 1), "primed", round(d["value_host_primed"], 1), "forward_api", round(d["forward_api"]["value"], 1),
      "frac", round(r["frac"], 3), r["frac_source"], "stale" if "stale_profile" in r else "current", "cpu", d["cpu_baseline"]["value"], "depth vs cpu", d.get("depth_max_abs_err_vs_cpu"))
PY
