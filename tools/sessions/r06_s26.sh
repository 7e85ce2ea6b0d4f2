#!/bin/bash
# Round 6, session 26: the tree as handed over - whole gpu suite, smoke, the driver command, c3 and configs[4] bf16 lines.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s26
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -6 $OUT/suite.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style_c2.json 2> $OUT/driver_style_c2.err; echo "driver-style rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_s26/driver_style_c2.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("driver-style", round(d["value"], 1), "kf/s; 200:", round(d.get("value_200_steps", 0), 1), "primed", round(d.get("value_host_primed", 0), 1), "forward_api", round(d["forward_api"]["value"], 1),
      "frac", round(r["frac"], 3), r["frac_source"], "kernel_only", r.get("kernel_only", {}).get("frac"), "dominant", r["dominant"]["name"], round(r["dominant"]["avg_us"], 2), round(r["dominant"]["frac"], 3),
      "cpu", round(d["cpu_baseline"]["value"], 3), d["cpu_baseline"].get("port_vs_reference_range"), "depth vs cpu", d.get("depth_max_abs_err_vs_cpu"), "skip_l4", round(d["secondary_skip_dead_layer4"]["value"], 1))
PY
timeout 400 python bench.py --batch 8 --frames 4 --depths 64 --steps 30 --no-primer > $OUT/c3_line.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_s26/c3_line.json').read().strip().splitlines()[-1]); print('c3', round(d['value'],1), d['roofline']['frac'], d.get('depth_max_abs_err_vs_cpu'), 'forward_api', round(d['forward_api']['value'],1))"
timeout 600 python bench.py --height 512 --width 1024 --frames 4 --depths 48 --bf16 --steps 60 --no-primer > $OUT/c5bf16_line.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_s26/c5bf16_line.json').read().strip().splitlines()[-1]); print('c5 bf16', round(d['value'],1), d['roofline']['frac'], d.get('depth_max_abs_err_vs_cpu'), 'forward_api', round(d['forward_api']['value'],1))"
