#!/bin/bash
# Round 4, session 39: the profile set of configs[4] bf16 once more (its cost-volume entry point now forms its window sums separably) and the lines that quote it.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s39
mkdir -p $OUT
bash tools/profile_round.sh r04_c5bf16 "--height 512 --width 1024 --frames 4 --depths 48 --bf16" 30 > $OUT/profile_c5bf16.log 2>&1; echo "profile c5 bf16 rc=$?"
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer > $OUT/c5_bf16.json 2> /dev/null
timeout 200 python bench.py --steps 200 --bf16 --no-primer --no-cpu-baseline > $OUT/c2_bf16.json 2> /dev/null
python - <<'PY'
import json
for f in ("c5_bf16", "c2_bf16"):
    d = json.loads(open(f"gpurun_out/r04_s39/{f}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "forward_api", round(d["forward_api"]["value"], 1), "bound", r["bound"], "frac", round(r["frac"], 3),
          "kernel_only", r.get("frac_kernel_only"), "stale" if "stale_profile" in r else "current", "cv", d["cost_volume_kernel"].get("rocprof_sad_us"), "depth err", d.get("depth_max_abs_err_vs_cpu"))
PY
tail -1 $OUT/c5_bf16.json > profiles/r04_c5bf16_bench.json
tail -1 $OUT/c2_bf16.json > profiles/r04_c2_bf16_line.json
mkdir -p $OUT/profiles && cp profiles/r04_c5bf16_* profiles/r04_c2_bf16_line.json $OUT/profiles/
