#!/bin/bash
# Round 4, session 35 (final library: F(4x4,3x3) kernel with spread DMA issue): whole gpu suite + smoke, profile sets of c2 and c3 once more, the lines that quote them.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s35
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log | cut -c1-300
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/profile_round.sh r04_c2 > $OUT/profile_c2.log 2>&1; echo "profile c2 rc=$?"
bash tools/profile_round.sh r04_c3 "--batch 8 --frames 4 --depths 64" 20 > $OUT/profile_c3.log 2>&1; echo "profile c3 rc=$?"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --in-flight 1 > $OUT/c2_inflight1.json 2> /dev/null
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline > $OUT/c3.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline --no-primer > $OUT/c5_f32.json 2> /dev/null
python - <<'PY'
import json
for f in ("driver_style", "c2_inflight1", "c3", "c5_f32"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s35/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        fa = d.get("forward_api", {}).get("value")
        print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "forward_api", fa and round(fa, 1), "bound", r["bound"], "frac", round(r["frac"], 3),
              "kernel_only", r.get("frac_kernel_only") and round(r["frac_kernel_only"], 3), "pipelined", round(r["frac_pipelined"], 3), "stale" if "stale_profile" in r else "current",
              "launches", r["all_kernel_launches_per_step"], "host inputs", d.get("with_host_inputs", {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -1 $OUT/driver_style.json > profiles/r04_c2_bench.json
tail -1 $OUT/c3.json > profiles/r04_c3_bench.json
tail -1 $OUT/c5_f32.json > profiles/r04_c5_f32_line.json
tail -1 $OUT/c2_inflight1.json > profiles/r04_c2_inflight1_line.json
mkdir -p $OUT/profiles && cp profiles/r04_* $OUT/profiles/ 2>/dev/null; ls $OUT/profiles | wc -l
