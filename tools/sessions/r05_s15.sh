#!/bin/bash
# Round 5, session 15 (final library): whole gpu suite + smoke, profile sets of c2 / c3 / configs[4] bf16 regenerated on ABI 18 (stamped per plan), the lines that quote them.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s15
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log | cut -c1-300
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/profile_round.sh r05_c2 > $OUT/profile_c2.log 2>&1; echo "profile c2 rc=$?"
bash tools/profile_round.sh r05_c3 "--batch 8 --frames 4 --depths 64" 20 > $OUT/profile_c3.log 2>&1; echo "profile c3 rc=$?"
bash tools/profile_round.sh r05_c5bf16 "--height 512 --width 1024 --frames 4 --depths 48 --bf16" 20 > $OUT/profile_c5bf16.log 2>&1; echo "profile c5bf16 rc=$?"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --in-flight 1 > $OUT/c2_inflight1.json 2> /dev/null
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --in-flight 2 --slot-streams 2 > $OUT/c2_2x2.json 2> /dev/null
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline > $OUT/c3.json 2> /dev/null
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api --cv-separable > $OUT/c3_separable.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer > $OUT/c5_bf16.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --lean-outputs --no-cpu-baseline --no-primer --no-forward-api > $OUT/c5_bf16_lean.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline --no-primer --no-forward-api > $OUT/c5_f32.json 2> /dev/null
python - <<'PY'
import json
for f in ("driver_style", "c2_inflight1", "c2_2x2", "c3", "c3_separable", "c5_bf16", "c5_bf16_lean", "c5_f32"):
    try:
        d = json.loads(open(f"gpurun_out/r05_s15/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        fa = d.get("forward_api", {}).get("value")
        print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "primed", d.get("value_host_primed") and round(d["value_host_primed"], 1),
              "forward_api", fa and round(fa, 1), "bound", r["bound"], "frac", round(r["frac"], 3), r.get("frac_source"),
              "vs_direct", r.get("vs_direct_conv_ceiling") and round(r["vs_direct_conv_ceiling"], 3), "pipelined", round(r["frac_pipelined"], 3), "stale" if "stale_profile" in r else "current",
              "launches", r.get("all_kernel_launches_per_step"), "host inputs", d.get("with_host_inputs", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "failed", repr(e))
PY
tail -1 $OUT/driver_style.json > profiles/r05_c2_bench_driver_style.json
tail -1 $OUT/c2_inflight1.json > profiles/r05_c2_inflight1_line.json
tail -1 $OUT/c2_2x2.json > profiles/r05_c2_two_slots_two_streams_line.json
tail -1 $OUT/c3.json > profiles/r05_c3_line.json
tail -1 $OUT/c3_separable.json > profiles/r05_c3_separable_line.json
tail -1 $OUT/c5_bf16.json > profiles/r05_c5bf16_line.json
tail -1 $OUT/c5_bf16_lean.json > profiles/r05_c5bf16_lean_line.json
tail -1 $OUT/c5_f32.json > profiles/r05_c5_f32_line.json
mkdir -p $OUT/profiles && cp profiles/r05_* $OUT/profiles/ 2>/dev/null; ls $OUT/profiles | wc -l
