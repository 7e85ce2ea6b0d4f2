#!/bin/bash
# Round 4, session 1: the host-side changes on hardware - forward() with caller-owned output arenas, harsh weight family, hip_exact_convs,
# tightened F(4,7) bar - then the whole suite, driver-style / 200-step / in-flight-1 lines and a one-keyframe-at-a-time trace of HEAD.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s1
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "arenas or handle_intact or wrongly_shaped or harsh or owned" > $OUT/new_tests.log 2>&1; echo "new model tests rc=$?"; tail -3 $OUT/new_tests.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "cooktoom" > $OUT/cooktoom.log 2>&1; echo "cooktoom rc=$?"; tail -2 $OUT/cooktoom.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
python - <<'PY'
import json
for f in ("driver_style",):
    try:
        d = json.loads(open(f"gpurun_out/r04_s1/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "kf/s; 200 steps", round(d.get("value_200_steps", 0), 1), "forward_api", round(d["forward_api"]["value"], 1),
              "host_enqueue_ms", round(d["host_enqueue_ms"], 3), "cpu ms/kf", round(d["host_cpu_ms_per_keyframe"], 3),
              "exact", {k: round(v["value"], 1) for k, v in d.get("secondary_exact_convs", {}).items() if isinstance(v, dict)},
              "stale" if "stale_profile" in d["roofline"] else "profile current", "frac_pipelined", round(d["roofline"]["frac_pipelined"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
for CFG in "--in-flight 1" "--in-flight 3"; do
  timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api $CFG 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG', round(d['value'],1))"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_seq -o t -- python $REPO/bench.py --steps 40 --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace_seq.log 2>&1
cd $REPO
python tools/summarize_prof.py --tag r04a_c2 --stats-seq $(find $OUT/trace_seq -name "*_results.db" | head -1) > /dev/null 2>&1
mkdir -p $OUT/profiles && cp profiles/r04a_c2_* $OUT/profiles/ 2>/dev/null
find $OUT -name "*.db" -delete
ls $OUT/profiles
