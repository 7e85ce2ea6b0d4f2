#!/bin/bash
# Round 5, session 11: forward() with its encoder stage on the NEXT slot's stream (r05_s10: on a fifth stream behind the four slots it fell from 566 to 533),
# one slot = two streams again, and where a 20-step line loses against a 200-step line with four slots (step marks).
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s11
mkdir -p $OUT
for rep in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-primer --step-times > $OUT/c2_marks_$rep.json 2> $OUT/c2_marks_$rep.err; echo "c2 rc=$?"
done
timeout 200 python bench.py --steps 200 --in-flight 1 --no-cpu-baseline --no-primer > $OUT/c2_if1.json 2> /dev/null
timeout 200 python bench.py --steps 200 --in-flight 2 --slot-streams 2 --no-cpu-baseline --no-primer > $OUT/c2_2x2.json 2> /dev/null
python - <<'PY'
import json
for f in ("c2_marks_1", "c2_marks_2", "c2_if1", "c2_2x2"):
    try:
        d = json.loads(open(f"gpurun_out/r05_s11/{f}.json").read().strip().splitlines()[-1])
        fa = d.get("forward_api", {}).get("value")
        print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "primed:", d.get("value_host_primed") and round(d["value_host_primed"], 1),
              "forward_api", fa and round(fa, 1), "streams/slot", d["config"].get("streams_per_slot"), "host enqueue", round(d["host_enqueue_ms"], 3))
        if "step_marks_ms" in d:
            m = d["step_marks_ms"]
            print("   marks", m, "\n   deltas", [round(b - a, 2) for a, b in zip([0] + m[:-1], m)], "\n   parts", d["step_parts_ms_prepare_wait_submit"])
    except Exception as e:
        print(f, "failed", repr(e))
PY
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_evaluate_loop.py -m gpu -q -p no:cacheprovider -k "fixed_order or forward or evaluat or arenas or owned" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log | cut -c1-300
