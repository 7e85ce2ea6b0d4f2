#!/bin/bash
# Round 5, session 10: the new defaults - four in-flight slots with one stream each (r05_s9: 820-827 keyframes/s at c2 over 200 steps, 749-754 over 20) -
# on the driver command, c3 and configs[4]; the host-path tests, the evaluation loop and the two-rank bench test.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s10
mkdir -p $OUT
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline > $OUT/c3.json 2> /dev/null
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api --in-flight 2 > $OUT/c3_if2.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer > $OUT/c5_bf16.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api --in-flight 2 > $OUT/c5_bf16_if2.json 2> /dev/null
python - <<'PY'
import json
for f in ("driver_style", "c3", "c3_if2", "c5_bf16", "c5_bf16_if2"):
    try:
        d = json.loads(open(f"gpurun_out/r05_s10/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        fa = d.get("forward_api", {}).get("value")
        print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "primed:", d.get("value_host_primed") and round(d["value_host_primed"], 1),
              "forward_api", fa and round(fa, 1), "frac", round(r["frac"], 3), r["frac_source"], "pipelined", round(r["frac_pipelined"], 3), "host cpu", round(d["host_cpu_ms_per_keyframe"], 2),
              "host inputs", d.get("with_host_inputs", {}).get("value"), "loading", d.get("with_data_loading", {}).get("value"),
              "batching", d.get("secondary_dynamic_batching", {}).get("requests_per_launch_2"), d.get("secondary_dynamic_batching", {}).get("requests_per_launch_4"),
              "exact", d.get("secondary_exact_convs", {}).get("f2_forms_only"), d.get("secondary_exact_convs", {}).get("direct_kernel_only"), "bf16x3", d.get("secondary_bf16x3", {}).get("value"))
    except Exception as e:
        print(f, "failed", repr(e))
PY
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_evaluate_loop.py tests/test_pointcloud.py tests/test_gpu_metrics.py -m gpu -q -p no:cacheprovider -k "fixed_order or flight or batching or pipeline or separate_streams or data_parallel or arenas or owned or evaluat or rank or pointcloud or loop or forward_between or inputs_are" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log | cut -c1-300
