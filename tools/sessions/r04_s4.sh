#!/bin/bash
# Round 4, session 4: conv_b8 with the tap loop software-pipelined over two register sets, fp32 staging compiled out of the pure-B8 instantiations.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s4
mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_b8.py -q > $OUT/b8_kernels.log 2>&1; echo "b8 kernel tests rc=$?"; tail -6 $OUT/b8_kernels.log | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_mode_end_to_end or c5_shape" > $OUT/model_bf16.log 2>&1; echo "bf16 model tests rc=$?"; tail -4 $OUT/model_bf16.log | cut -c1-300
for NB4 in 1 0; do
  MR_B8_NB4=$NB4 timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api --dump-layers $OUT/c5_bf16_layers_nb4_$NB4.json > $OUT/c5_bf16_nb4_$NB4.json 2> $OUT/c5_bf16_nb4_$NB4.err; echo "c5 bf16 NB4=$NB4 rc=$?"
done
timeout 200 python bench.py --steps 200 --bf16 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c2_bf16.json 2> $OUT/c2_bf16.err; echo "c2 bf16 rc=$?"
python - <<'PY'
import json
for f in ("c5_bf16_nb4_0", "c5_bf16_nb4_1", "c2_bf16"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s4/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 1), "kf/s, ms/step", round(d["ms_per_step"], 3), "conv ms", round(r["conv_ms_per_step"], 3), "cv us", round(d["cost_volume_kernel"]["us"], 1),
              "bound", r["bound"], "frac", round(r["frac"], 3), "sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
    except Exception as e:
        print(f, "failed", e)
a = {r["name"]: r for r in json.load(open("gpurun_out/r04_s4/c5_bf16_layers_nb4_0.json"))}
b = {r["name"]: r for r in json.load(open("gpurun_out/r04_s4/c5_bf16_layers_nb4_1.json"))}
for n in sorted(a, key=lambda n: -a[n]["seconds"])[:30]:
    print(f"{n:24s} {a[n]['seconds']*1e6:8.1f} us {a[n]['sched']}   nb4: {b[n]['seconds']*1e6:8.1f} us {b[n]['sched']}")
PY
