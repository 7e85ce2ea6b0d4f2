#!/bin/bash
# Round 4, session 6: conv_b8 with 16-byte stores (lane swap), deferred flush behind the next tile's first sweep, bias hoisted; tests, ablations, c5 line.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s6
mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_b8.py -q > $OUT/b8_kernels.log 2>&1; echo "b8 kernel tests rc=$?"; tail -6 $OUT/b8_kernels.log | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_mode_end_to_end or c5_shape" > $OUT/model_bf16.log 2>&1; echo "bf16 model tests rc=$?"; tail -4 $OUT/model_bf16.log | cut -c1-300
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api --dump-layers $OUT/c5_bf16_layers.json > $OUT/c5_bf16.json 2> $OUT/c5_bf16.err; echo "c5 bf16 rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_s6/c5_bf16.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("c5_bf16", round(d["value"], 1), "kf/s, ms/step", round(d["ms_per_step"], 3), "conv ms", round(r["conv_ms_per_step"], 3), "cv us", round(d["cost_volume_kernel"]["us"], 1),
      "frac", round(r["frac"], 3), "sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
rows = json.load(open("gpurun_out/r04_s6/c5_bf16_layers.json"))
for r in sorted(rows, key=lambda r: -r["seconds"])[:26]:
    print(f"{r['name']:24s} {r['seconds']*1e6:8.1f} us {r['sched']}")
PY
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for L in enc0.1 enc0.1x; do
  for DBG in 0 1 2 8 9 15; do
    MR_B8_DBG=$DBG timeout 60 python tools/bench_b8.py --layer $L --scheds 3,4,8 3,2,8 2>&1 | grep "sched"
  done
done | tee $OUT/ablate.log
