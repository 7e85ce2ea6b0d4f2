#!/bin/bash
# split_k finished inside the launch (KFIN instantiation: agent-scope atomic stores / loads, arrival counter, no fence, no finishing
# launch): parity, repeated runs against a race, interleaved A/B against the finishing-launch path.
OUT=gpurun_out/s21
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv or refine or upconv" > $OUT/pytest_conv.log 2>&1; echo "conv tests rc=$?"; tail -3 $OUT/pytest_conv.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "forward_matches or c2_config or in_flight or batch_independence or depth_large" 2>&1 | tail -1; done
b() { echo "$1: $(env $2 timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'kf/s', 'sum-of-kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3), 'conv ms', round(d['roofline']['conv_ms_per_step'],3))")"; }
timeout 300 python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1     # primer + warm box
for r in 1 2 3; do
  b "finishing launches" "MR_SPLITK_INLINE=0" ""
  b "split_k finished in the launch" "A=1" ""
done
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --dump-layers $OUT/layers.json > /dev/null 2>&1
MR_SPLITK_INLINE=0 timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --dump-layers $OUT/layers_legacy.json > /dev/null 2>&1
python - <<'PY'
import json
a = {r["name"]: r for r in json.load(open("gpurun_out/s21/layers.json"))}
b = {r["name"]: r for r in json.load(open("gpurun_out/s21/layers_legacy.json"))}
print("split-K layers: in-launch us | finishing-launch us")
for k in a:
    if a[k]["sched"] and a[k]["sched"][2] > 1:
        print(f"{k:28s} ks={a[k]['sched'][2]} {a[k]['seconds']*1e6:8.1f} {b[k]['seconds']*1e6:8.1f}")
PY
