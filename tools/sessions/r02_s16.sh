#!/bin/bash
# One-channel layers on their own kernels (depth heads in one launch, classifier + mask multiply), staggered-DMA conv
# instantiation and marching cost volume with more waves (TY / one plane per wave): parity first, then interleaved A/B.
OUT=gpurun_out/s16
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "heads or classifier" > $OUT/pytest_heads.log 2>&1; echo "heads tests rc=$?"; tail -3 $OUT/pytest_heads.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "not c3_full and not c5_shape" > $OUT/pytest_model.log 2>&1; echo "model tests rc=$?"; tail -3 $OUT/pytest_model.log
MR_CV_MARCH_DP=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "marching or cost_volume" > $OUT/pytest_cv_dp1.log 2>&1; echo "cv (one plane per wave) tests rc=$?"; tail -3 $OUT/pytest_cv_dp1.log
MR_CONV_STAGGER=4 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "c2_config or forward_matches or refine or upconv or register_tile" > $OUT/pytest_stagger.log 2>&1; echo "stagger tests rc=$?"; tail -3 $OUT/pytest_stagger.log
cv() { echo "cv $1: $(env $2 timeout 120 python tools/bench_cv.py --impl march $3 2>/dev/null | tail -1)"; }
cv "c2 default" "A=1" ""
cv "c2 TY=19" "MR_CV_MARCH_TY=19" ""
cv "c2 TY=21" "MR_CV_MARCH_TY=21" ""
cv "c2 DP=1" "MR_CV_MARCH_DP=1" ""
cv "c2 DP=1 TY=37" "MR_CV_MARCH_DP=1 MR_CV_MARCH_TY=37" ""
cv "c2 DP=1 TY=64" "MR_CV_MARCH_DP=1 MR_CV_MARCH_TY=64" ""
cv "c3 default" "A=1" "--batch 8 --frames 4 --depths 64 --iters 20"
cv "c3 DP=1" "MR_CV_MARCH_DP=1" "--batch 8 --frames 4 --depths 64 --iters 20"
b() { echo "$1: $(env $2 timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'kf/s', 'sum-of-kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3), 'conv ms', round(d['roofline']['conv_ms_per_step'],3), 'one-channel us', round(d.get('one_channel_layers',{}).get('us_per_step',0),1))")"; }
timeout 300 python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1     # primer + warm box
for r in 1 2; do
  b "conv launches for the one-channel layers" "MR_ONE_CHANNEL_KERNELS=0" ""
  b "one-channel kernels" "A=1" ""
  b "one-channel kernels + stagger>=4" "MR_CONV_STAGGER=4" ""
  b "one-channel kernels + stagger>=8" "MR_CONV_STAGGER=8" ""
done
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --dump-layers $OUT/layers.json > $OUT/bench_c2.json 2>/dev/null
MR_CONV_STAGGER=4 timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --dump-layers $OUT/layers_stagger4.json > /dev/null 2>&1
python - <<'PY'
import json
a = {r["name"]: r["seconds"] for r in json.load(open("gpurun_out/s16/layers.json"))}
b = {r["name"]: r["seconds"] for r in json.load(open("gpurun_out/s16/layers_stagger4.json"))}
print("layer  default us  stagger us")
for k in a:
    if abs(a[k] - b.get(k, 0)) > 1.5e-6 or k in ("mask.classifier", "depth.heads"):
        print(f"{k:28s} {a[k]*1e6:8.1f} {b.get(k,0)*1e6:8.1f}")
PY
