#!/bin/bash
# Second pass of the one-channel kernels (16 loads in flight in the classifier, head weights from LDS), DataParallel fix,
# one-plane-per-wave cost volume by default where it pays.
OUT=gpurun_out/s17
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "heads or classifier or marching or cost_volume" > $OUT/pytest_k.log 2>&1; echo "kernel tests rc=$?"; tail -3 $OUT/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "data_parallel or forward_matches or c2_config or in_flight or owned or dynamic_batching" > $OUT/pytest_model.log 2>&1; echo "model tests rc=$?"; tail -3 $OUT/pytest_model.log
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_model.py -m gpu -q -k "data_parallel" 2>&1 | tail -1; done
cv() { echo "cv $1: $(env $2 timeout 120 python tools/bench_cv.py --impl march $3 2>/dev/null | tail -1)"; }
cv "c2 default" "A=1" ""
cv "c2 DP=2" "MR_CV_MARCH_DP=2" ""
cv "c5 shape default" "A=1" "--height 512 --width 1024 --frames 4 --depths 48 --iters 50"
cv "c5 shape DP=1" "MR_CV_MARCH_DP=1" "--height 512 --width 1024 --frames 4 --depths 48 --iters 50"
b() { echo "$1: $(env $2 timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'kf/s', 'sum-of-kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3), 'conv ms', round(d['roofline']['conv_ms_per_step'],3), 'one-channel us', round(d.get('one_channel_layers',{}).get('us_per_step',0),1), 'cv us', round(d['cost_volume_kernel']['us'],1))")"; }
timeout 300 python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1     # primer + warm box
for r in 1 2; do
  b "conv launches for the one-channel layers, DP=2" "MR_ONE_CHANNEL_KERNELS=0 MR_CV_MARCH_DP=2" ""
  b "defaults (one-channel kernels, DP auto)" "A=1" ""
done
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --dump-layers $OUT/layers.json > $OUT/bench_c2.json 2>/dev/null
python - <<'PY'
import json
for r in json.load(open("gpurun_out/s17/layers.json")):
    if r["name"] in ("mask.classifier", "depth.heads", "cost_volume", "resnet.normalize", "mask.dec3.2", "depth.dec4.2"):
        print(f'{r["name"]:24s} {r["seconds"]*1e6:8.1f} us')
PY
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-200 $OUT/bench_driver.json
