#!/bin/bash
# Round 4, session 36: conv_b8_kernel with the next chunk's input DMA instructions spread over the taps of the sweep (resident-weight mode, B8
# sources) against the burst behind the barrier (diagnostic library, MR_B8_DBG=16): B8 kernel tests, layer times, then the configs[4] bf16 line.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s36
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_b8.py -x -q > $OUT/k.log 2>&1; echo "b8 kernel tests rc=$?"; tail -2 $OUT/k.log | cut -c1-300
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for dbg in 0 16 0 16; do
  echo "== MR_B8_DBG=$dbg (0: spread, 16: burst)"
  for layer in enc0.1 dec3.1 enc0.1x; do
    MR_B8_DBG=$dbg timeout 200 python tools/bench_b8.py --layer $layer 2>/dev/null | tail -2 | cut -c1-200
  done
  MR_B8_DBG=$dbg timeout 200 python tools/bench_b8.py --layer enc1.0 --height 256 --width 512 2>/dev/null | tail -2 | cut -c1-200
done
unset MR_HIP_LIBRARY
for i in 1 2; do
  timeout 300 python bench.py --no-primer --no-cpu-baseline --no-forward-api --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 > $OUT/c5b_$i.json 2> $OUT/c5b_$i.err
done
python - <<'PY'
import json
for i in (1, 2):
    d = json.loads(open(f"gpurun_out/r04_s36/c5b_{i}.json").read().strip().splitlines()[-1])
    print("configs[4] bf16", round(d["value"], 1), "kf/s; sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
PY
