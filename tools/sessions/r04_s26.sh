#!/bin/bash
# Round 4, session 26: F(4x4,3x3) for the four full-resolution 3x3 layers of c2 (mask.enc0.*, mask.dec3.1 / .2) with the round-4 kernel (A operands
# up front): interleaved A/B of the c2 line (candidate table = tools/sessions/_cand_winograd_c2f44.json, written before the call), parity of the
# c2-shaped model tests on it; c3 / configs[4] fp32 lines on the new kernel.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s26
mkdir -p $OUT
CAND=$REPO/tools/sessions/_cand_winograd_c2f44.json
B="--steps 200 --no-primer --no-cpu-baseline --no-forward-api"
for i in 1 2 3; do
  timeout 200 python bench.py $B > $OUT/c2_base_$i.json 2> $OUT/c2_base_$i.err
  MR_TUNED_WINOGRAD=$CAND timeout 200 python bench.py $B > $OUT/c2_f44_$i.json 2> $OUT/c2_f44_$i.err
done
MR_TUNED_WINOGRAD=$CAND timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "c2_config or reference_example or harsh_weights_c2 or two_keyframes" > $OUT/model.log 2>&1; echo "model tests on the candidate table rc=$?"; tail -2 $OUT/model.log | cut -c1-300
timeout 300 python bench.py --no-primer --no-cpu-baseline --no-forward-api --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3.json 2> $OUT/c3.err
timeout 300 python bench.py --no-primer --no-cpu-baseline --no-forward-api --steps 60 --height 512 --width 1024 --frames 4 --depths 48 > $OUT/c5.json 2> $OUT/c5.err
python - <<'PY'
import json
for f in ("c2_base_1", "c2_f44_1", "c2_base_2", "c2_f44_2", "c2_base_3", "c2_f44_3", "c3", "c5"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s26/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "kf/s; ms/step", round(d["ms_per_step"], 3), "sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
