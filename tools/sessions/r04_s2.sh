#!/bin/bash
# Round 4, session 2: FIRST hardware run of the bf16 path with channel-blocked bf16 activation storage (csrc/conv_b8.hip), the fast split-K
# finishing kernel and the library without F(4x4,3x3) / F(2,7) (ABI 16).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s2
mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_b8.py -q > $OUT/b8_kernels.log 2>&1; echo "b8 kernel tests rc=$?"; tail -15 $OUT/b8_kernels.log | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_mode_end_to_end or c5_shape or fixtures_own_matrices" > $OUT/model_bf16.log 2>&1; echo "bf16 model tests rc=$?"; tail -8 $OUT/model_bf16.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "split or conv2d or splitk" > $OUT/splitk.log 2>&1; echo "split-K kernel tests rc=$?"; tail -2 $OUT/splitk.log
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c2_200.json 2> $OUT/c2_200.err; echo "c2 rc=$?"
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api --dump-layers $OUT/c5_bf16_layers.json > $OUT/c5_bf16.json 2> $OUT/c5_bf16.err; echo "c5 bf16 rc=$?"
MR_B8=0 timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api > $OUT/c5_bf16_fp32storage.json 2> $OUT/c5_bf16_fp32storage.err; echo "c5 bf16 (fp32 storage) rc=$?"
timeout 200 python bench.py --steps 200 --bf16 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c2_bf16.json 2> $OUT/c2_bf16.err; echo "c2 bf16 rc=$?"
python - <<'PY'
import json
for f in ("c2_200", "c5_bf16", "c5_bf16_fp32storage", "c2_bf16"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s2/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 1), "kf/s, ms/step", round(d["ms_per_step"], 3), "conv ms", round(r["conv_ms_per_step"], 3), "cv us", round(d["cost_volume_kernel"]["us"], 1),
              "bound", r["bound"], "frac", round(r["frac"], 3), "launches", r["all_kernel_launches_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
python - <<'PY'
import json
try:
    rows = json.load(open("gpurun_out/r04_s2/c5_bf16_layers.json"))
    for r in sorted(rows, key=lambda r: -r["seconds"])[:24]:
        print(f"{r['name']:24s} {r['seconds']*1e6:8.1f} us  {2*r['ref_macs']/max(r['seconds'],1e-9)/1e12:7.1f} TF")
except Exception as e:
    print("layers failed", e)
PY
