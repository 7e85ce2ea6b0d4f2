#!/bin/bash
# Round 4, session 19: F(4x4,3x3) back in the product library (ABI 17) behind the c3 / configs[4] table entries of r04_s18: its kernel cases,
# the c3 / configs[4]-shaped model tests on both weight families, the c3 and configs[4] fp32 lines before / after (MR_TUNED_WINOGRAD = the old table).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s19
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd or wino" > $OUT/kernels.log 2>&1; echo "winograd kernel tests rc=$?"; tail -2 $OUT/kernels.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -s -k "c3_full or c5_shape or harsh_weights_c3 or forward_matches" > $OUT/model.log 2>&1; echo "model tests rc=$?"; grep -E "passed|failed|result" $OUT/model.log | tail -12 | cut -c1-400
git show HEAD:monorec_amd/tuned_winograd.json > $OUT/old_winograd.json 2>/dev/null || cp monorec_amd/tuned_winograd.json $OUT/old_winograd.json
B="--no-primer --no-cpu-baseline --no-forward-api"
for i in 1 2; do
  MR_TUNED_WINOGRAD=$OUT/old_winograd.json timeout 300 python bench.py $B --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3_old_$i.json 2> $OUT/c3_old_$i.err
  timeout 300 python bench.py $B --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3_new_$i.json 2> $OUT/c3_new_$i.err
done
MR_TUNED_WINOGRAD=$OUT/old_winograd.json timeout 300 python bench.py $B --steps 60 --height 512 --width 1024 --frames 4 --depths 48 > $OUT/c5_old.json 2> $OUT/c5_old.err
timeout 300 python bench.py $B --steps 60 --height 512 --width 1024 --frames 4 --depths 48 > $OUT/c5_new.json 2> $OUT/c5_new.err
python - <<'PY'
import json
for f in ("c3_old_1", "c3_new_1", "c3_old_2", "c3_new_2", "c5_old", "c5_new"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s19/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "kf/s; ms/step", round(d["ms_per_step"], 3), "depth err", d.get("depth_max_abs_err_vs_cpu"), "frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
