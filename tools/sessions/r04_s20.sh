#!/bin/bash
# Round 4, session 20: A/B of the F(4x4,3x3) table entries (r04_s19's "old" leg had no git history on the box and ran the new table twice).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s20
mkdir -p $OUT
OLD=$REPO/tools/sessions/_old_winograd_r04_s18.json     # = git show 98e4112:monorec_amd/tuned_winograd.json (written before the call, not committed)
B="--no-primer --no-cpu-baseline --no-forward-api"
for i in 1 2 3; do
  MR_TUNED_WINOGRAD=$OLD timeout 300 python bench.py $B --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3_old_$i.json 2> $OUT/c3_old_$i.err
  timeout 300 python bench.py $B --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3_new_$i.json 2> $OUT/c3_new_$i.err
done
MR_TUNED_WINOGRAD=$OLD timeout 300 python bench.py $B --steps 60 --height 512 --width 1024 --frames 4 --depths 48 > $OUT/c5_old.json 2> $OUT/c5_old.err
timeout 300 python bench.py $B --steps 60 --height 512 --width 1024 --frames 4 --depths 48 > $OUT/c5_new.json 2> $OUT/c5_new.err
python - <<'PY'
import json
for f in ("c3_old_1", "c3_new_1", "c3_old_2", "c3_new_2", "c3_old_3", "c3_new_3", "c5_old", "c5_new"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s20/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "kf/s; ms/step", round(d["ms_per_step"], 3), "sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3), "conv ms", round(d["roofline"]["conv_ms_per_step"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
