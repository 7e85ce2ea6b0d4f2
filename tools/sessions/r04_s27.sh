#!/bin/bash
# Round 4, session 27: every MFMA kernel compiled with -fno-slp-vectorize (the SLP pass packs adjacent fp32 adds / multiplies of the transforms
# and epilogues into v_pk_*_f32, which MI355X_MICROARCH.md prices as an anti-lever beside MFMAs) against the library built without the flag
# (monorec_amd/libmonorec_hip_slp_baseline.so, copied before the rebuild): interleaved lines at c2, c3, configs[4] bf16.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s27
mkdir -p $OUT
BASE=$REPO/monorec_amd/libmonorec_hip_slp_baseline.so
B="--no-primer --no-cpu-baseline --no-forward-api"
for i in 1 2 3; do
  MR_HIP_LIBRARY=$BASE timeout 200 python bench.py $B --steps 200 > $OUT/c2_slp_$i.json 2> $OUT/c2_slp_$i.err
  timeout 200 python bench.py $B --steps 200 > $OUT/c2_noslp_$i.json 2> $OUT/c2_noslp_$i.err
done
for i in 1 2; do
  MR_HIP_LIBRARY=$BASE timeout 300 python bench.py $B --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3_slp_$i.json 2> $OUT/c3_slp_$i.err
  timeout 300 python bench.py $B --steps 40 --batch 8 --frames 4 --depths 64 > $OUT/c3_noslp_$i.json 2> $OUT/c3_noslp_$i.err
  MR_HIP_LIBRARY=$BASE timeout 300 python bench.py $B --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 > $OUT/c5b_slp_$i.json 2> $OUT/c5b_slp_$i.err
  timeout 300 python bench.py $B --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 > $OUT/c5b_noslp_$i.json 2> $OUT/c5b_noslp_$i.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r04_s27/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d["value"], 1), "kf/s; sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
