#!/bin/bash
# Round 4, session 12: (a) does the LDS footprint of the one-workgroup-per-CU layers decide how much two keyframes in flight overlap?
# (MR_LDS_CAP_KB: chunks halved until a workgroup takes <= 78 / 52 KB, so workgroups of different keyframes share a CU), at 2 and 3 keyframes
# in flight; (b) one-plane cost-volume marching kernel with software prefetch (diagnostic library, MR_CV_PREFETCH).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s12
mkdir -p $OUT
B="--steps 200 --no-primer --no-cpu-baseline --no-forward-api"
for cap in 0 78 52; do
  for fl in 2 3; do
    MR_LDS_CAP_KB=$cap timeout 200 python bench.py $B --in-flight $fl > $OUT/c2_cap${cap}_if${fl}.json 2> $OUT/c2_cap${cap}_if${fl}.err
  done
  MR_LDS_CAP_KB=$cap timeout 200 python bench.py $B --in-flight 1 > $OUT/c2_cap${cap}_if1.json 2> $OUT/c2_cap${cap}_if1.err
done
python - <<'PY'
import json
for cap in (0, 78, 52):
    for fl in (1, 2, 3):
        f = f"gpurun_out/r04_s12/c2_cap{cap}_if{fl}.json"
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            print("cap", cap, "in flight", fl, round(d["value"], 1), "kf/s; sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
        except Exception as e:
            print(f, "failed", e)
PY
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for pf in 0 1; do
  echo "== MR_CV_PREFETCH=$pf"
  MR_CV_PREFETCH=$pf timeout 120 python tools/bench_cv.py --impl march --iters 200 2>&1 | tail -3
  MR_CV_PREFETCH=$pf timeout 120 python tools/bench_cv.py --impl march --iters 100 --height 512 --width 1024 --frames 4 --depths 48 2>&1 | tail -3
done
echo "== one plane per wave + prefetch at the configs[4] shape"
MR_CV_MARCH_DP=1 MR_CV_PREFETCH=1 timeout 120 python tools/bench_cv.py --impl march --iters 100 --height 512 --width 1024 --frames 4 --depths 48 2>&1 | tail -3
MR_CV_MARCH_DP=1 MR_CV_PREFETCH=0 timeout 120 python tools/bench_cv.py --impl march --iters 100 --height 512 --width 1024 --frames 4 --depths 48 2>&1 | tail -3
MR_CV_PREFETCH=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "cost_volume" 2>&1 | tail -2
