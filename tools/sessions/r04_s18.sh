#!/bin/bash
# Round 4, session 18: F(4x4,3x3) (diagnostic library) against the F(2x2,3x3) variants on the c3 and configs[4] shapes - VERDICT r3 #6 asked for
# these tables before the kernel's place is decided; round 3 only measured c2.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s18
mkdir -p $OUT
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
timeout 900 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --min-pixels 8192 --emit $OUT/wino_c3.json > $OUT/wino_c3.log 2>&1; echo "c3 rc=$?"
timeout 900 python tools/bench_wino.py --height 512 --width 1024 --frames 4 --depths 48 --min-pixels 8192 --emit $OUT/wino_c5.json > $OUT/wino_c5.log 2>&1; echo "c5 rc=$?"
python - <<'PY'
import json
for tag in ("c3", "c5"):
    print("==", tag)
    for l in open(f"gpurun_out/r04_s18/wino_{tag}.log"):
        if not l.startswith("{"):
            continue
        r = json.loads(l)
        if "name" not in r:
            print(l.strip()[:300]); continue
        ts = {k[4:-3]: v for k, v in r.items() if k.startswith("wino") and k.endswith("_us")}
        print(f"{r['name']:14s} cin {r['cin']:4d} cout {r['cout']:4d} {r['hw']} n {r['n']:3d} direct {r['direct_us']:8.1f} " + " ".join(f"{k}:{v:.1f}" for k, v in ts.items()) + f" best {r['best']}  maxdiff31 {r.get('wino31_maxdiff')}")
PY
