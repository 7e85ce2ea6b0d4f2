#!/bin/bash
# Round 4, session 22: conv3x3_wino44_kernel with the 36 A operands of a channel quad read up front (18 8-byte LDS reads behind a scheduling
# barrier instead of one 4-byte read in front of every MFMA) and without SLP-packed VALU: parity cases, then the c3 / c2 layer times.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s22
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd44 or wino44" > $OUT/k.log 2>&1; echo "wino44 kernel tests rc=$?"; tail -2 $OUT/k.log | cut -c1-300
timeout 600 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --min-pixels 8192 > $OUT/wino_c3.log 2>&1
timeout 600 python tools/bench_wino.py --min-pixels 8192 > $OUT/wino_c2.log 2>&1
python - <<'PY'
import json
for tag in ("c3", "c2"):
    print("==", tag)
    for l in open(f"gpurun_out/r04_s22/wino_{tag}.log"):
        if not l.startswith("{"):
            continue
        r = json.loads(l)
        if "name" not in r:
            print(l.strip()[:300]); continue
        ts = {k[4:-3]: v for k, v in r.items() if k.startswith("wino") and k.endswith("_us")}
        print(f"{r['name']:14s} cin {r['cin']:4d} cout {r['cout']:4d} {r['hw']} n {r['n']:3d} direct {r['direct_us']:8.1f} " + " ".join(f"{k}:{v:.1f}" for k, v in ts.items()) + f" best {r['best']}")
PY
