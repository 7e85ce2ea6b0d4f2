#!/bin/bash
# Round 4, session 24: conv3x3_wino44_kernel with one 16-byte + one 4-byte LDS read per patch row (outer columns by DPP from the neighbouring
# tiles) and the shorter generated transform chains (a +-1 term opens every fmaf chain: F(4,3) input transform 17 -> 13 instructions; the 1-D
# Cook-Toom kernels share the header): parity cases of both kernel families, layer times at c3 / c2, the 1-D layers at c2.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s24
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd or wino or cooktoom" > $OUT/k.log 2>&1; echo "kernel tests rc=$?"; tail -2 $OUT/k.log | cut -c1-300
timeout 600 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --min-pixels 8192 > $OUT/wino_c3.log 2>&1
timeout 600 python tools/bench_wino.py --min-pixels 8192 > $OUT/wino_c2.log 2>&1
python - <<'PY'
import json
for tag in ("c3", "c2"):
    print("==", tag)
    for l in open(f"gpurun_out/r04_s24/wino_{tag}.log"):
        if not l.startswith("{"):
            continue
        r = json.loads(l)
        if "name" not in r:
            print(l.strip()[:300]); continue
        ts = {k[4:-3]: v for k, v in r.items() if k.startswith("wino") and k.endswith("_us")}
        print(f"{r['name']:14s} cin {r['cin']:4d} cout {r['cout']:4d} {r['hw']} n {r['n']:3d} direct {r['direct_us']:8.1f} " + " ".join(f"{k}:{v:.1f}" for k, v in ts.items()) + f" best {r['best']}")
PY
timeout 600 python tools/bench_wino1d.py --no-upconv 2>/dev/null | tail -3 | cut -c1-400
