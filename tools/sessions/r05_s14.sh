#!/bin/bash
# Round 5, session 14: LDS bank conflicts of the patch / B-fragment reads taken out (tools/lds_banks.py): plane pitches of conv_wino44 / 44s / the 1 x k Cook-Toom
# forms / conv_b8 re-derived for 16-byte reads (64 banks, groups of 16 lanes); the lane-stride-2 patch reads of the in-register-transform kernels
# (F(2x2,3x3), ConvTranspose F(2x2,2x2), F(2,3) 1 x 3) as aligned 8-byte reads on a 32-mod-64 pitch.  Parity of every kernel, then old library (ab_old/) against new.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s14
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_b8.py -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/kernels.log 2>&1; echo "kernel tests rc=$?"; tail -15 $OUT/kernels.log | cut -c1-250
OLD=$REPO/ab_old/libmonorec_hip.so
run() {  # tag lib args...
  tag=$1; lib=$2; shift 2
  if [ "$lib" = old ]; then export MR_HIP_LIBRARY=$OLD; else unset MR_HIP_LIBRARY; fi
  timeout 300 python bench.py --no-cpu-baseline --no-primer --no-forward-api --dump-layers $OUT/layers_${tag}_$lib.json "$@" > $OUT/${tag}_$lib.json 2>/dev/null
  python - "$tag" "$lib" $OUT/${tag}_$lib.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["value"], 1), "kf/s; sum of kernels ms", round(d["device_ms_per_step_sum_of_kernels"], 3))
PY
  unset MR_HIP_LIBRARY
}
for rep in 1 2; do
  for lib in old new; do
    run c2_$rep $lib --steps 200
    run c3_$rep $lib --steps 40 --batch 8 --frames 4 --depths 64
    run c5bf16_$rep $lib --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16
  done
done
python - <<'PY'
import json
for cfg in ("c2", "c3", "c5bf16"):
    tot = {}
    for lib in ("old", "new"):
        rows = {}
        for rep in (1, 2):
            for r in json.load(open(f"gpurun_out/r05_s14/layers_{cfg}_{rep}_{lib}.json")):
                rows.setdefault(r["name"], []).append(r["seconds"] * 1e6)
        tot[lib] = {k: min(v) for k, v in rows.items()}
    print(cfg, "sum of kernels us: old", round(sum(tot["old"].values()), 1), "new", round(sum(tot["new"].values()), 1))
    ch = sorted(((tot["new"][k] - tot["old"][k], k) for k in tot["old"] if k in tot["new"]))
    for dlt, k in ch[:14] + ch[-6:]:
        if abs(dlt) > 0.02 * tot["old"][k] and abs(dlt) > 0.3:
            print("   %-34s %8.1f -> %8.1f us (%+.1f %%)" % (k, tot["old"][k], tot["new"][k], 100 * dlt / tot["old"][k]))
PY
