#!/bin/bash
# Round 5, session 20: a MEASURED schedule table for conv_b8_kernel at the configs[4] shape (tools/tune_b8.py: every launch signature rebuilt in isolation, every
# (MB, NB, waves) the kernel can launch), A/B of the bf16 line with and without it, and - only if the line gains >= 1.5 % and the bf16 parity tests stay green on it -
# adoption: table into monorec_amd/tuned_b8.json, profile set r05_c5bf16 regenerated on the new plan stamp, lines.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s20
mkdir -p $OUT
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16"
timeout 500 python tools/tune_b8.py --emit $OUT/tuned_b8.json > $OUT/tune_b8.log 2>&1; echo "tune rc=$?"; tail -1 $OUT/tune_b8.log | cut -c1-300
val() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2))"; }
: > $OUT/ab.txt
for rep in 1 2; do
  a=$(timeout 200 python bench.py --steps 60 $C5 --no-cpu-baseline --no-primer --no-forward-api 2>/dev/null | val); echo "rule $a" | tee -a $OUT/ab.txt
  b=$(MR_TUNED_B8=$OUT/tuned_b8.json timeout 200 python bench.py --steps 60 $C5 --no-cpu-baseline --no-primer --no-forward-api 2>/dev/null | val); echo "table $b" | tee -a $OUT/ab.txt
done
ADOPT=$(python - <<'PY'
rows = [l.split() for l in open("gpurun_out/r05_s20/ab.txt")]
r = [float(v) for k, v in rows if k == "rule"]; t = [float(v) for k, v in rows if k == "table"]
gain = (sum(t) / len(t)) / (sum(r) / len(r)) - 1.0
print("yes" if gain >= 0.015 and min(t) > max(r) else "no", round(100 * gain, 2))
PY
)
echo "adopt: $ADOPT"
if [ "${ADOPT%% *}" = yes ]; then
  cp $OUT/tuned_b8.json monorec_amd/tuned_b8.json
  timeout 900 python -m pytest tests/test_gpu_b8.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -k "b8 or bf16 or c5" > $OUT/tests.log 2>&1; rc=$?; echo "bf16 tests rc=$rc"; tail -2 $OUT/tests.log | cut -c1-200
  if [ $rc -ne 0 ]; then
    rm -f monorec_amd/tuned_b8.json; echo "table withdrawn"
  else
    bash tools/profile_round.sh r05_c5bf16 "$C5" 20 > $OUT/profile_c5bf16.log 2>&1; echo "profile c5bf16 rc=$?"
    timeout 200 python bench.py --steps 60 $C5 --no-cpu-baseline --no-primer > $OUT/c5_bf16.json 2> /dev/null
    timeout 200 python bench.py --steps 60 $C5 --lean-outputs --no-cpu-baseline --no-primer --no-forward-api > $OUT/c5_bf16_lean.json 2> /dev/null
    tail -1 $OUT/c5_bf16.json > profiles/r05_c5bf16_line.json
    tail -1 $OUT/c5_bf16_lean.json > profiles/r05_c5bf16_lean_line.json
    mkdir -p $OUT/profiles && cp profiles/r05_c5bf16_* $OUT/profiles/ && cp monorec_amd/tuned_b8.json $OUT/profiles/tuned_b8.adopted.json
    python - <<'PY'
import json
for f in ("c5_bf16", "c5_bf16_lean"):
    d = json.loads(open(f"gpurun_out/r05_s20/{f}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "frac", round(r["frac"], 3), r.get("frac_source"), "stale" if "stale_profile" in r else "current")
PY
  fi
fi
