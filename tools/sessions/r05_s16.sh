#!/bin/bash
# Round 5, session 16: the 3x3 stride-1 tables once more after the LDS pitch changes (F(2x2,3x3) in registers -7.5 %, F(4x4,3x3) -2 %: r05_s14): every
# candidate kernel per layer at c2 and c3, choices merged into a COPY of the table (reviewed before it replaces monorec_amd/tuned_winograd.json).
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s16
mkdir -p $OUT
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd_new.json
timeout 500 python tools/bench_wino.py --codes 1,2,11,12,21,31 --emit $OUT/tuned_winograd_new.json > $OUT/wino_c2.log 2>&1; echo "c2 rc=$?"; tail -1 $OUT/wino_c2.log
timeout 700 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --codes 1,2,11,12,21,31 --emit $OUT/tuned_winograd_new.json > $OUT/wino_c3.log 2>&1; echo "c3 rc=$?"; tail -1 $OUT/wino_c3.log
python - <<'PY'
import json
old = json.load(open("monorec_amd/tuned_winograd.json")); new = json.load(open("gpurun_out/r05_s16/tuned_winograd_new.json"))
rows = {}
for f in ("wino_c2", "wino_c3"):
    for line in open(f"gpurun_out/r05_s16/{f}.log"):
        if line.startswith("{") and '"sig"' in line:
            r = json.loads(line); rows[r["sig"]] = r
for k in sorted(new):
    if old.get(k) != new[k]:
        r = rows.get(k, {})
        print(k, old.get(k), "->", new[k], r.get("name"), "direct", r.get("direct_us"), {c: r.get(f"wino{c}_us") for c in (1, 2, 11, 12, 21, 31) if f"wino{c}_us" in r})
PY
