#!/bin/bash
# FIRST hardware run of the 4-multiply Upconv kernel: parity, per-layer times (tables emitted), end to end
OUT=gpurun_out/r03_s21; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "upconv" 2>&1 | grep -E "passed|failed|^E" | tail -4
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd.json
for cfg in "" "--batch 8 --frames 4 --depths 64" "--height 512 --width 1024 --frames 4 --depths 48"; do
  echo "== $cfg"
  timeout 400 python tools/bench_wino1d.py $cfg --emit $OUT/tuned_winograd.json 2>$OUT/err.log | grep -E "mask.dec|upconv" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'name' in r: print(r['name'], 'direct', r['direct_us'], {k[4:-3]:v for k,v in r.items() if k.startswith('wino') and k.endswith('_us')}, 'best', r['best'], 'maxdiff', max([v for k,v in r.items() if k.endswith('maxdiff')] or [0]))
    else: print(r)"
done
cp $OUT/tuned_winograd.json monorec_amd/tuned_winograd.json
B="python bench.py --steps 300 --no-cpu-baseline --no-primer --no-forward-api"
for cfg in "" "--host-mats" "--batch 8 --frames 4 --depths 64 --steps 60"; do
  timeout 200 $B $cfg > $OUT/b.json 2>/dev/null; python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_s21/b.json").read().strip().splitlines()[-1])
print("bench", sys.argv[1], round(d["value"],1), "ms", round(d["ms_per_step"],3))
PY
done
