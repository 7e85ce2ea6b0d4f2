#!/bin/bash
# FIRST hardware run of the 1-D Winograd F(2,3) kernel: parity, then per-layer times at c2 / c3 / configs[4] (tables emitted)
OUT=gpurun_out/r03_s16; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "1d or 3tap" 2>&1 | grep -E "passed|failed|^E" | tail -6
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd.json
for cfg in "" "--batch 8 --frames 4 --depths 64" "--height 512 --width 1024 --frames 4 --depths 48"; do
  echo "== $cfg"
  timeout 400 python tools/bench_wino1d.py $cfg --emit $OUT/tuned_winograd.json 2>$OUT/err.log | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'name' in r: print(r['name'], 'direct', r['direct_us'], {k[4:-3]:v for k,v in r.items() if k.startswith('wino') and k.endswith('_us')}, 'best', r['best'], 'maxdiff', max([v for k,v in r.items() if k.endswith('maxdiff')] or [0]))
    else: print(r)"
  tail -2 $OUT/err.log
done
