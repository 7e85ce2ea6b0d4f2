#!/bin/bash
# transposed Winograd with the transform in registers: parity, per-layer times at c2 / c3 / configs[4] (tables emitted)
OUT=gpurun_out/r03_s11; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "transposed or refine" 2>&1 | grep -E "passed|failed" | tail -2
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd.json
for cfg in "" "--batch 8 --frames 4 --depths 64" "--height 512 --width 1024 --frames 4 --depths 48"; do
  echo "== $cfg"
  timeout 300 python tools/bench_wino_t.py $cfg --emit $OUT/tuned_winograd.json 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'name' in r: print(r['name'], 'direct', r['direct_us'], {k[4:-3]:v for k,v in r.items() if k.startswith('wino') and k.endswith('_us')}, 'best', r['best'], 'maxdiff', max(v for k,v in r.items() if k.endswith('maxdiff')))
    else: print(r)"
done
