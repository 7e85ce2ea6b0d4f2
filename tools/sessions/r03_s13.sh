#!/bin/bash
# 16-channel tail workgroups of the in-register-transform Winograd kernel (48-channel layers): parity, then per-layer times
OUT=gpurun_out/r03_s13; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "winograd" 2>&1 | grep -E "passed|failed" | tail -2
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd.json
for cfg in "" "--batch 8 --frames 4 --depths 64" "--height 512 --width 1024 --frames 4 --depths 48"; do
  echo "== $cfg"
  timeout 300 python tools/bench_wino.py $cfg --only mask --emit $OUT/tuned_winograd.json 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'name' in r:
        if r['cout'] % 32 and r['cout'] % 32 <= 16: print(r['name'], 'direct', r['direct_us'], {k[4:-3]:v for k,v in r.items() if k.startswith('wino') and k.endswith('_us')}, 'best', r['best'])
    else: print(r)"
done
