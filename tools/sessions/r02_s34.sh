#!/bin/bash
# Winograd kernel with the input transform in registers (no V buffer, one barrier per chunk): parity, then times next to the V-buffer kernel.
MR_WINO_REGB=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "winograd_conv or winograd_bad" 2>&1 | tail -2
for v in 0 1; do
  if [ $v = 1 ]; then export MR_WINO_REGB=1; echo "--- transform in registers"; else unset MR_WINO_REGB; echo "--- V buffer in LDS"; fi
  timeout 200 python tools/bench_wino.py --only mask 2>/dev/null | grep -E "enc0.0|enc1.1|dec2.1|dec3.1|layers" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print({k:r[k] for k in r if k in ('name','direct_us','wino1_us','wino2_us','wino1_maxdiff','best_of_both_total_us','direct_total_us')})"
  timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 2>/dev/null | grep -E "enc0.0" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('c3', {k:r[k] for k in r if k in ('name','direct_us','wino1_us','wino2_us')})"
done
