#!/bin/bash
# Round 5, session 5: r05_s4 found the 8 % - the creation ORDER of the model's streams (a parse-only token on an idle device created the slot streams before
# the gather stream: 694-710 keyframes/s; round 4's order: 742-762).  The streams are now created at one point in a fixed order; which order is best?
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s5
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3))"; }
run() { MR_DIAG_STREAM_LAYOUT="$2" timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "$1 [$2]"; }
for rep in 1 2; do
  run "A default" ""
  MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "A default, no parse-only token"
  run B "m0,e0,_,g,m1,e1,_"
  run C "m0,e0,m1,e1,g"
  run D "m0,m1,e0,e1,g"
  run E "g,m0,_,_,_,m1,_,_,_,e0,_,_,_,e1"
  run F "m0,_,m1,_,e0,_,e1,_,g"
  run G "g,_,_,_,m0,e0,m1,e1"
  run H "g,m0,e0,m1,e1"
  run I "g,m0,m1,e0,e1"
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "A default, 20 steps"
MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "A default, 20 steps, no parse-only token"
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "A default, 20 steps"
MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "A default, 20 steps, no parse-only token"
