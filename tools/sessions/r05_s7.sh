#!/bin/bash
# Round 5, session 7: r05_s6 - where the model's four busy streams sit among ROCm's hardware queues (bound at first use, in order of first use) sets the
# two-in-flight rate: next to each other 757-769 keyframes/s, something between them 693-717, spread out 509-558.  The parse-only token is gone (705-709 vs
# 716-718 on 20-step lines).  Here: the default order once more against its neighbours, the number of hardware queues and a third keyframe in flight under a
# good order, and the c3 / configs[4] lines with it.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s7
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3))"; }
run() { MR_DIAG_STREAM_LAYOUT="$1" timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api $2 2>/dev/null | line "[$1] $2"; }
for rep in 1 2; do
  run "" ""
  run "g,m0,e0,m1,e1" ""
  run "m0,m1,e0,e1,g" ""
done
run "" "--hw-queues 4"
run "" "--hw-queues 8"
run "" "--hw-queues 32"
run "" "--in-flight 3"
run "" "--in-flight 1"
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "default, 20 steps"
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "default, 20 steps"
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "c3"
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api --cv-separable 2>/dev/null | line "c3 separable cost-volume sums"
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api 2>/dev/null | line "configs[4] bf16"
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --lean-outputs --no-cpu-baseline --no-primer --no-forward-api 2>/dev/null | line "configs[4] bf16 lean"
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -k "arenas or fixed_order or prepare_then" 2>&1 | tail -2
