#!/bin/bash
# Round 5, session 9: r05_s8 - ONE stream per slot with THREE keyframes in flight: 791.5 keyframes/s at c2 (two streams per slot, two in flight: 765).  How far
# does that go (in-flight 3 / 4 / 5), and what does it do to a 20-step line (more keyframes to fill and drain)?
cd "$(dirname "$0")/../.." || exit 1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'host cpu', round(d['host_cpu_ms_per_keyframe'],2))"; }
run() { timeout 200 python bench.py --no-primer --no-cpu-baseline --no-forward-api $1 2>/dev/null | line "$1"; }
for rep in 1 2; do
  run "--steps 200"
  run "--steps 200 --single-stream --in-flight 3"
  run "--steps 200 --single-stream --in-flight 4"
done
run "--steps 200 --single-stream --in-flight 5"
run "--steps 200 --in-flight 3"
run "--steps 200 --in-flight 4"
for rep in 1 2; do
  run "--steps 20 --warmup 5"
  run "--steps 20 --warmup 5 --single-stream --in-flight 3"
  run "--steps 20 --warmup 5 --single-stream --in-flight 4"
done
run "--steps 40 --batch 8 --frames 4 --depths 64"
run "--steps 40 --batch 8 --frames 4 --depths 64 --single-stream --in-flight 3"
run "--steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16"
run "--steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --single-stream --in-flight 3"
