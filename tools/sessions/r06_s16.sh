#!/bin/bash
# Round 6, session 16: the 20-step regime - eager launches vs hipGraph replay per stage (host enqueue 0.4 ms vs ~0.05 ms per keyframe: does the pipeline fill faster?)
cd "$(dirname "$0")/../.." || exit 1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'host cpu ms', round(d['host_cpu_ms_per_keyframe'],2))"; }
Q="--no-cpu-baseline --no-forward-api --no-secondary"
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | line "eager 20:"
  timeout 300 python bench.py --steps 20 --warmup 5 $Q --graph 2>/dev/null | line "graph 20:"
done
timeout 300 python bench.py --steps 200 $Q 2>/dev/null | line "eager 200:"
timeout 300 python bench.py --steps 200 $Q --graph 2>/dev/null | line "graph 200:"
