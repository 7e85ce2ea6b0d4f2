#!/bin/bash
# Round 6, session 27: where do the 20 timed steps behind a synchronize() lose 9 % against the same steps behind 3 ms of prepare() calls?  Per-step host
# stamps (prepare / wait / submit) of both regimes.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06_s27
Q="--no-cpu-baseline --no-forward-api --no-secondary --steps 20 --warmup 5 --step-times"
for rep in 1 2; do
for P in 0 3; do
  timeout 300 python bench.py $Q --host-prime-ms $P 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prime $P value', round(d['value'],1), 'elapsed', d['elapsed_ms'])
print('  marks', d['step_marks_ms'])
print('  parts', d['step_parts_ms_prepare_wait_submit'])"
done
done | tee gpurun_out/r06_s27/steps.txt
