#!/bin/bash
# Round 6, session 30: the first prepare() behind an idle device costs 0.5 ms (r06_s27).  Round 5 measured a parse-only token for that case with two streams per slot and
# dropped it (705-709 against 716-718); with one stream per slot the order of a keyframe's stages no longer matters for its latency - measured again (MR_DIAG_LAZY_FIRST=1).
cd "$(dirname "$0")/../.." || exit 1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s', d.get('step_marks_ms', [])[:5])"; }
Q="--no-cpu-baseline --no-forward-api --no-secondary --steps 20 --warmup 5 --step-times"
for rep in 1 2 3 4; do
  timeout 300 python bench.py $Q 2>/dev/null | line "default:"
  MR_DIAG_LAZY_FIRST=1 timeout 300 python bench.py $Q 2>/dev/null | line "lazy first token:"
done
MR_DIAG_LAZY_FIRST=1 timeout 300 python bench.py --no-cpu-baseline --no-forward-api --no-secondary --steps 200 2>/dev/null | line "lazy first token, 200 steps:"
