#!/bin/bash
# Round 5, session 4: what separates this tree (695-710 keyframes/s, c2, 200 steps, two in flight) from the round-4 tree (742-756) on the same box?
# r05_s3 ruled out the event queries of prepare()'s idle test.  Candidates left: the warm-up tail (round 4 timed after 3 ms of prepare() calls, this tree
# does not), the parse-only token of the first request on an idle device (encoder stage first instead of cost volume first: does the start set a
# persistent completion pattern - pairs vs staggered?).
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s4
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('step_parts_ms_prepare_wait_submit') or []
print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'waits of steps 10-17:', [x[1] for x in p[10:18]])"; }
for rep in 1 2; do
  (cd ab_r4 && timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --step-times 2>/dev/null | line "r4-tree primed(default)")
  (cd ab_r4 && timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --step-times --host-prime-ms 0 2>/dev/null | line "r4-tree unprimed")
  timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --step-times 2>/dev/null | line "r5-tree unprimed(default)"
  timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --step-times --host-prime-ms 3 2>/dev/null | line "r5-tree primed"
  MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --step-times 2>/dev/null | line "r5-tree unprimed, no lazy prepare"
  MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --step-times --host-prime-ms 3 2>/dev/null | line "r5-tree primed, no lazy prepare"
done
