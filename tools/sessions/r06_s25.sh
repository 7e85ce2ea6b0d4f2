#!/bin/bash
# Round 6, session 25: does a spinning host wait (hipDeviceScheduleSpin / ROC_ACTIVE_WAIT_TIMEOUT) take back what the idle host behind the closing
# synchronize() of the warm-up costs the 20-step line (value_host_primed 798 against value 763)?
cd "$(dirname "$0")/../.." || exit 1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s')"; }
Q="--no-cpu-baseline --no-forward-api --no-secondary --steps 20 --warmup 5"
for rep in 1 2 3; do
  timeout 300 python bench.py $Q 2>/dev/null | line "default:"
  ROC_ACTIVE_WAIT_TIMEOUT=3000 timeout 300 python bench.py $Q 2>/dev/null | line "ROC_ACTIVE_WAIT_TIMEOUT=3000:"
  MR_DIAG_SPIN=1 timeout 300 python bench.py $Q 2>/dev/null | line "hipDeviceScheduleSpin:"
  timeout 300 python bench.py $Q --host-prime-ms 3 2>/dev/null | line "host-primed 3 ms:"
done
