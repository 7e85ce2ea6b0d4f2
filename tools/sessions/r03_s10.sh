#!/bin/bash
OUT=gpurun_out/r03_s10; mkdir -p $OUT
B="python bench.py --steps 300 --no-cpu-baseline --no-primer --no-forward-api"
run() { timeout 200 $B "$@" > $OUT/b.json 2>$OUT/b.err; python - "$*" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r03_s10/b.json").read().strip().splitlines()[-1])
    print("bench", sys.argv[1], round(d["value"],1), "host_enq", round(d["host_enqueue_ms"],3))
except Exception as e: print("bench", sys.argv[1], "FAILED", e)
PY
}
python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1
for r in 1 2 3; do
run
run --host-mats
done
run --stream-collect
run --in-flight 3
run --in-flight 3 --host-mats
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
