#!/bin/bash
# host-gated submit (nothing enqueued behind an unsatisfied dependency), results collected by a host wait before the submit that reuses the slot
OUT=gpurun_out/r03_s9; mkdir -p $OUT
for cfg in "--in-flight 1" "--in-flight 2" "--in-flight 2 --host-mats" "--in-flight 1 --host-mats"; do
  echo "cfg: $cfg"; python tools/trace_forward.py $cfg 2>&1 | grep "forward()" | cut -c1-150
done
B="python bench.py --steps 300 --no-cpu-baseline --no-primer"
run() { timeout 200 $B "$@" > $OUT/b.json 2>$OUT/b.err; python - "$*" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r03_s9/b.json").read().strip().splitlines()[-1]); fa=d.get("forward_api",{})
    print("bench", sys.argv[1], round(d["value"],1), "host_enq", round(d["host_enqueue_ms"],3), "forward_api", round(fa.get("value",0),1))
except Exception as e: print("bench", sys.argv[1], "FAILED", e)
PY
}
run
run --host-mats
run --stream-collect
run --stream-collect --host-mats
run --in-flight 3
run --in-flight 3 --host-mats
run --in-flight 1
run --queue-depth 2 --host-mats
run --hw-queues 4 --host-mats
run --host-mats
run
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
