#!/bin/bash
OUT=gpurun_out/r03_s18; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^E" $OUT/pytest.log | head -5
B="python bench.py --steps 300 --no-cpu-baseline --no-primer"
python bench.py --steps 50 --no-cpu-baseline --no-forward-api > /dev/null 2>&1
for cfg in "" "--host-mats" "" "--host-mats" "--in-flight 3" "--in-flight 3 --host-mats" "--in-flight 1"; do
  timeout 200 $B $cfg > $OUT/b.json 2>/dev/null; python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_s18/b.json").read().strip().splitlines()[-1]); fa=d.get("forward_api",{})
print("bench", sys.argv[1], round(d["value"],1), "ms", round(d["ms_per_step"],3), "host_enq", round(d["host_enqueue_ms"],3), "fwd", round(fa.get("value",0),1))
PY
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-160
