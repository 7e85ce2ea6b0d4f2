#!/bin/bash
OUT=gpurun_out/r03_s17; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
B="python bench.py --steps 300 --no-cpu-baseline --no-primer --no-forward-api"
python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1
for cfg in "" "--host-mats" "" "--host-mats" "--batch 8 --frames 4 --depths 64 --steps 60" "--height 512 --width 1024 --frames 4 --depths 48 --steps 100"; do
  timeout 200 $B $cfg > $OUT/b.json 2>/dev/null; python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_s17/b.json").read().strip().splitlines()[-1])
print("bench", sys.argv[1], round(d["value"],1), "ms", round(d["ms_per_step"],3), "sumk", round(d["device_ms_per_step_sum_of_kernels"],3), "frac", round(d["roofline"]["frac"],3))
PY
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-200
