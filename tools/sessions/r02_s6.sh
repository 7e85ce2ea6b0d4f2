#!/bin/bash
OUT=gpurun_out/s6
mkdir -p $OUT
cp monorec_amd/tuned_schedules.json $OUT/table_A.json
b() { timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s', 'sum-of-kernels ms', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
b "table A (isolated, LDS <= 160K)"
b "table A in-flight 3" "--in-flight 3"
for CAP in 81920 54613; do
  timeout 900 python tools/tune_conv.py --lds-cap $CAP --out $OUT/table_cap$CAP.json > $OUT/tune_cap$CAP.log 2>&1; tail -1 $OUT/tune_cap$CAP.log
  python - $CAP <<'PY'
import json, sys
cap = sys.argv[1]
a = json.load(open("gpurun_out/s6/table_A.json")); c = json.load(open(f"gpurun_out/s6/table_cap{cap}.json"))
a.update(c); json.dump(a, open("monorec_amd/tuned_schedules.json", "w"), indent=0, sort_keys=True)
PY
  b "table LDS cap $CAP"
  b "table LDS cap $CAP in-flight 3" "--in-flight 3"
done
cp $OUT/table_A.json monorec_amd/tuned_schedules.json
