#!/bin/bash
# Hybrid tables: LDS-capped schedules (two keyframes' workgroups can share a CU) only for the launches that are short in isolation.
OUT=gpurun_out/s11
mkdir -p $OUT
cp monorec_amd/tuned_schedules.json $OUT/table_K.json
b() { timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s')"; }
b "table K"
for T in cap81920_thr15 cap81920_thr25 cap81920_thr40 cap54613_thr15 cap54613_thr25; do
  cp tools/experiments/hybrid_$T.json monorec_amd/tuned_schedules.json   # (tables were generated offline for this run; not kept)
  b "hybrid $T"
done
cp $OUT/table_K.json monorec_amd/tuned_schedules.json
b "table K again"
