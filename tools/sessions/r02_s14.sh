#!/bin/bash
# Selective LDS-capped table (only where the capped schedule is within 4 % in isolation), interleaved A/B, 3 rounds.
OUT=gpurun_out/s14
mkdir -p $OUT
cp monorec_amd/tuned_schedules.json $OUT/table_K.json
b() { timeout 300 python bench.py --steps 400 --no-cpu-baseline --no-primer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s')"; }
for r in 1 2 3; do
  cp $OUT/table_K.json monorec_amd/tuned_schedules.json; b "table K"
  cp tools/experiments/hybrid_slow4.json monorec_amd/tuned_schedules.json; b "selective cap"
done
cp $OUT/table_K.json monorec_amd/tuned_schedules.json
