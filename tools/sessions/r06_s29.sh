#!/bin/bash
# Round 6, session 29: spread of the driver command on one box - `python bench.py --gpus 1 --steps 20 --warmup 5`, six runs, full line each (primer, CPU baseline, secondaries).
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06_s29
for i in 1 2 3 4 5 6; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_s29/run$i.json
  python -c "
import json; d=json.load(open('gpurun_out/r06_s29/run$i.json')); print('run $i', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), '200:', round(d['value_200_steps'],1), 'forward', round(d['forward_api']['value'],1), 'frac', round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value'],3))"
done | tee gpurun_out/r06_s29/runs.txt
