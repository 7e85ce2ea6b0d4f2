#!/bin/bash
# Round 4, session 16: any number of depth hypotheses (odd counts included) through the cost-volume kernels and the model; the kernel and
# model suites on the library with that change; the dynamic-batching test on the tables with batch-2 / batch-4 entries.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r04_s16
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cost_volume" > $OUT/cv.log 2>&1; echo "cost-volume kernel tests rc=$?"; tail -3 $OUT/cv.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "any_number or dynamic_batching or cv_depths or forward_matches or patch_size or without_mult_mask" > $OUT/model.log 2>&1; echo "model tests rc=$?"; tail -3 $OUT/model.log | cut -c1-400
timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c2.json 2> $OUT/c2.err; python -c "
import json; d=json.loads(open('$OUT/c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value'],1), d['roofline'].get('profile_stamp'), d['roofline'].get('stale_profile'))"
