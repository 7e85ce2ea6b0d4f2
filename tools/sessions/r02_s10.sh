#!/bin/bash
OUT=gpurun_out/s10
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for B in 2 4 8; do
  timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-primer --batch $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 shape, batch $B per launch:', round(d['value'],1), 'kf/s', 'conv TF', round(d['roofline']['achieved'],1))"
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-300 $OUT/bench_driver.json
