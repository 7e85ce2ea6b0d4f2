#!/bin/bash
# Final state of round 2: full gpu suite, smoke, driver-style bench, c2 / c3 profiles (concurrent + one-keyframe-at-a-time traces, PMC).
OUT=gpurun_out/s7
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-330 $OUT/bench_driver.json
bash tools/profile_round.sh r02_c2
bash tools/profile_round.sh r02_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver2.json 2> $OUT/bench_driver2.err; python -c "
import json; d=json.loads(open('$OUT/bench_driver2.json').read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline','cost_volume_kernel','cpu_baseline')}, indent=0)[:3000])"
