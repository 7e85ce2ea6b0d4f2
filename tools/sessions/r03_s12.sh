#!/bin/bash
# interim state of round 3: gpu suite, smoke, driver-style line, two-rank control flow on one device, c2 profile (r03_c2_*)
OUT=gpurun_out/r03_s12; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-260 $OUT/bench_driver.json
MR_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; python -c "
import json;d=json.loads(open('$OUT/bench_2rank.json').read().strip().splitlines()[-1]);print('2 ranks on one device:', d['value'], d['ranks_seen'], d['per_rank_keyframes_per_s'])"
bash tools/profile_round.sh r03_c2 > $OUT/prof_c2.log 2>&1; tail -4 $OUT/prof_c2.log | cut -c1-250
