#!/bin/bash
# State with the Winograd kernel in the plan: full gpu suite, smoke, driver-style line, c3 / configs[4]-shape lines, then the profiles.
OUT=gpurun_out/s29
mkdir -p $OUT
S=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-200 $OUT/bench_driver.json
timeout 300 python bench.py --steps 100 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline > $OUT/bench_c5_f32.json 2>/dev/null; cut -c1-200 $OUT/bench_c5_f32.json
bash tools/profile_round.sh r02_c2 > $OUT/prof_c2.log 2>&1; tail -3 $OUT/prof_c2.log | cut -c1-200
bash tools/profile_round.sh r02_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12 > $OUT/prof_c3.log 2>&1; tail -3 $OUT/prof_c3.log | cut -c1-200
