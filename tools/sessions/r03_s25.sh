#!/bin/bash
# final state (native launch lists per stage, 4 finishing launches fewer): gpu suite, smoke, c2 profile, bench lines
OUT=gpurun_out/r03_s25; mkdir -p $OUT
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/profile_round.sh r03_c2 > $OUT/prof_c2.log 2>&1; tail -2 $OUT/prof_c2.log | cut -c1-120
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-160 $OUT/bench_driver.json
timeout 400 python bench.py --steps 200 --no-primer > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-160 $OUT/bench_c2.json
timeout 300 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer > $OUT/bench_c3.json 2>/dev/null; cut -c1-160 $OUT/bench_c3.json
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --in-flight 1 > $OUT/bench_c2_if1.json 2>/dev/null; cut -c1-160 $OUT/bench_c2_if1.json
for f in r03_c2_bench.json r03_c2_kernel_stats.csv r03_c2_kernel_stats_seq.csv r03_c2_layer_times.json r03_c2_pmc_summary.json; do cp profiles/$f $OUT/; done
