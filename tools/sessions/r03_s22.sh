#!/bin/bash
# final state of round 3: gpu suite, smoke, profiles of c2 / c3 (copied into profiles/ before the bench lines so that the lines quote them),
# driver-style line, 200-step line, c3, configs[4] shape in fp32 / bf16, c2 in bf16
OUT=gpurun_out/r03_s22; mkdir -p $OUT
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/profile_round.sh r03_c2 > $OUT/prof_c2.log 2>&1
bash tools/profile_round.sh r03_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12 > $OUT/prof_c3.log 2>&1
ls profiles | grep r03
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-200 $OUT/bench_driver.json
timeout 400 python bench.py --steps 200 --no-primer > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-200 $OUT/bench_c2.json
timeout 300 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer > $OUT/bench_c3.json 2>/dev/null; cut -c1-200 $OUT/bench_c3.json
timeout 300 python bench.py --steps 100 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline --no-primer > $OUT/bench_c5_f32.json 2>/dev/null; cut -c1-200 $OUT/bench_c5_f32.json
timeout 300 python bench.py --steps 100 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline --no-primer --bf16 > $OUT/bench_c5_bf16.json 2>/dev/null; cut -c1-200 $OUT/bench_c5_bf16.json
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --bf16 > $OUT/bench_c2_bf16.json 2>/dev/null; cut -c1-200 $OUT/bench_c2_bf16.json
timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-primer --host-mats > $OUT/bench_c2_hostmats.json 2>/dev/null; cut -c1-200 $OUT/bench_c2_hostmats.json
cp profiles/r03_* $OUT/
