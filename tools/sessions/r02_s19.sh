#!/bin/bash
# State after the one-channel kernels / one-plane marching / DataParallel fix: full gpu suite, smoke, driver-style line, c3 and
# configs[4]-shape bench lines.
OUT=gpurun_out/s19
mkdir -p $OUT
S=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
S=$(date +%s)
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "driver-style bench: $(( $(date +%s) - S )) s wall"; cut -c1-260 $OUT/bench_driver.json
timeout 300 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --dump-layers $OUT/layers_c3.json > $OUT/bench_c3.json 2>/dev/null; cut -c1-260 $OUT/bench_c3.json
timeout 300 python bench.py --steps 100 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline > $OUT/bench_c5_f32.json 2>/dev/null; cut -c1-200 $OUT/bench_c5_f32.json
timeout 300 python bench.py --steps 100 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline --bf16 > $OUT/bench_c5_bf16.json 2>/dev/null; cut -c1-200 $OUT/bench_c5_bf16.json
