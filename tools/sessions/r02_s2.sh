#!/bin/bash
# Round-2 second hardware session: VALU issue rates, real-sample validity diagnosis, all gpu tests, schedules for the new
# phase-decomposed Upconv launches, driver-style bench, c2 and c3 profiles.
OUT=gpurun_out/s2
mkdir -p $OUT
tools/probes/valu_rates > $OUT/valu_rates.txt 2>&1; cat $OUT/valu_rates.txt
timeout 300 python tests/diagnostics/diag_kitti_flips.py > $OUT/diag_flips.txt 2>&1; tail -30 $OUT/diag_flips.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 600 python tools/tune_conv.py --merge --missing > $OUT/tune_c2.log 2>&1; tail -6 $OUT/tune_c2.log
timeout 600 python tools/tune_conv.py --merge --missing --batch 8 --frames 4 --depths 64 > $OUT/tune_c3.log 2>&1; tail -6 $OUT/tune_c3.log
timeout 600 python tools/tune_conv.py --merge --missing --height 512 --width 1024 --frames 4 --depths 48 > $OUT/tune_c5.log 2>&1; tail -6 $OUT/tune_c5.log
cp monorec_amd/tuned_schedules.json $OUT/tuned_schedules.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-330 $OUT/bench_driver.json
bash tools/profile_round.sh r02_c2
bash tools/profile_round.sh r02_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12
