#!/bin/bash
# PROPOSED first hardware session of round 4 (not run yet): what round 3 could not measure any more.  ~8 GPU-minutes on the fast boxes.
#   1. the whole gpu suite on the final library / table of round 3 (the last whole-suite run, r03_s26.sh, predates the F(4x4,3x3) kernel);
#   2. F(4x4,3x3) and the Cook-Toom forms on the c3 and configs[4] shapes (tables), then c3 / configs[4] lines;
#   3. where forward() loses 0.27 ms per call to the --in-flight 1 submit loop (kernel + memory-copy timeline of a few forwards);
#   4. the c2 / c3 profile sets with whatever table results.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s1
mkdir -p $OUT
T=monorec_amd/tuned_winograd.json
timeout 420 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -2 $OUT/suite.log
cp $T $OUT/table_before.json
timeout 300 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --min-pixels 32768 --emit $T > $OUT/wino_c3.log 2>&1; tail -1 $OUT/wino_c3.log
timeout 300 python tools/bench_wino.py --height 512 --width 1024 --frames 4 --depths 48 --min-pixels 131072 --emit $T > $OUT/wino_c5.log 2>&1; tail -1 $OUT/wino_c5.log
timeout 200 python tools/bench_wino1d.py --height 512 --width 1024 --frames 4 --depths 48 --no-upconv --emit $T > $OUT/wino1d_c5.log 2>&1; tail -1 $OUT/wino1d_c5.log
cp $T $OUT/table_after.json
timeout 200 python -m pytest tests/test_gpu_model.py -q -x -k "c3_full_shape or c5_shape" > $OUT/model_c3_c5.log 2>&1; echo "c3 / c5 model tests with the new table rc=$?"
timeout 150 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer > $OUT/c3_line.json 2> $OUT/c3.err; tail -1 $OUT/c3_line.json | cut -c1-150
timeout 150 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --no-cpu-baseline --no-primer > $OUT/c5_line.json 2> $OUT/c5.err; tail -1 $OUT/c5_line.json | cut -c1-150
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/fwd_trace -o t -- python $REPO/tools/trace_forward.py --in-flight 2 --steps 30 > $OUT/fwd_trace.log 2>&1
cd $REPO
python tools/timeline_overlap.py $(find $OUT/fwd_trace -name "*kernel_trace.csv" | head -1) > $OUT/fwd_timeline.txt 2>&1
bash tools/profile_round.sh r04_c2 > $OUT/profile_c2.log 2>&1
bash tools/profile_round.sh r04_c3 "--batch 8 --frames 4 --depths 64" 20 > $OUT/profile_c3.log 2>&1
