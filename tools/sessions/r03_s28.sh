#!/bin/bash
# Round 3, session 28 (the rest of the budget): the committed bench lines once more with the final table AND the final profiles in place
# (their frac_kernel_only / traffic quote profiles/r03_c2_*), and the one-batch-at-a-time kernel trace of c3 with the final table.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03_s28
mkdir -p $OUT
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err
tail -1 $OUT/driver_style.json | cut -c1-150
timeout 150 python bench.py --steps 200 > $OUT/c2_200.json 2> $OUT/c2_200.err
tail -1 $OUT/c2_200.json | cut -c1-150
timeout 100 python bench.py --steps 200 --in-flight 1 --no-cpu-baseline --no-primer --no-forward-api > $OUT/c2_inflight1.json 2> $OUT/c2_inflight1.err
tail -1 $OUT/c2_inflight1.json | cut -c1-150
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_seq_c3 -o t -- python $REPO/bench.py --steps 20 --batch 8 --frames 4 --depths 64 --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace_seq_c3.log 2>&1
cd $REPO
DB=$(find $OUT/trace_seq_c3 -name "*_results.db" | head -1)
if [ -n "$DB" ] && [ -s "$DB" ]; then
  python tools/summarize_prof.py --tag r03_c3 --stats-seq $DB > /dev/null 2>&1
  cp profiles/r03_c3_kernel_stats_seq.csv $OUT/
fi
find $OUT -name "*.db" -delete
timeout 150 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer > $OUT/c3_line.json 2> $OUT/c3.err
tail -1 $OUT/c3_line.json | cut -c1-150
