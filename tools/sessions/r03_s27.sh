#!/bin/bash
# Round 3, session 27 (what is left of the budget, <= 6 GPU-minutes): the committed c2 profile set once more with the table of session 26
# (Cook-Toom forms adopted): full bench line + per-layer times, pipelined kernel trace, six PMC passes; the summaries are rewritten after
# every pass so that a cut leaves the latest complete state.  (The one-keyframe-at-a-time trace is the one of session 26.)
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03_s27
mkdir -p $OUT
timeout 200 python bench.py --steps 200 --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
cp $OUT/layers.json profiles/r03_c2_layer_times.json && tail -1 $OUT/bench.json > profiles/r03_c2_bench.json
mkdir -p $OUT/profiles_out && cp profiles/r03_c2_* $OUT/profiles_out/
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --steps 40 --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace.log 2>&1
cd $REPO
DB=$(find $OUT/trace -name "*_results.db" | head -1)
[ -n "$DB" ] && [ -s "$DB" ] && python tools/summarize_prof.py --tag r03_c2 --stats $DB > /dev/null 2>&1 && cp profiles/r03_c2_kernel_stats.csv $OUT/profiles_out/
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  cd /tmp
  timeout 120 rocprofv3 --pmc $C -d $OUT/pmc$i -o p -- python $REPO/bench.py --steps 6 --warmup 2 --spinup-seconds 0 --no-cpu-baseline --no-primer --no-forward-api > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
  cd $REPO
  if [ $i -ge 2 ]; then
    python tools/summarize_prof.py --tag r03_c2 --pmc $(find $OUT/pmc* -name "*_results.db") > /dev/null 2>&1 && cp profiles/r03_c2_pmc_summary.json $OUT/profiles_out/ && echo "$i passes" > $OUT/profiles_out/pmc_passes.txt
  fi
done
find $OUT -name "*.db" -delete
# the line once more, now quoting the counters of THIS table
timeout 120 python bench.py --steps 200 --no-cpu-baseline --no-primer > $OUT/bench_after.json 2> $OUT/bench_after.err
tail -1 $OUT/bench_after.json | cut -c1-200
