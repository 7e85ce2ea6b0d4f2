#!/bin/bash
# Profile one configuration on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03_c2                                      # c2 = bench.py defaults
#   bash tools/profile_round.sh r03_c3 "--batch 8 --frames 4 --depths 64"   # c3
# kernel-trace stats and the PMC counters are collected in separate rocprofv3 runs (counters never together with trace
# domains, one counter group per pass); the result databases are condensed into profiles/<tag>_* by tools/summarize_prof.py
# and deleted (gpurun_out stays small).  Every pass is the SAME command as the bench line, shortened.
TAG=${1:-r03_c2}
SHAPE=${2:-}
STEPS=${3:-40}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 600 python bench.py --steps 200 $SHAPE --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --steps $STEPS --no-cpu-baseline --no-primer --no-forward-api $SHAPE > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_seq -o t -- python $REPO/bench.py --steps $STEPS --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api $SHAPE > $OUT/trace_seq.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d $OUT/pmc$i -o p -- python $REPO/bench.py --steps 6 --warmup 2 --spinup-seconds 0 --no-cpu-baseline --no-primer --no-forward-api $SHAPE > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
done
cd $REPO
python tools/summarize_prof.py --tag $TAG --bench-line $OUT/bench.json --stats $(find $OUT/trace -name "*_results.db" | head -1) --stats-seq $(find $OUT/trace_seq -name "*_results.db" | head -1) --pmc $(find $OUT/pmc* -name "*_results.db") | tail -40
cp $OUT/layers.json profiles/${TAG}_layer_times.json
tail -1 $OUT/bench.json > profiles/${TAG}_bench.json
mkdir -p $REPO/gpurun_out/profiles_out && cp profiles/${TAG}_* $REPO/gpurun_out/profiles_out/
find $OUT -name "*.db" -delete
du -sh $OUT
