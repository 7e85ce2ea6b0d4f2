#!/bin/bash
# Profile one round on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r01c_c2
# kernel-trace stats and the PMC counters are collected in separate rocprofv3 runs (counters never together with
# trace domains); the result databases are condensed into profiles/<tag>_* and deleted (gpurun_out stays small).
TAG=${1:-r01c_c2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py --steps 300 --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --steps 40 --no-cpu-baseline > $OUT/trace.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C -d $OUT/pmc$i -o p -- python $REPO/bench.py --steps 10 --warmup 2 --spinup-seconds 0 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
cd $REPO
python tools/summarize_prof.py --tag $TAG --stats $(find $OUT/trace -name "*_results.db" | head -1) --pmc $(find $OUT/pmc* -name "*_results.db")
cp $OUT/layers.json profiles/${TAG}_layer_times.json
tail -1 $OUT/bench.json > profiles/${TAG}_bench.json
cp -r profiles $REPO/gpurun_out/profiles_out
find $OUT -name "*.db" -delete
du -sh $OUT
