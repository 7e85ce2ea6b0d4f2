#!/bin/bash
# Round 6, session 9: SQ counters of conv_b8_kernel on mask.enc0.1 / dec3.1 (product library): where do the waves wait?
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o p -- python $REPO/tools/bench_b8.py --layer enc0.1 --scheds 3,4,8 3,2,4 --reps 5 > $OUT/pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r06_s9/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "")
        if "conv_b8" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / cnt[(k, c)]:16.0f}  per dispatch")
PY
find $OUT -name "*.db" -delete
