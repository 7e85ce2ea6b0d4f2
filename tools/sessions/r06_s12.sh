#!/bin/bash
# Round 6, session 12: SQ counters of the two F(4x4,3x3) kernels (8 waves vs one wave per SIMD) on the c3 layers.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s12
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o p -- python $REPO/tools/bench_wino.py --batch 8 --frames 4 --depths 64 --codes 31,51 --min-pixels 100000 > $OUT/pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r06_s12/pmc*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/r06_s12/pmc*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wino44" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-40:] + " grid=" + r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    print(k)
    print("   ", {c: round(v / cnt[(k, c)]) for c, v in sorted(d.items())})
PY
find $OUT -name "*.db" -delete
