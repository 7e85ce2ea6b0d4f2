#!/bin/bash
# Round 4, session 21: what bounds conv3x3_wino44_kernel at the c3 shape (mask.enc0.0: 32 images of 256x512, 64 -> 64 channels; 1469 us for
# 492 us of matrix work)?  PMC passes over tools/bench_wino.py --only mask.enc0.0 (counters per kernel name; one group per pass).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s21
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $OUT/pmc$i -o p -- python $REPO/tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
done
cd $REPO
python tools/summarize_prof.py --tag r04_w44probe --pmc $(find $OUT/pmc* -name "*_results.db") > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open("profiles/r04_w44probe_pmc_summary.json"))["per_kernel"]
for k, v in d.items():
    if "wino" in k:
        print(k, {cn: round(x["per_dispatch"], 1) for cn, x in sorted(v.items())})
PY
mkdir -p $OUT/profiles && mv profiles/r04_w44probe_* $OUT/profiles/
find $OUT -name "*.db" -delete
