#!/bin/bash
# Round 3, session 26 (last of the round, <= 9 GPU-minutes): FIRST hardware run of the Cook-Toom forms F(4,3) / F(2,7) / F(4,7)
# (csrc/conv1d_wino.hip conv1d_ct_kernel, numerics gate: oracle/numerics_study_winograd.py), per-layer timing next to the direct kernel and
# F(2,3), A/B of the whole c2 bench with the old and the new table, then the whole gpu suite + the driver-style line with the table that won.
# Every step has its own timeout; the decision (which table stays) is made here and written to gpurun_out/r03_s26/decision.txt.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03_s26
mkdir -p $OUT
T=monorec_amd/tuned_winograd.json
cp $T $OUT/table_old.json
cp $T $OUT/table_new.json

timeout 240 python -m pytest tests/test_gpu_kernels.py -q -x -k "cooktoom or winograd_1d" > $OUT/t_new.log 2>&1
NEW_RC=$?
tail -3 $OUT/t_new.log
echo "new kernel tests rc=$NEW_RC" | tee $OUT/decision.txt

ADOPT=0
if [ $NEW_RC -eq 0 ]; then
  timeout 150 python tools/bench_wino1d.py --no-upconv --emit $OUT/table_new.json > $OUT/wino1d_c2.log 2>&1
  tail -1 $OUT/wino1d_c2.log
  timeout 150 python tools/bench_wino1d.py --batch 8 --frames 4 --depths 64 --no-upconv --emit $OUT/table_new.json > $OUT/wino1d_c3.log 2>&1
  tail -1 $OUT/wino1d_c3.log
  timeout 120 python bench.py --steps 200 --no-cpu-baseline --no-primer --no-forward-api > $OUT/A_old.json 2> $OUT/A_old.err
  cp $OUT/table_new.json $T
  timeout 120 python bench.py --steps 200 --no-cpu-baseline --no-primer --no-forward-api > $OUT/B_new.json 2> $OUT/B_new.err
  ADOPT=$(python - <<PY
import json
def val(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])["value"]
    except Exception:
        return 0.0
a, b = val("$OUT/A_old.json"), val("$OUT/B_new.json")
open("$OUT/decision.txt", "a").write(f"c2 200 steps: old table {a:.1f}, new table {b:.1f} keyframes/s\n")
print(1 if b > 0 and b >= 0.998 * a else 0)
PY
)
fi
if [ "$ADOPT" = "1" ]; then
  echo "adopted: new table" | tee -a $OUT/decision.txt
else
  cp $OUT/table_old.json $T
  echo "kept: old table" | tee -a $OUT/decision.txt
fi
cp $T $OUT/table_final.json

if [ $NEW_RC -eq 0 ]; then
  timeout 420 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1
else
  timeout 420 python -m pytest tests -m gpu -x -q -k "not cooktoom" > $OUT/suite.log 2>&1
fi
echo "suite rc=$?" | tee -a $OUT/decision.txt
tail -3 $OUT/suite.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" | tee -a $OUT/decision.txt
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err
tail -1 $OUT/driver_style.json | cut -c1-160
timeout 120 python bench.py --steps 40 --warmup 5 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer --no-forward-api > $OUT/c3_line.json 2> $OUT/c3.err
tail -1 $OUT/c3_line.json | cut -c1-160
# one-keyframe-at-a-time kernel trace of the final table (kernel-only roofline of the bench line), if the budget still allows
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_seq -o t -- python $REPO/bench.py --steps 40 --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace_seq.log 2>&1
cd $REPO
DB=$(find $OUT/trace_seq -name "*_results.db" | head -1)
if [ -n "$DB" ] && [ -s "$DB" ]; then
  python tools/summarize_prof.py --tag r03s26_c2 --stats-seq $DB > /dev/null 2>&1
  cp profiles/r03s26_c2_kernel_stats_seq.csv $OUT/ 2>/dev/null
  find $OUT -name "*.db" -delete
fi
cat $OUT/decision.txt
