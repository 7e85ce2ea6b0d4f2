#!/bin/bash
# Round 6, session 10: conv_b8_kernel with buffer-descriptor stores (lane offsets once per workgroup), v_permlane16_swap exchange, incremental tile cursors.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s10
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_b8.py -m gpu -q -x -p no:cacheprovider > $OUT/b8_tests.log 2>&1; echo "b8 tests rc=$?"; tail -4 $OUT/b8_tests.log | cut -c1-300
timeout 200 python tools/bench_b8.py --layer enc0.1 --scheds 3,4,8 3,2,4 3,2,8 3,2,8,4 2>/dev/null | grep sched
timeout 200 python tools/bench_b8.py --layer dec3.1 --scheds 3,4,8 3,2,8 2>/dev/null | grep sched
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python $REPO/tools/bench_b8.py --layer enc0.1 --scheds 3,4,8 --reps 5 > $OUT/pmc1.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r06_s10/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "conv_b8" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: round(v / cnt[(k, c)]) for c, v in sorted(d.items())})
PY
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16"
for rep in 1 2; do
timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_c5.json 2>/dev/null | line "c5 bf16, round-5 table:"
done
