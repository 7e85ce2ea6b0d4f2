#!/bin/bash
# Round 3, session 30 (the last ~2 GPU-minutes): FIRST hardware run of the F(4x4,3x3) kernel (csrc/conv_wino44.hip; numerics gate and a numpy
# replay of its index arithmetic + packer passed on the CPU): parity cases, per-layer timing on the 256x512 layers of c2, A/B of the c2 bench,
# and - only if it wins - the c2-shaped model tests with the new table.  Progress is appended to decision.txt step by step (the budget may cut the run).
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r03_s30
mkdir -p $OUT
T=monorec_amd/tuned_winograd.json
cp $T $OUT/table_old.json; cp $T $OUT/table_new.json
timeout 45 python -m pytest tests/test_gpu_model.py -q -x -k "c2_config" > $OUT/t_default.log 2>&1     # the default path on THIS library (ABI 15), old table
echo "c2 fixture test, committed table rc=$?" | tee $OUT/decision.txt; tail -1 $OUT/t_default.log
timeout 45 python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd44 or f44_kernel" > $OUT/t_new.log 2>&1
RC=$?; tail -2 $OUT/t_new.log; echo "parity rc=$RC" | tee -a $OUT/decision.txt
[ $RC -eq 0 ] || exit 0
timeout 45 python tools/bench_wino.py --min-pixels 131072 --emit $OUT/table_new.json > $OUT/wino_c2.log 2>&1
echo "microbench rc=$?" | tee -a $OUT/decision.txt; tail -1 $OUT/wino_c2.log
B="--steps 80 --warmup 10 --spinup-seconds 1 --no-cpu-baseline --no-primer --no-forward-api"
timeout 30 python bench.py $B > $OUT/A_old.json 2> $OUT/A_old.err
cp $OUT/table_new.json $T
timeout 30 python bench.py $B > $OUT/B_new.json 2> $OUT/B_new.err
python - <<PY | tee -a $OUT/decision.txt
import json
def val(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])["value"]
    except Exception:
        return 0.0
a, b = val("$OUT/A_old.json"), val("$OUT/B_new.json")
print(f"c2 80 steps: old table {a:.1f}, new table {b:.1f} keyframes/s")
PY
timeout 70 python -m pytest tests/test_gpu_model.py -q -x -k "c2_config or reference_example" > $OUT/model_tests.log 2>&1
echo "model tests with the new table rc=$?" | tee -a $OUT/decision.txt; tail -2 $OUT/model_tests.log
timeout 45 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_style_new.json 2> $OUT/driver_style_new.err
echo "driver-style rc=$?" | tee -a $OUT/decision.txt
tail -1 $OUT/driver_style_new.json | cut -c1-150
cp $OUT/table_old.json $T
