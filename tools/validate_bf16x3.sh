#!/bin/bash
# First hardware session of the bf16x3 split mode - run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/validate_bf16x3.sh'
# 1. kernel-level and end-to-end parity (the tests that MR_TEST_EXPERIMENTAL gates), each in its own process so that a faulting
#    kernel cannot hide the other results;  2. a first throughput number next to fp32 and plain bf16 on the same box;
# 3. measured schedules for the c2 shape (merged into the committed table as *_bf16x3 entries), then the number again.
OUT=gpurun_out/bf16x3
mkdir -p $OUT
export MR_TEST_EXPERIMENTAL=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16x3" -s > $OUT/kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $OUT/kernels.log
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "bf16x3" -s > $OUT/model.log 2>&1; echo "model test rc=$?"; tail -3 $OUT/model.log
for MODE in "" "--bf16" "--bf16x3"; do
  timeout 200 python bench.py --steps 200 --no-cpu-baseline $MODE > $OUT/bench$MODE.json 2> $OUT/bench$MODE.err
  python - "$OUT/bench$MODE.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["dtype"], round(d["value"], 1), "keyframes/s", "conv TF/s", round(d["roofline"]["achieved"], 1))
PY
done
timeout 600 python tools/tune_conv.py --bf16x3 --merge > $OUT/tune.log 2>&1; echo "tuner rc=$?"; tail -2 $OUT/tune.log
cp monorec_amd/tuned_schedules.json $OUT/tuned_schedules.json
timeout 200 python bench.py --steps 200 --no-cpu-baseline --bf16x3 > $OUT/bench_tuned.json 2> $OUT/bench_tuned.err
tail -c 400 $OUT/bench_tuned.json
