#!/bin/bash
# Round 4, session 10: branch-free activations in every convolution epilogue (direct, Winograd, Cook-Toom, transposed, Upconv, finisher);
# kernel + model suites, c2 lines, one-keyframe-at-a-time trace; c5 bf16 with 3 keyframes in flight.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s10
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q > $OUT/kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $OUT/kernels.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "not c3_full and not c5_shape and not harsh_weights_c3" > $OUT/model.log 2>&1; echo "model tests rc=$?"; tail -3 $OUT/model.log | cut -c1-300
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c2_200.json 2> $OUT/c2_200.err; echo "c2 rc=$?"
timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --in-flight 1 > $OUT/c2_if1.json 2> $OUT/c2_if1.err
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$?"
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api --in-flight 3 > $OUT/c5_bf16_if3.json 2> $OUT/c5_bf16_if3.err
python - <<'PY'
import json
for f in ("c2_200", "c2_if1", "c3", "c5_bf16_if3"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s10/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 1), "kf/s, ms/step", round(d["ms_per_step"], 3), "conv ms (events)", round(r["conv_ms_per_step"], 3), "sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_seq -o t -- python $REPO/bench.py --steps 40 --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace_seq.log 2>&1
cd $REPO
python tools/summarize_prof.py --tag r04b_c2 --stats-seq $(find $OUT/trace_seq -name "*_results.db" | head -1) > /dev/null 2>&1
mkdir -p $OUT/profiles && cp profiles/r04b_c2_* $OUT/profiles/ 2>/dev/null
find $OUT -name "*.db" -delete
