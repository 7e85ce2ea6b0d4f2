#!/bin/bash
# Round 3, session 31 (the last GPU-minute): the final table (F(4x4,3x3) for mask.enc0.* at c2, everything else as in session 26) - the
# c2-shaped model tests, smoke and the driver-style line.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r03_s31
mkdir -p $OUT
timeout 40 python -m pytest tests/test_gpu_model.py -q -x -k "c2_config or reference_example" > $OUT/model_tests.log 2>&1
echo "model tests rc=$?"; tail -1 $OUT/model_tests.log
timeout 20 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-120
timeout 45 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err
tail -1 $OUT/driver_style.json | cut -c1-150
