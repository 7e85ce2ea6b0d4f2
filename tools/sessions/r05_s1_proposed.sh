#!/bin/bash
# Proposed first hardware session of the next round (not run): the state of the tree as the round-4 evidence left it, then the two measurements
# DESIGN section 8 asks for before any new kernel is written.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s1
mkdir -p $OUT
# 1. the tree as committed: whole suite, smoke, the driver command (expected: 347 passed / 21 skipped; c2 ~700 driver-style, ~750 over 200 steps)
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $OUT/suite.log | tail -1
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
# 2. F(4x4,3x3): what a second workgroup per CU could buy - the existing kernel with HALF the chunk (4 channels: 78 KB of LDS) still runs one workgroup per CU
#    (254 registers); an upper bound for the position-split design is the F(2x2,3x3) rb<1> ratio between one and two workgroups per CU (r03: 15-21 %).
#    Measure the ablation instantiations again on the final kernel (MFMA / patch reads / transform / DMA / A reads):
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
for dbg in 0 1 2 4 8 16 32; do
  MR_W44_DBG=$dbg timeout 200 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --only mask.enc0.0 2>/dev/null | grep -o '"wino31_us": [0-9.]*' | sed "s/^/dbg $dbg /"
done
unset MR_HIP_LIBRARY
# 3. the stride-2 layers on the direct kernel (the polyphase candidates of DESIGN 8.4: numerics already cleared on the oracle): their times at c2 / c3
timeout 300 python bench.py --steps 40 --no-primer --no-cpu-baseline --no-forward-api --dump-layers $OUT/layers_c2.json > /dev/null 2>&1
timeout 300 python bench.py --steps 20 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api --dump-layers $OUT/layers_c3.json > /dev/null 2>&1
python - <<'PY'
import json
for tag in ("c2", "c3"):
    rows = json.load(open(f"gpurun_out/r05_s1/layers_{tag}.json"))
    sel = [r for r in rows if any(k in r["name"] for k in ("enc1.0", "enc2.0", "enc3.0", "enc4.0")) and r["name"].startswith("depth.")]
    print(tag, "stride-2 layers:", round(sum(r["seconds"] for r in sel) * 1e6, 1), "us of", round(sum(r["seconds"] for r in rows) * 1e6, 1), [(r["name"], round(r["seconds"] * 1e6, 1), r["tflops"] and round(r["tflops"], 1)) for r in sel])
PY
