#!/bin/bash
# Round 6, session 2: where does the sweep of the direct kernel spend its time?  r06_s1 showed that the register double buffer of the LDS operand
# reads (software-pipelined sweep) and the LDS ring change NOTHING on the c2 layers (l1: 14.6 -> 14.6 us) - the "load-all / wait / MFMA-all" loop
# body was not the limiter with two waves per SIMD.  Ablation on the diagnostic library (per-workgroup stamps, tools/wg_timeline.py):
# MR_TL_DBG bits: 16 stamps, 2 no input DMA, 4 no weight DMA, 32 no A (weight) LDS reads, 64 no B (input) LDS reads.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s2
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv and not cost_volume" -p no:cacheprovider > $OUT/conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -3 $OUT/conv_tests.log | cut -c1-300
L=resnet.l1b0.conv1,mask.enc2.1,depth.dec1.0,resnet.l3b1.conv1
for DBG in 16 22 54 86 118; do
  echo "== MR_TL_DBG=$DBG"
  MR_TL_DBG=$DBG timeout 300 python tools/wg_timeline.py $L 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ' {' not in ln: continue
    name, js = ln.split(' ', 1); d = json.loads(js)
    print(f\"{name:22s} sched {d['sched']} span {d['span_us']} setup {d['setup'][1]} first {d['first_chunk'][1]} k_loop {d['k_loop'][1]} sweep {d['chunk_sweep'][1]} issue {d['chunk_issue'][1]} stores {d['stores'][1] if d['stores'] else None} clk {d['clock64_ticks_per_us']}\")
"
  cp gpurun_out/wg_timeline.json $OUT/wg_timeline_dbg$DBG.json
done
