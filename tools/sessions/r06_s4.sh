#!/bin/bash
# Round 6, session 4: the sweep of the direct kernel specialised on the LDS plane pitch (operand reads with immediate offsets: 3 VALU per 8 MFMAs
# instead of 10) - parity, per-workgroup timeline, c2 line on the round-5 table, re-tune of the direct-kernel layers (ring candidates included), A/B.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv and not cost_volume" -p no:cacheprovider > $OUT/conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -3 $OUT/conv_tests.log | cut -c1-300
L=resnet.l1b0.conv1,mask.enc2.1,depth.dec1.0,resnet.l3b1.conv1,mask.dec1.1,depth.enc3.1.conv_x
MR_TL_DBG=16 timeout 300 python tools/wg_timeline.py $L 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ' {' not in ln: continue
    name, js = ln.split(' ', 1); d = json.loads(js)
    print(f\"{name:22s} sched {d['sched']} span {d['span_us']} setup {d['setup'][1]} first {d['first_chunk'][1]} k_loop {d['k_loop'][1]} sweep {d['chunk_sweep'][1]} issue {d['chunk_issue'][1]} stores {d['stores'][1] if d['stores'] else None} clk {d['clock64_ticks_per_us']}\")
"
cp gpurun_out/wg_timeline.json $OUT/wg_timeline.json
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_old_table.json 2>$OUT/bench1.err | tee $OUT/bench_old_table.json | line "r5 table, 200 steps:"
DIRECT=resnet.,mask.enc2,mask.enc3,mask.enc4,mask.dec0,mask.dec1,depth.enc2.0.conv_x,depth.enc3,depth.enc4,depth.dec0,depth.dec1.0
cp monorec_amd/tuned_schedules.json $OUT/tuned_new.json
timeout 1500 python tools/tune_conv.py --merge --only $DIRECT --ring 3,4 --out $OUT/tuned_new.json --report $OUT/tune_report.json > $OUT/tune.log 2>&1; echo "tune rc=$?"; grep "best=" $OUT/tune.log | cut -c1-120; tail -1 $OUT/tune.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "r5 table:"
  MR_TUNED_SCHEDULES=$OUT/tuned_new.json timeout 300 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_new_table.json 2>/dev/null | line "new table:"
done
MR_TUNED_SCHEDULES=$OUT/tuned_new.json timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "new table, driver-style 20 steps:"
