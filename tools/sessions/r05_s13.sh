#!/bin/bash
# Round 5, session 13: (1) per-workgroup phase breakdown of the short direct launches with the product's split-K (stamps behind the slabs; r05_s12 stopped at the
# first split-K layer); (2) r05_s11: four keyframes submitted into an empty pipeline within 2 ms run in lock-step and complete in bursts - does spacing the
# submits of a FILLING pipeline (MR_DIAG_FILL_PACE_US) change a 20-step line?
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s13
mkdir -p $OUT
timeout 300 python tools/wg_timeline.py resnet.l1b0.conv1,resnet.l2b1.conv1,resnet.l3b1.conv1,resnet.l4b1.conv1,depth.enc3.1.conv_y,depth.enc3.1.conv_x,depth.enc4.1.conv_y,depth.enc4.1.conv_x > $OUT/wg_timeline.log 2>&1; echo "timeline rc=$?"
cp gpurun_out/wg_timeline.json $OUT/wg_timeline.json 2>/dev/null
grep -c '"sched"' $OUT/wg_timeline.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('step_marks_ms') or []
print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'marks', [round(x,1) for x in m])"; }
for rep in 1 2; do
  for pace in 0 600 900 1200 1500; do
    MR_DIAG_FILL_PACE_US=$pace timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-primer --no-forward-api --step-times 2>/dev/null | line "pace $pace us, 20 steps"
  done
done
for pace in 0 1200; do
  MR_DIAG_FILL_PACE_US=$pace timeout 200 python bench.py --steps 200 --no-cpu-baseline --no-primer --no-forward-api 2>/dev/null | line "pace $pace us, 200 steps"
done
