#!/bin/bash
# Round 5, session 12: (1) per-workgroup phase breakdown (start after the first workgroup / set-up / first chunk in LDS / K loop / stores) of the direct kernel on one
# layer per ResNet stage and on the 3-tap pairs of depth.enc3 / enc4 (VERDICT r4 #1a) from the -DMR_CONV_TIMELINE library; (2) the whole gpu suite on the new defaults.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s12
mkdir -p $OUT
timeout 300 python tools/wg_timeline.py resnet.l1b0.conv1,resnet.l2b1.conv1,resnet.l3b1.conv1,resnet.l4b1.conv1,depth.enc3.1.conv_y,depth.enc3.1.conv_x,depth.enc4.1.conv_y,depth.enc4.1.conv_x > $OUT/wg_timeline.log 2>&1; echo "timeline rc=$?"
cp gpurun_out/wg_timeline.json $OUT/wg_timeline.json 2>/dev/null
cut -c1-600 $OUT/wg_timeline.log | tail -12
timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -25 $OUT/suite.log | cut -c1-300
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log | cut -c1-300
