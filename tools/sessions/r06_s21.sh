#!/bin/bash
# Round 6, session 21: evidence on the final kernels / tables (direct kernel on the two-buffer loop, conv_b8 two-stage, B8 table of r06_s19): whole gpu suite,
# smoke, profile sets c2 / c3 / configs[4] bf16 (tools/profile_round.sh), per-workgroup timeline of the direct kernel, the committed lines.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s21
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -6 $OUT/suite.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
bash tools/profile_round.sh r06_c2 > $OUT/profile_c2.log 2>&1; echo "profile c2 rc=$?"; tail -3 $OUT/profile_c2.log | cut -c1-200
bash tools/profile_round.sh r06_c3 "--batch 8 --frames 4 --depths 64" 12 > $OUT/profile_c3.log 2>&1; echo "profile c3 rc=$?"; tail -3 $OUT/profile_c3.log | cut -c1-200
bash tools/profile_round.sh r06_c5bf16 "--height 512 --width 1024 --frames 4 --depths 48 --bf16" 20 > $OUT/profile_c5.log 2>&1; echo "profile c5bf16 rc=$?"; tail -3 $OUT/profile_c5.log | cut -c1-200
L=resnet.l1b0.conv1,resnet.l2b1.conv1,resnet.l3b1.conv1,resnet.l4b1.conv1,mask.enc2.1,mask.enc3.1,mask.dec1.1,depth.enc3.1.conv_x,depth.enc4.1.conv_x,depth.dec0,depth.dec1.0
MR_TL_DBG=16 timeout 300 python tools/wg_timeline.py $L > $OUT/wg_timeline.log 2>&1; cp gpurun_out/wg_timeline.json $OUT/wg_timeline.json; grep -c sched $OUT/wg_timeline.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style_c2.json 2> $OUT/driver_style_c2.err; echo "driver-style rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_s21/driver_style_c2.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("driver-style", round(d["value"], 1), "kf/s; 200:", round(d.get("value_200_steps", 0), 1), "primed", round(d.get("value_host_primed", 0), 1), "forward_api", round(d["forward_api"]["value"], 1),
      "frac", round(r["frac"], 3), r["frac_source"], "kernel_only", r.get("kernel_only", {}).get("frac"), "dominant", r["dominant"]["name"], round(r["dominant"]["avg_us"], 2), round(r["dominant"]["frac"], 3),
      "cpu", round(d["cpu_baseline"]["value"], 3), "depth vs cpu", d.get("depth_max_abs_err_vs_cpu"), "skip_l4", round(d["secondary_skip_dead_layer4"]["value"], 1))
PY
timeout 400 python bench.py --batch 8 --frames 4 --depths 64 --steps 30 --no-primer --no-forward-api > $OUT/c3_line.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_s21/c3_line.json').read().strip().splitlines()[-1]); print('c3', round(d['value'],1), d['roofline']['frac'], d.get('depth_max_abs_err_vs_cpu'))"
timeout 600 python bench.py --height 512 --width 1024 --frames 4 --depths 48 --bf16 --steps 60 --no-primer --no-forward-api > $OUT/c5bf16_line.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_s21/c5bf16_line.json').read().strip().splitlines()[-1]); print('c5 bf16', round(d['value'],1), d['roofline']['frac'], d.get('depth_max_abs_err_vs_cpu'))"
