#!/bin/bash
# the 19 split-K layers of c2: fastest schedule without split_k (K split across waves included) next to the current one; adopt it where it is
# within 4 % (one launch less per layer); then the end-to-end A/B
OUT=gpurun_out/r03_s24; mkdir -p $OUT
cp monorec_amd/tuned_schedules.json $OUT/tuned_before.json
B="python bench.py --steps 300 --no-cpu-baseline --no-primer --no-forward-api"
python bench.py --steps 50 --no-cpu-baseline --no-forward-api > /dev/null 2>&1
timeout 200 $B > $OUT/b0.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/b0.json').read().strip().splitlines()[-1]);print('before', round(d['value'],1), d['roofline']['all_kernel_launches_per_step'])"
timeout 900 python tools/tune_conv.py --merge --prefer-nosplit 4 --only resnet.l2b0.conv2,resnet.l2b1,resnet.l3b0.conv,resnet.l3b1,resnet.l4b,mask.dec0.0,mask.dec0.1,mask.dec1.1,depth.enc4.0.conv_x,depth.enc4.1,depth.dec1.0 --out $OUT/tuned_schedules.json 2>$OUT/tune.err | tee $OUT/tune.log | cut -c1-230
cp monorec_amd/tuned_schedules.json $OUT/tuned_merged.json
python - <<'PY'
import json
new=json.load(open("gpurun_out/r03_s24/tuned_schedules.json")); cur=json.load(open("monorec_amd/tuned_schedules.json"))
cur.update(new); json.dump(cur, open("monorec_amd/tuned_schedules.json","w"), indent=0, sort_keys=True)
print("merged", len(new), "entries")
PY
for i in 1 2; do
timeout 200 $B > $OUT/b1.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/b1.json').read().strip().splitlines()[-1]);print('after', round(d['value'],1), d['roofline']['all_kernel_launches_per_step'], round(d['device_ms_per_step_sum_of_kernels'],3))"
done
cp monorec_amd/tuned_schedules.json $OUT/tuned_after.json
