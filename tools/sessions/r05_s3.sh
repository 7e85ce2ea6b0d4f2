#!/bin/bash
# Round 5, session 3: (1) the A/B of r05_s2 again with prepare()'s idle test done on the host (r05_s2: this tree 701-710 against 742-756 keyframes/s of the
# round-4 tree, two keyframes in flight only; suspect: hipEventQuery on the other slot's running keyframe from prepare()); (2) driver-style line;
# (3) every 3x3 form at c2 (the table candidates for the split F(4x4,3x3) kernel); (4) the test that failed in r05_s2 + the host-path tests.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s3
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'host cpu ms', round(d['host_cpu_ms_per_keyframe'],2), 'sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
for rep in 1 2; do
  (cd ab_r4 && timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r4-tree 200")
  timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r5-tree 200"
done
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_s3/driver_style.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver_style", round(d["value"], 1), "200:", round(d["value_200_steps"], 1), "primed:", round(d["value_host_primed"], 1), "forward_api", round(d["forward_api"]["value"], 1),
      "frac", round(r["frac"], 3), r["frac_source"], "pipelined", round(r["frac_pipelined"], 3), "launches", r["all_kernel_launches_per_step"], "host inputs", round(d["with_host_inputs"]["value"], 1),
      "data loading", round(d["with_data_loading"]["value"], 1), "batching", d["secondary_dynamic_batching"]["requests_per_launch_2"]["value"], d["secondary_dynamic_batching"]["requests_per_launch_4"]["value"])
PY
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd_w.json
timeout 400 python tools/bench_wino.py --emit $OUT/tuned_winograd_w.json --min-pixels 8192 > $OUT/wino_c2.log 2>&1; echo "wino c2 rc=$?"
grep -o '"name": "[a-z0-9.]*"\|"direct_us": [0-9.]*\|"wino[0-9]*_us": [0-9.]*\|"best": [0-9]*' $OUT/wino_c2.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
for rep in 1 2; do
  for tab in monorec_amd/tuned_winograd.json $OUT/tuned_winograd_w.json; do
    MR_TUNED_WINOGRAD=$tab timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "c2 $(basename $tab)"
  done
done
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -k "arenas or prepare or flight or owned or submit or forward_between or data_parallel or batching" > $OUT/host_tests.log 2>&1; echo "host tests rc=$?"; tail -3 $OUT/host_tests.log
