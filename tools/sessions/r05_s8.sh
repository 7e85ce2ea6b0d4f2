#!/bin/bash
# Round 5, session 8: three quick questions before the final session: stream priorities under the fixed first-use order, one stream per slot, and the stride-2
# tables of the other measured shapes (batch 2 / 4 at the KITTI shape, 512x1024).
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s8
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3))"; }
run() { env $1 timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api $2 2>/dev/null | line "[$1] $2"; }
run "MR_X=0" ""
run "MR_DIAG_STREAM_PRIO=m" ""
run "MR_DIAG_STREAM_PRIO=e" ""
run "MR_DIAG_STREAM_PRIO=me" ""
run "MR_X=0" "--single-stream"
run "MR_X=0" "--single-stream --in-flight 3"
run "MR_X=0" ""
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd_s2.json
for shp in "--batch 2" "--batch 4" "--height 512 --width 1024 --frames 4 --depths 48"; do
  timeout 300 python tools/bench_stride2.py $shp --emit $OUT/tuned_winograd_s2.json > $OUT/stride2.log 2>&1
  grep -o '"name": "[a-z0-9.]*"\|"sig": "[a-z0-9_x]*"\|"direct_us": \[[0-9., ]*\]\|"best": [0-9]*\|"algorithmic_tflops_[a-z]*": [0-9.]*' $OUT/stride2.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
done
python - <<'PY'
import json
a = json.load(open("monorec_amd/tuned_winograd.json")); b = json.load(open("gpurun_out/r05_s8/tuned_winograd_s2.json"))
print({k: v for k, v in b.items() if a.get(k) != v})
PY
