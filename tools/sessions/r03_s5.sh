#!/bin/bash
# Round 3, session 5: gpu suite with the reworked cost-volume arithmetic (exact constant division, border-mask test as logic, reciprocal
# in the SSIM ratio, branch-free marching step) and the reordered submit; bench lines for device / host 4x4s; cost-volume timings.
OUT=gpurun_out/r03_s5
mkdir -p $OUT
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "fixtures_own_matrices" 2>&1 | grep -E "kitti example|passed|failed" | cut -c1-300
B="python bench.py --steps 300 --no-cpu-baseline"
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    fa=d.get("forward_api",{})
    print(sys.argv[1].split('/')[-1], round(d["value"],1), "keyframes/s", round(d["ms_per_step"],3), "ms  host_enq", round(d.get("host_enqueue_ms",0),3), " forward_api", round(fa.get("value",0),1), " cv us", round(d["cost_volume_kernel"]["us"],1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 240 env "$@" > $OUT/$tag.json 2> $OUT/$tag.err; val $OUT/$tag.json; }
run dev $B
run host $B --no-primer --host-mats
run dev_b $B --no-primer
run host_b $B --no-primer --host-mats
run host_if3 $B --no-primer --host-mats --in-flight 3
run host_if1 $B --no-primer --host-mats --in-flight 1
run dev_if1 $B --no-primer --in-flight 1
run host_hq32 $B --no-primer --host-mats --hw-queues 32
run driver_style python bench.py --gpus 1 --steps 20 --warmup 5
run c3 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer
run c3_host python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer --host-mats
timeout 200 python tools/bench_cv.py 2>&1 | tail -8
