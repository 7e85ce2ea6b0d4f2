#!/bin/bash
# Round 3, first hardware session: what limits the overlap of the two in-flight keyframes?  HW-queue count, kernarg placement, host-side
# 4x4s (no D2H wait in submit), more keyframes in flight; then the Winograd kernel with the input transform in registers per layer.
OUT=gpurun_out/r03_s1
mkdir -p $OUT
B="python bench.py --steps 200 --no-cpu-baseline"
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],1), "keyframes/s", round(d["ms_per_step"],3), "ms  sumk", round(d["device_ms_per_step_sum_of_kernels"],3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 240 env "$@" > $OUT/$tag.json 2> $OUT/$tag.err; val $OUT/$tag.json; }
run base0 $B
run base1 $B --no-primer
run hwq8 GPU_MAX_HW_QUEUES=8 $B --no-primer
run hwq8_if3 GPU_MAX_HW_QUEUES=8 $B --no-primer --in-flight 3
run hwq16_if4 GPU_MAX_HW_QUEUES=16 $B --no-primer --in-flight 4
run hwq2 GPU_MAX_HW_QUEUES=2 $B --no-primer
run kernarg1 HIP_FORCE_DEV_KERNARG=1 $B --no-primer
run kernarg0 HIP_FORCE_DEV_KERNARG=0 $B --no-primer
run hostmats $B --no-primer --host-mats
run hostmats_hwq8_if3 GPU_MAX_HW_QUEUES=8 $B --no-primer --host-mats --in-flight 3
run hostmats_if1 $B --no-primer --host-mats --in-flight 1
run if1 $B --no-primer --in-flight 1
run regb MR_WINO_REGB=1 $B --no-primer
run base2 $B --no-primer
echo "--- winograd per layer, V buffer"
timeout 200 python tools/bench_wino.py > $OUT/wino_v.jsonl 2>/dev/null; tail -1 $OUT/wino_v.jsonl
echo "--- winograd per layer, transform in registers"
MR_WINO_REGB=1 timeout 200 python tools/bench_wino.py > $OUT/wino_rb.jsonl 2>/dev/null; tail -1 $OUT/wino_rb.jsonl
python - <<'PY'
import json
a=[json.loads(l) for l in open("gpurun_out/r03_s1/wino_v.jsonl") if '"name"' in l]
b={r["name"]:r for r in (json.loads(l) for l in open("gpurun_out/r03_s1/wino_rb.jsonl") if '"name"' in l)}
for r in a:
    q=b.get(r["name"],{})
    print(r["name"], "direct", r["direct_us"], "V", r.get("wino1_us"), r.get("wino2_us"), "RB", q.get("wino1_us"), q.get("wino2_us"))
PY
