#!/bin/bash
# Round 3, session 2: gpu suite with the round's host-side changes (owned outputs by one copy launch, submit / result streams, run-ahead
# bound, 1-rank RCCL group, fresh-model DataParallel, composed real-sample leg, Winograd variants per descriptor), then the pipelining
# matrix (in-flight x queue depth x HW queues), forward_api, and the Winograd tables of c3 / configs[4] with both variants.
OUT=gpurun_out/r03_s2
mkdir -p $OUT
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "kitti example|undetermined" $OUT/pytest.log | head
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "fixtures_own_matrices" 2>&1 | grep -E "kitti example|passed|failed" | cut -c1-300
B="python bench.py --steps 200 --no-cpu-baseline"
val() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    fa=d.get("forward_api",{})
    print(sys.argv[1].split('/')[-1], round(d["value"],1), "keyframes/s", round(d["ms_per_step"],3), "ms  host_enq", round(d.get("host_enqueue_ms",0),3), " forward_api", round(fa.get("value",0),1), fa.get("outputs_owned_by_caller"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run() { tag=$1; shift; timeout 240 env "$@" > $OUT/$tag.json 2> $OUT/$tag.err; val $OUT/$tag.json; }
run default $B
run default_b $B --no-primer
run caller_stream $B --no-primer --caller-stream
run qd1 $B --no-primer --queue-depth 1
run qd3 $B --no-primer --queue-depth 3
run if3 $B --no-primer --in-flight 3
run if4 $B --no-primer --in-flight 4
run if3_qd1 $B --no-primer --in-flight 3 --queue-depth 1
run hwq4 $B --no-primer --hw-queues 4
run hwq8 $B --no-primer --hw-queues 8
run hwq8_if3 $B --no-primer --hw-queues 8 --in-flight 3
run hostmats $B --no-primer --host-mats
run hostmats_if3 $B --no-primer --host-mats --in-flight 3
run if1 $B --no-primer --in-flight 1
run graph_if1 $B --no-primer --in-flight 1 --graph
run graph $B --no-primer --graph
run wino_off MR_WINOGRAD=0 $B --no-primer
run driver_style python bench.py --gpus 1 --steps 20 --warmup 5
echo "--- c3 / configs[4] Winograd tables, both variants"
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd.json
timeout 400 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --emit $OUT/tuned_winograd.json > $OUT/wino_c3.jsonl 2>$OUT/wino_c3.err; tail -1 $OUT/wino_c3.jsonl
timeout 400 python tools/bench_wino.py --height 512 --width 1024 --frames 4 --depths 48 --emit $OUT/tuned_winograd.json > $OUT/wino_c5.jsonl 2>$OUT/wino_c5.err; tail -1 $OUT/wino_c5.jsonl
python - <<'PY'
import json
for f in ("wino_c3","wino_c5"):
    for l in open(f"gpurun_out/r03_s2/{f}.jsonl"):
        if '"name"' in l:
            r=json.loads(l); print(f, r["name"], "direct", r["direct_us"], {c:r.get(f"wino{c}_us") for c in (1,2,11,12)}, "best", r["best"])
PY
run c3 python bench.py --steps 60 --batch 8 --frames 4 --depths 64 --no-cpu-baseline --no-primer
