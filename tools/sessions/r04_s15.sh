#!/bin/bash
# Round 4, session 15: (a) the c2 schedule table re-tuned with every candidate timed on TWO concurrent streams (the regime of two keyframes
# in flight) - A/B against the table in the tree; (b) tables for coalesced requests (hip_batch_keyframes = 2 / 4: plans of batch 2 / 4 at the c2
# shape had no measured entries - schedules, 3x3 Winograd variants, 1-D forms, transposed) and the secondary dynamic-batching numbers on them.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s15
mkdir -p $OUT
B="--steps 200 --no-primer --no-cpu-baseline --no-forward-api"
# (a)
cp monorec_amd/tuned_schedules.json $OUT/tuned_s2.json
timeout 600 python tools/tune_conv.py --streams 2 --out $OUT/tuned_s2.json --merge --report $OUT/tune_s2_report.json > $OUT/tune_s2.log 2>&1; echo "tune streams=2 rc=$?"; tail -1 $OUT/tune_s2.log
for i in 1 2; do
  timeout 200 python bench.py $B > $OUT/c2_base_$i.json 2> $OUT/c2_base_$i.err
  MR_TUNED_SCHEDULES=$OUT/tuned_s2.json timeout 200 python bench.py $B > $OUT/c2_s2_$i.json 2> $OUT/c2_s2_$i.err
done
# (b)
cp monorec_amd/tuned_schedules.json $OUT/tuned_schedules_b.json
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd_b.json
for k in 2 4; do
  timeout 600 python tools/tune_conv.py --batch $k --out $OUT/tuned_schedules_b.json --merge --missing --report $OUT/tune_b$k.json > $OUT/tune_b$k.log 2>&1; echo "tune batch $k rc=$?"; tail -1 $OUT/tune_b$k.log
  export MR_TUNED_SCHEDULES=$OUT/tuned_schedules_b.json
  timeout 600 python tools/bench_wino.py --batch $k --emit $OUT/tuned_winograd_b.json > $OUT/wino_b$k.log 2>&1; echo "wino batch $k rc=$?"; tail -1 $OUT/wino_b$k.log | cut -c1-200
  timeout 600 python tools/bench_wino_t.py --batch $k --emit $OUT/tuned_winograd_b.json > $OUT/wino_t_b$k.log 2>&1; echo "wino_t batch $k rc=$?"; tail -1 $OUT/wino_t_b$k.log | cut -c1-200
  timeout 600 python tools/bench_wino1d.py --batch $k --emit $OUT/tuned_winograd_b.json > $OUT/wino1d_b$k.log 2>&1; echo "wino1d batch $k rc=$?"; tail -1 $OUT/wino1d_b$k.log | cut -c1-200
  unset MR_TUNED_SCHEDULES
done
python - <<'PY'
import json, sys, time, collections
sys.path.insert(0, ".")
import os
def run(tables):
    import subprocess
    env = dict(os.environ)
    if tables:
        env["MR_TUNED_SCHEDULES"] = "gpurun_out/r04_s15/tuned_schedules_b.json"
        env["MR_TUNED_WINOGRAD"] = "gpurun_out/r04_s15/tuned_winograd_b.json"
    code = r'''
import sys, time, collections, json, torch
sys.path.insert(0, ".")
from monorec_amd import MonoRecModel, synth
dev = torch.device("cuda:0")
m0 = MonoRecModel(cv_depth_steps=32)
sd = synth.seeded_state_dict(m0.state_dict())
batch = synth.make_batch(1, 256, 512, 2, seed=0)
bd = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in batch.items()}
out = {}
for k in (1, 2, 4):
    m = MonoRecModel(cv_depth_steps=32, hip_in_flight=2, hip_batch_keyframes=k)
    m.load_state_dict(sd); m = m.to(dev).eval()
    pending = collections.deque()
    def run(n):
        for _ in range(n):
            pending.append(m.submit(dict(bd)))
            if len(pending) >= 2 * k:
                pending.popleft().synchronize()
        while pending:
            pending.popleft().result()
        torch.cuda.synchronize()
    with torch.no_grad():
        run(60 * k)
        t0 = time.perf_counter(); run(400); dt = time.perf_counter() - t0
    out[k] = round(400 / dt, 1)
    del m
print(json.dumps(out))
'''
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    return (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]
print("requests per launch 1/2/4, tables of the tree:", run(False))
print("requests per launch 1/2/4, tables with batch-2/4 entries:", run(True))
for f in ("c2_base_1", "c2_s2_1", "c2_base_2", "c2_s2_2"):
    try:
        d = json.loads(open(f"gpurun_out/r04_s15/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "sum of kernels", round(d["device_ms_per_step_sum_of_kernels"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
