#!/bin/bash
# Round 3, session 3: a clean grid over what the host loop controls - HW queues x submit/result streams x run-ahead depth x where the
# 4x4s live - twice each, interleaved, to separate the effects session 2 mixed up.
OUT=gpurun_out/r03_s3
mkdir -p $OUT
B="python bench.py --steps 300 --no-cpu-baseline --no-primer --no-forward-api"
python bench.py --steps 50 --no-cpu-baseline > /dev/null 2>&1     # primer + page-in
for rep in 1 2; do
for hq in 4 16; do
for st in "" "--caller-stream"; do
for qd in 1 2 100; do
for hm in "" "--host-mats"; do
  tag="hq${hq}_s${st:+C}_qd${qd}_m${hm:+H}_r$rep"
  timeout 120 $B --hw-queues $hq $st --queue-depth $qd $hm > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1][:-5], round(d["value"],1), "host_enq", round(d.get("host_enqueue_ms",0),3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done; done; done; done; done
