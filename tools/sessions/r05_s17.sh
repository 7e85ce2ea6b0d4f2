#!/bin/bash
# Round 5, session 17: more in-flight slots than streams.  r05_s11's step marks say results come back in bursts and the host then needs 0.6 ms per keyframe
# (prepare + enqueue) to refill each idle stream.  Slots share the four streams now (slot s on stream s % 4): a stream's next keyframe is already enqueued
# when the previous one finishes.  c2 at 4 / 6 / 8 / 12 slots, c3 and configs[4] bf16 at 8.
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s17
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'host cpu', round(d['host_cpu_ms_per_keyframe'],2), 'slots', d['config']['keyframes_in_flight'], 'streams', d['config'].get('streams'))"; }
B="--no-cpu-baseline --no-primer --no-forward-api"
for rep in 1 2; do
  for n in 4 6 8 12; do
    timeout 200 python bench.py --steps 200 $B --in-flight $n 2>/dev/null | line "c2 200 steps, $n slots"
  done
done
for n in 4 8; do
  timeout 200 python bench.py --steps 20 --warmup 5 $B --in-flight $n 2>/dev/null | line "c2 20 steps, $n slots"
  timeout 200 python bench.py --steps 20 --warmup 5 $B --in-flight $n 2>/dev/null | line "c2 20 steps, $n slots"
done
timeout 200 python bench.py --steps 200 $B --in-flight 8 --streams 3 2>/dev/null | line "c2 200 steps, 8 slots on 3 streams"
timeout 200 python bench.py --steps 200 $B --in-flight 8 --streams 8 2>/dev/null | line "c2 200 steps, 8 slots on 8 streams"
for n in 4 8; do
  timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 $B --in-flight $n 2>/dev/null | line "c3, $n slots"
  timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 $B --in-flight $n 2>/dev/null | line "configs[4] bf16, $n slots"
done
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_evaluate_loop.py tests/test_pointcloud.py -m gpu -q -p no:cacheprovider -k "fixed_order or flight or batching or pipeline or separate_streams or data_parallel or arenas or owned or evaluat or pointcloud or loop" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log | cut -c1-200
