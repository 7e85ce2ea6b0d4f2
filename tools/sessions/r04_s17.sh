#!/bin/bash
# Round 4, session 17: do two streams overlap at all when the host is out of the way?  (tools/probes/stream_overlap.py)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r04_s17
timeout 300 python tools/probes/stream_overlap.py > gpurun_out/r04_s17/overlap.json 2> gpurun_out/r04_s17/overlap.err; echo "rc=$?"; cat gpurun_out/r04_s17/overlap.json; tail -3 gpurun_out/r04_s17/overlap.err
GPU_MAX_HW_QUEUES=4 timeout 300 python tools/probes/stream_overlap.py > gpurun_out/r04_s17/overlap_q4.json 2>/dev/null; echo "4 HW queues:"; cat gpurun_out/r04_s17/overlap_q4.json
