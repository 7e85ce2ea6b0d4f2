#!/bin/bash
# forward() after the 4x4s come back through one gather launch and the host waits for the caller-stream marker first
OUT=gpurun_out/r03_s8; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_pointcloud.py -m gpu -q -x -k "gather or forward or pointcloud or reference_loop or data_parallel or separate_streams or oracle_and_fixture" 2>&1 | tail -3
for cfg in "--in-flight 1" "--in-flight 2" "--in-flight 2 --host-mats" "--in-flight 1 --host-mats"; do
  echo "cfg: $cfg"; python tools/trace_forward.py $cfg 2>&1 | tail -6 | cut -c1-150
done
B="python bench.py --steps 300 --no-cpu-baseline --no-primer"
for cfg in "" "--host-mats" "--in-flight 1" ""; do
  timeout 200 $B $cfg > $OUT/b.json 2>/dev/null; python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_s8/b.json").read().strip().splitlines()[-1]); fa=d.get("forward_api",{})
print("bench", sys.argv[1], round(d["value"],1), "host_enq", round(d["host_enqueue_ms"],3), "forward_api", round(fa.get("value",0),1))
PY
done
