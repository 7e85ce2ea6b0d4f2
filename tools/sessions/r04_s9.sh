#!/bin/bash
# Round 4, session 9: tile bookkeeping hoisted out of the tile loop; workgroups per launch of the resident-weight mode.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s9
mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_b8.py -q > $OUT/b8_kernels.log 2>&1; echo "b8 kernel tests rc=$?"; tail -3 $OUT/b8_kernels.log | cut -c1-300
export MR_HIP_LIBRARY=$REPO/monorec_amd/libmonorec_hip_timeline.so
{
for WGS in 256 512 1024 2048; do
  MR_B8_WGS=$WGS timeout 100 python tools/bench_b8.py --layer enc0.1 --scheds 3,4,8 3,2,4 2>&1 | grep sched | sed "s/^/wgs_target $WGS: /"
  MR_B8_WGS=$WGS timeout 100 python tools/bench_b8.py --layer dec3.1 --scheds 3,4,8 2>&1 | grep sched | sed "s/^/wgs_target $WGS: /"
  MR_B8_WGS=$WGS timeout 100 python tools/bench_b8.py --layer enc0.1x --scheds 3,2,8 2>&1 | grep sched | sed "s/^/wgs_target $WGS: /"
done
MR_B8_DBG=15 timeout 100 python tools/bench_b8.py --layer enc0.1 --scheds 3,4,8 2>&1 | grep sched
} | tee $OUT/wgs.log
unset MR_HIP_LIBRARY
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer --no-forward-api 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 bf16', round(d['value'],1), 'kf/s, conv ms', round(d['roofline']['conv_ms_per_step'],3))"
