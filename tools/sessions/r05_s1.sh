#!/bin/bash
# Round 5, session 1: first hardware run of the round's first batch of changes (ABI 18):
#   branch-free activation as max(x, lo), forward() with its arenas bound ahead of the input wait + gather launch ahead of the encoder,
#   prepare() that leaves the pose algebra to submit() on an idle device, mr_cost_volume_relaxed_f32 (fp32 opt-in) + the oracle legs of both relaxed
#   entry points, F(4,4) + the stride-2 ConvReLU2 pairs on the Cook-Toom kernel over [even | odd] views, bench line hygiene, 2-rank bench on one device.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s1
mkdir -p $OUT
# 1. whole gpu suite (r04: 347 passed / 21 skipped in 5.8 min)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -x -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -15 $OUT/suite.log | cut -c1-400
# the measured values behind the tightened bf16 bars and the relaxed cost-volume bars (prints are swallowed by -q)
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -s -k "bf16_mode or c5_shape or relaxed_cost_volume or separable" -p no:cacheprovider 2>&1 | grep -E "bf16|relaxed|separable|passed|failed" | cut -c1-300 > $OUT/bars.log; cat $OUT/bars.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
# 2. the driver command
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; echo "driver-style rc=$?"
# 3. stride-2 pairs: direct vs the Cook-Toom forms over [even | odd] (c2, c3)
cp monorec_amd/tuned_winograd.json $OUT/tuned_winograd_s2.json
timeout 300 python tools/bench_stride2.py --emit $OUT/tuned_winograd_s2.json > $OUT/stride2_c2.log 2>&1; echo "stride2 c2 rc=$?"; cat $OUT/stride2_c2.log | cut -c1-600
timeout 300 python tools/bench_stride2.py --batch 8 --frames 4 --depths 64 --emit $OUT/tuned_winograd_s2.json > $OUT/stride2_c3.log 2>&1; echo "stride2 c3 rc=$?"; cat $OUT/stride2_c3.log | cut -c1-600
# A/B of the emitted table, interleaved, 200 steps each
for rep in 1 2; do
  for tab in monorec_amd/tuned_winograd.json $OUT/tuned_winograd_s2.json; do
    MR_TUNED_WINOGRAD=$tab timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', '$tab'.split('/')[-1], round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'ms')"
  done
done
for tab in monorec_amd/tuned_winograd.json $OUT/tuned_winograd_s2.json; do
  MR_TUNED_WINOGRAD=$tab timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', '$tab'.split('/')[-1], round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'ms')"
done
# 4. where the wall time of one keyframe goes (kernel trace, one keyframe at a time on one stream)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_seq -o t -- python $REPO/bench.py --steps 60 --in-flight 1 --single-stream --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace_seq.json 2> $OUT/trace_seq.err
cd $REPO
f=$(find $OUT/trace_seq -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py "$f" --out $OUT/c2_launch_gaps.json > $OUT/gaps.log 2>&1; head -60 $OUT/gaps.log
rm -rf $OUT/trace_seq
# 5. the fp32 opt-in with separable cost-volume sums on the c3 line; lean outputs on the configs[4] bf16 line
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline > $OUT/c3.json 2> /dev/null
timeout 200 python bench.py --steps 40 --batch 8 --frames 4 --depths 64 --no-primer --no-cpu-baseline --no-forward-api --cv-separable > $OUT/c3_sep.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --no-cpu-baseline --no-primer > $OUT/c5_bf16.json 2> /dev/null
timeout 200 python bench.py --steps 60 --height 512 --width 1024 --frames 4 --depths 48 --bf16 --lean-outputs --no-cpu-baseline --no-primer --no-forward-api > $OUT/c5_bf16_lean.json 2> /dev/null
python - <<'PY'
import json
for f in ("driver_style", "c3", "c3_sep", "c5_bf16", "c5_bf16_lean"):
    try:
        d = json.loads(open(f"gpurun_out/r05_s1/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        fa = d.get("forward_api", {}).get("value")
        print(f, round(d["value"], 1), "kf/s; 200:", d.get("value_200_steps") and round(d["value_200_steps"], 1), "primed:", d.get("value_host_primed") and round(d["value_host_primed"], 1),
              "forward_api", fa and round(fa, 1), "bound", r["bound"], "frac", round(r["frac"], 3), r["frac_source"], "ceiling", round(r["vs_direct_conv_ceiling"], 3),
              "hip_events frac", round(r["hip_events"]["frac"], 3), "pipelined", round(r["frac_pipelined"], 3), "launches", r["all_kernel_launches_per_step"],
              "cv us", round(d["cost_volume_kernel"]["us"], 1), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("port_vs_reference"),
              "depth err", d.get("depth_max_abs_err_vs_cpu"))
    except Exception as e:
        print(f, "failed", repr(e))
PY
