#!/bin/bash
# Round 6, session 6: bf16 path (BASELINE configs[4]) - FIRST hardware run of (a) the input ring of conv_b8_kernel (mr_b8_conv_desc.pipeline_stages) and
# (b) B8 copies of the image features (decoders no longer stage fp32 planes through registers).  Parity, per-layer effect, re-tune with ring depths, lines.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_s6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_b8.py -m gpu -q -x -p no:cacheprovider > $OUT/b8_tests.log 2>&1; echo "b8 tests rc=$?"; tail -4 $OUT/b8_tests.log | cut -c1-300
for L in "enc0.1 3,2,4 3,2,4,3 3,4,8 3,4,8,3 3,4,8,4 3,4,8,6 3,2,8 3,2,8,4 3,2,8,6" "dec3.1 3,4,8 3,4,8,3 3,4,8,4 3,2,8,4" "enc1.0 3,2,4 3,4,8,4 3,2,8,4"; do
  set -- $L; layer=$1; shift
  timeout 200 python tools/bench_b8.py --layer $layer --scheds "$@" 2>/dev/null | grep sched
done | tee $OUT/b8_ring_layers.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3), 'depth err', d.get('depth_max_abs_err_vs_cpu'))"; }
C5="--height 512 --width 1024 --frames 4 --depths 48 --bf16"
MR_B8_FEATS=0 timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "c5 bf16, fp32 features (round 5 plan):"
timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_c5_feats.json 2>/dev/null | line "c5 bf16, B8 feature copies, round-5 table:"
cp monorec_amd/tuned_b8.json $OUT/tuned_b8_new.json
timeout 1500 python tools/tune_b8.py --stages 0,3,4,6 --emit $OUT/tuned_b8_new.json > $OUT/tune_b8.log 2>&1; echo "tune_b8 rc=$?"; tail -1 $OUT/tune_b8.log | cut -c1-300
python - <<'PY'
import json
for ln in open("gpurun_out/r06_s6/tune_b8.log"):
    if not ln.startswith("{") or '"name"' not in ln: continue
    r = json.loads(ln)
    print(f"{r['name']:24s} rule {r['rule']} {r.get('rule_us')} best {r.get('best')} {r.get('best_us')}")
PY
for rep in 1 2; do
  timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary 2>/dev/null | line "c5 bf16, round-5 table:"
  MR_TUNED_B8=$OUT/tuned_b8_new.json timeout 400 python bench.py $C5 --steps 100 --no-primer --no-cpu-baseline --no-forward-api --no-secondary --dump-layers $OUT/layers_c5_new.json 2>/dev/null | line "c5 bf16, new table:"
done
