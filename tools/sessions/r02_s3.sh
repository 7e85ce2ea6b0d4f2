#!/bin/bash
# Round-2 third hardware session: real-sample diagnosis, stream-aware schedule tuning A/B, bf16x3 schedules, c5-shape bench lines.
OUT=gpurun_out/s3
mkdir -p $OUT
timeout 300 python tests/diagnostics/diag_kitti_flips.py > $OUT/diag_flips.txt 2>&1; tail -22 $OUT/diag_flips.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -k "upconv or in_place or bf16 or owned or data_parallel" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python -m pytest tests/test_pointcloud.py tests/test_evaluate_loop.py -m gpu -q > $OUT/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -4 $OUT/pytest2.log
# A: the isolated-launch table (committed).  B: every layer re-timed with the same launch repeated on two streams
cp monorec_amd/tuned_schedules.json $OUT/table_A.json
timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table A', round(d['value'],1), 'kf/s')"
timeout 900 python tools/tune_conv.py --streams 2 --out $OUT/table_B.json > $OUT/tune_B.log 2>&1; tail -2 $OUT/tune_B.log
python - <<'PY'
import json
a = json.load(open("gpurun_out/s3/table_A.json")); b = json.load(open("gpurun_out/s3/table_B.json"))
a.update(b); json.dump(a, open("monorec_amd/tuned_schedules.json", "w"), indent=0, sort_keys=True)
print("entries changed by the two-stream tuning:", sum(1 for k in b if json.load(open("gpurun_out/s3/table_A.json")).get(k) != b[k]), "of", len(b))
PY
timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table B', round(d['value'],1), 'kf/s')"
cp monorec_amd/tuned_schedules.json $OUT/table_AB.json
cp $OUT/table_A.json monorec_amd/tuned_schedules.json
# bf16x3 schedules (merged into table A), then the secondary numbers
timeout 900 python tools/tune_conv.py --bf16x3 --merge > $OUT/tune_bf16x3.log 2>&1; tail -2 $OUT/tune_bf16x3.log
cp monorec_amd/tuned_schedules.json $OUT/table_A_bf16x3.json
timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-primer --bf16x3 > $OUT/c2_bf16x3_bench.json 2>/dev/null; cut -c1-200 $OUT/c2_bf16x3_bench.json
# BASELINE configs[4] shape: fp32, bf16 MFMA mode, bf16x3
for MODE in "" "--bf16" "--bf16x3"; do
  timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-primer --height 512 --width 1024 --frames 4 --depths 48 $MODE > $OUT/c5$MODE.json 2>/dev/null; cut -c1-220 $OUT/c5$MODE.json
done
