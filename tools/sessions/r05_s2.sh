#!/bin/bash
# Round 5, session 2:
#   (1) A/B of the round-4 tree (ab_r4/: `git archive 8feae71`, its own library) against this tree on ONE box - r05_s1 measured the c2 line at 650 / 685
#       (20 / 200 steps) against round 4's 707 / 754 while c3, the batched and the direct-kernel configurations were level: is it the box or the tree?
#   (2) whole gpu suite (r05_s1 stopped at its first failure: a too strict zero-for-zero check in a new test);
#   (3) first hardware run of F(4x4,3x3) with the positions split over two waves (csrc/conv_wino44s.hip): parity + times next to conv_wino44.hip;
#   (4) stride-2 pairs once more with the "k x 1 half only" mode.
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s2
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3), 'host cpu ms', round(d['host_cpu_ms_per_keyframe'],2), 'sum of kernels', round(d['device_ms_per_step_sum_of_kernels'],3))"; }
for rep in 1 2 3; do
  (cd ab_r4 && timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r4-tree 200")
  timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r5-tree 200"
done
(cd ab_r4 && timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api --step-times --host-prime-ms 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r4-tree 20 steps unprimed', round(d['value'],1), d['step_parts_ms_prepare_wait_submit'][:8], d['step_marks_ms'][-3:])")
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api --step-times 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5-tree 20 steps unprimed', round(d['value'],1), d['step_parts_ms_prepare_wait_submit'][:8], d['step_marks_ms'][-3:])"
(cd ab_r4 && timeout 200 python bench.py --steps 200 --in-flight 1 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r4-tree in-flight 1")
timeout 200 python bench.py --steps 200 --in-flight 1 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r5-tree in-flight 1"
# (2) whole suite
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -25 $OUT/suite.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -s -k "relaxed_cost_volume or separable" -p no:cacheprovider 2>&1 | grep -E "relaxed|separable|passed|failed" | cut -c1-300
# (3) F(4x4,3x3): one workgroup per CU (31) vs positions split over two waves, two workgroups per CU (41)
timeout 300 python tools/bench_wino.py --codes 11,31,41 --min-pixels 32768 > $OUT/wino_c2.log 2>&1; echo "wino c2 rc=$?"; grep -o '"name": "[a-z0-9.]*"\|"direct_us": [0-9.]*\|"wino[0-9]*_us": [0-9.]*\|"wino41_maxdiff": [0-9.e-]*' $OUT/wino_c2.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
timeout 400 python tools/bench_wino.py --batch 8 --frames 4 --depths 64 --codes 31,41 --min-pixels 32768 > $OUT/wino_c3.log 2>&1; echo "wino c3 rc=$?"; grep -o '"name": "[a-z0-9.]*"\|"direct_us": [0-9.]*\|"wino[0-9]*_us": [0-9.]*\|"wino41_maxdiff": [0-9.e-]*' $OUT/wino_c3.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
# (4) stride-2 pairs with the y-only mode
timeout 300 python tools/bench_stride2.py > $OUT/stride2_c2.log 2>&1; grep -o '"name": "[a-z0-9.]*"\|"direct_us": \[[0-9., ]*\]\|"best": [0-9]*\|"s[0-9]0_us": \[[0-9., ]*\]' $OUT/stride2_c2.log | paste -sd' ' | sed 's/"name"/\n"name"/g'
