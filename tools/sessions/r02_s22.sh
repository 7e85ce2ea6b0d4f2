#!/bin/bash
# Marching cost volume at c2 with one plane per wave: row-segment lengths around the automatic choice, keyframe statistics inline.
cv() { echo "cv $1: $(env $2 timeout 120 python tools/bench_cv.py --impl march $3 2>/dev/null | tail -1)"; }
cv "c2 default (DP=1, TY auto)" "A=1" ""
cv "c2 no prepass" "MR_CV_NO_KF_PREPASS=1" ""
for ty in 22 26 29 32 43 52; do cv "c2 TY=$ty" "MR_CV_MARCH_TY=$ty" ""; done
cv "c2 TY=32 no prepass" "MR_CV_MARCH_TY=32 MR_CV_NO_KF_PREPASS=1" ""
MR_CV_NO_KF_PREPASS=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "marching" 2>&1 | tail -1
