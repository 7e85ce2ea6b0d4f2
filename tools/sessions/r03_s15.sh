#!/bin/bash
# marching cost volume after the VGPR drop (59 / 95): row-segment length TY and planes per wave, c2 / c3 / configs[4] shape
for shape in "" "--batch 8 --frames 4 --depths 64 --iters 20" "--height 512 --width 1024 --frames 4 --depths 48 --iters 50"; do
  echo "== $shape"
  for dp in 0 1 2; do for ty in 0 12 16 19 21 24 28 32 37 43 52; do
    r=$(MR_CV_MARCH_TY=$ty MR_CV_MARCH_DP=$dp timeout 60 python tools/bench_cv.py --impl march $shape 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['march_us'])")
    echo -n "dp$dp ty$ty: $r | "
  done; echo; done
done
