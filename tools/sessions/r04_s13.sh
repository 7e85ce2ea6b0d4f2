#!/bin/bash
# Round 4, session 13: where does the pipelined regime lose time?  Kernel timeline (csv) of the c2 bench at 2 and at 1 keyframes in flight:
# idle fraction, overlap levels, gaps by size (tools/timeline_overlap.py).
cd "$(dirname "$0")/../.." || exit 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04_s13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for fl in 2 1 3; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$fl -o t -- python $REPO/bench.py --steps 200 --in-flight $fl --no-cpu-baseline --no-primer --no-forward-api > $OUT/trace$fl.log 2>&1
  f=$(find $OUT/tr$fl -name "*kernel_trace.csv" | head -1)
  echo "== in flight $fl ($f)"; tail -1 $OUT/trace$fl.log | cut -c1-200
  python $REPO/tools/timeline_overlap.py $f --tail 0.4 > $OUT/overlap_if$fl.json; cat $OUT/overlap_if$fl.json
done
find $OUT -name "*.csv" -size +20M -delete
