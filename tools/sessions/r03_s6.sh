#!/bin/bash
# Round 3, session 6: why is forward() slower on a model with two slots?  Plain runs, then timelines (kernels + memory copies).
OUT=$(pwd)/gpurun_out/r03_s6
mkdir -p $OUT
REPO=$(pwd)
for cfg in "--in-flight 1" "--in-flight 2" "--in-flight 1 --host-mats" "--in-flight 2 --host-mats"; do
  echo "cfg: $cfg"; python tools/trace_forward.py $cfg 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "--in-flight 1" "--in-flight 2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t$i -o t -- python $REPO/tools/trace_forward.py $cfg > $OUT/t$i.log 2>&1
  tail -1 $OUT/t$i.log
  for kind in kernel_trace memory_copy_trace; do
    f=$(find $OUT/t$i -name "*${kind}.csv" | head -1)
    [ -n "$f" ] && tail -n 1500 "$f" > $OUT/t${i}_$kind.csv && head -1 "$f" > $OUT/t${i}_$kind.hdr
  done
  rm -rf $OUT/t$i
done
ls -la $OUT
