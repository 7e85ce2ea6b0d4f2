#!/bin/bash
# Round 3, session 4: kernel timelines (rocprofv3 --kernel-trace, csv) of four host-loop configurations, to see where two keyframes
# overlap and where the streams run dry.  Analysis offline (tools/timeline_overlap.py).
OUT=$(pwd)/gpurun_out/r03_s4
mkdir -p $OUT
REPO=$(pwd)
python bench.py --steps 30 --no-cpu-baseline --no-forward-api > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "--caller-stream --queue-depth 1 --host-mats" "--queue-depth 1 --host-mats" "--queue-depth 2" "--caller-stream --queue-depth 2" "--caller-stream --queue-depth 1 --host-mats --in-flight 1"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -o t -- python $REPO/bench.py --steps 40 --warmup 5 --spinup-seconds 1 --no-cpu-baseline --no-primer --no-forward-api --hw-queues 16 $cfg > $OUT/t$i.json 2> $OUT/t$i.err
  echo "cfg $i: $cfg"; tail -1 $OUT/t$i.json | cut -c1-160
  f=$(find $OUT/t$i -name "*kernel_trace.csv" | head -1)
  python - "$f" $OUT/t$i.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
# keep the last ~6000 dispatches, compact columns
keep=rows[-6000:]
w=csv.writer(open(sys.argv[2],"w"))
w.writerow(["queue","stream","kernel","start","end"])
for r in keep:
    w.writerow([r.get("Queue_Id"), r.get("Stream_Id",""), r["Kernel_Name"][:60], r["Start_Timestamp"], r["End_Timestamp"]])
PY
  rm -rf $OUT/t$i
done
ls -la $OUT
