#!/bin/bash
# where does the host wait inside forward()?
for cfg in "--in-flight 1" "--in-flight 2" "--in-flight 2 --host-mats" "--in-flight 3"; do
  echo "cfg: $cfg"; python tools/trace_forward.py $cfg 2>&1 | tail -7
done
