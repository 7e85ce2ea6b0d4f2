#!/bin/bash
for cfg in "--in-flight 2" "--in-flight 2 --host-mats" "--in-flight 1"; do
  echo "cfg: $cfg"; python tools/trace_forward.py $cfg 2>&1 | grep -E "forward\(\)|ms/forward" | cut -c1-170
done
