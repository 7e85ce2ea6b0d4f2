#!/bin/bash
# The c2 bench line once more with the frac_executed key.
mkdir -p gpurun_out/s35
timeout 150 python bench.py --steps 200 > gpurun_out/s35/bench_c2.json 2>/dev/null; tail -1 gpurun_out/s35/bench_c2.json | cut -c1-200
