#!/bin/bash
# Round 6, session 3: MFMA issue-rate probe (tools/probes/mfma_rates.hip) - what costs the 50 cycles per fp32 MFMA of the product's sweeps?
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06_s3
timeout 120 tools/probes/mfma_rates | tee gpurun_out/r06_s3/mfma_rates.txt
