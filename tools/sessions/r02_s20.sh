#!/bin/bash
# Final-state profiles of round 2 after the one-channel kernels / one-plane marching cost volume (same commands as r02_s15).
bash tools/profile_round.sh r02_c2
bash tools/profile_round.sh r02_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12
