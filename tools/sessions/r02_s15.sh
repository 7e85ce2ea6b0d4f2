#!/bin/bash
# Final-state profiles of round 2 (same commands as r02_s7, after the wave-K-split / K=32 / batching changes).
bash tools/profile_round.sh r02_c2
bash tools/profile_round.sh r02_c3 "--batch 8 --frames 4 --depths 64 --no-cpu-baseline" 12
