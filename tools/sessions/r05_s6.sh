#!/bin/bash
# Round 5, session 6: r05_s5 - the creation order of the streams is NOT it (every order with the parse-only first token: 683-710; without: 753-757); what the
# parse-only token changes is the order in which the streams are first USED (encoder stream of slot 0 before its main stream), and ROCm binds a stream to a
# hardware queue at first use.  The model now gives every stream one 4-byte launch at creation: which first-use order is best?
cd "$(dirname "$0")/../.." || exit 1
OUT=$(pwd)/gpurun_out/r05_s6
mkdir -p $OUT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s; host enqueue ms', round(d['host_enqueue_ms'],3))"; }
run() { MR_DIAG_STREAM_LAYOUT="$1" timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "[$1]"; }
run "g,m0,e0,m1,e1"
MR_DIAG_LAZY_PREPARE=0 MR_DIAG_STREAM_LAYOUT="g,m0,e0,m1,e1" timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "[g,m0,e0,m1,e1] no parse-only token"
run "g,e0,m0,m1,e1"
run "m0,e0,m1,e1,g"
run "m0,m1,e0,e1,g"
run "g,m0,m1,e0,e1"
run "g,m0,e0,_,m1,e1"
run "g,_,m0,e0,m1,e1"
run "g,m0,_,e0,_,m1,_,e1"
run "g,m0,e0,e1,m1"
run "m0,g,e0,m1,e1"
run "g,_,_,m0,e0,_,_,m1,e1"
run "g,m0,e0,m1,e1"
(cd ab_r4 && timeout 200 python bench.py --steps 200 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "r4-tree")
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "default, 20 steps"
MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "default, 20 steps, no parse-only token"
timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "default, 20 steps"
MR_DIAG_LAZY_PREPARE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-primer --no-cpu-baseline --no-forward-api 2>/dev/null | line "default, 20 steps, no parse-only token"
