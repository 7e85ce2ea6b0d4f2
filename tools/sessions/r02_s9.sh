#!/bin/bash
# Driver-style line with the secondary bf16x3 key; the multi-rank branch of bench.py on one device (gloo); c3 / c5-shape re-tune.
OUT=gpurun_out/s9
mkdir -p $OUT
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], d.get('secondary_bf16x3'), d['cpu_baseline']['value'], d['with_data_loading']['value'])"
MR_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "2-rank rc=$?"; tail -1 $OUT/bench_2rank.json | cut -c1-400
cp monorec_amd/tuned_schedules.json $OUT/table_K.json
b() { timeout 400 python bench.py --steps $3 --no-cpu-baseline --no-primer $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'kf/s', 'conv TF', round(d['roofline']['achieved'],1))"; }
b "c3 before" "--batch 8 --frames 4 --depths 64" 100
b "c5 before" "--height 512 --width 1024 --frames 4 --depths 48" 60
timeout 900 python tools/tune_conv.py --merge --batch 8 --frames 4 --depths 64 > $OUT/tune_c3.log 2>&1; tail -1 $OUT/tune_c3.log
timeout 900 python tools/tune_conv.py --merge --height 512 --width 1024 --frames 4 --depths 48 > $OUT/tune_c5.log 2>&1; tail -1 $OUT/tune_c5.log
cp monorec_amd/tuned_schedules.json $OUT/table_K35.json
b "c3 after" "--batch 8 --frames 4 --depths 64" 100
b "c5 after" "--height 512 --width 1024 --frames 4 --depths 48" 60
