#!/bin/bash
# the other SURVEY 8d configurations (C3 trot mix 65536, C5 three-contact paper, large batch)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/configs
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" > gpurun_out/configs/$name.json; python3 -c "
import json,sys; d=json.load(open('gpurun_out/configs/$name.json')); print('$name', round(d['value']), 'solves/s', round(d['ms_per_step'],3), 'ms/step', d['solver'])"; }
run c2_ref
run c2_paper --mode paper
run c3_trot65536_ref --batch 65536 --contact trot --seed 2345
run c3_trot65536_paper --batch 65536 --contact trot --seed 2345 --mode paper
run c2_paper65536 --batch 65536 --mode paper
run c5_three4096_paper --contact three --seed 4567 --mode paper --mu 0.2 --tau-max 40
run c5_three65536_paper --batch 65536 --contact three --seed 4567 --mode paper --mu 0.2 --tau-max 40
run c5_three4096_paper_defaults --contact three --seed 4567 --mode paper
run c4_shard131072_ref --batch 131072 --seed 3456
# CPU baseline detail: all-core throughput on C2 + single-thread latency on C1
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --cpu-detail > gpurun_out/configs/c2_ref_cpu_detail.json
python3 -c "
import json; d=json.load(open('gpurun_out/configs/c2_ref_cpu_detail.json')); print('cpu_baseline', d['cpu_baseline'])"
