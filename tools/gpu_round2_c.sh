#!/bin/bash
# bench lines of the round: default (full, with CPU legs), c4 at N = 1, rehearsal test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_shim.py -m gpu -x -q --timeout 400 --timeout-method thread 2>&1 | tail -15 || exit 1
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json || exit 1
cat gpurun_out/bench_default.json
timeout -k 10 500 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c4.json || exit 1
cut -c1-900 gpurun_out/bench_c4.json
