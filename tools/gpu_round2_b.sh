#!/bin/bash
# GPU round trip: all GPU tests, short bench, then (optional, $1 = states per case) the random parity sweep
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/bench_ref.json && cat gpurun_out/bench_ref.json || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --contact three > gpurun_out/bench_three.json && cut -c1-200 gpurun_out/bench_three.json || exit 1
if [ -n "$1" ]; then
  timeout -k 10 900 python -u tests/stress_parity.py $1 ${2:-600} > gpurun_out/stress_r2_$1_${2:-600}.txt 2>&1
  cat gpurun_out/stress_r2_$1_${2:-600}.txt | cut -c1-420
fi
