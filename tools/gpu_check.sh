#!/bin/bash
# one GPU round trip: parity tests, bench (both modes), section profile.  Stops at the first failure.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/bench_ref.json
cat gpurun_out/bench_ref.json
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --mode paper > gpurun_out/bench_paper.json
cat gpurun_out/bench_paper.json
HOQP_DBG=1 HOQP_LIB=$PWD/quadruped_locomotion_amd/libhoqp_hip_prof.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2> gpurun_out/prof_ref.txt > /dev/null
grep "hoqp profile" gpurun_out/prof_ref.txt | tail -1
HOQP_DBG=1 HOQP_LIB=$PWD/quadruped_locomotion_amd/libhoqp_hip_prof.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --mode paper 2> gpurun_out/prof_paper.txt > /dev/null
grep "hoqp profile" gpurun_out/prof_paper.txt | tail -1
