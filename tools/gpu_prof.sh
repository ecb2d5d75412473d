#!/bin/bash
# section profile only (profiling build of the library)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for mode in reference paper; do
HOQP_DBG=1 HOQP_LIB=$PWD/quadruped_locomotion_amd/libhoqp_hip_prof.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --latency-batches 2 --mode $mode 2> gpurun_out/prof_$mode.txt > /dev/null
grep "hoqp profile" gpurun_out/prof_$mode.txt | tail -2
done
