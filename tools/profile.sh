#!/bin/bash
# rocprofv3 passes over `bench.py` on the GPU box; raw output under gpurun_out/rocprof/.
# One counter group per run (the pool refuses --pmc together with trace domains).
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/rocprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 --latency-batches 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
echo trace done
pmc() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1; echo $name done; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
pmc sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
cd $ROOT
# what FETCH_SIZE / WRITE_SIZE report for this kernel's access pattern (factors applied by the summary)
bash tools/calib/fetch_calib.sh > gpurun_out/rocprof/calib.log 2>&1 || echo "calibration failed"
python3 tools/profile_summarize.py gpurun_out/rocprof gpurun_out/rocprof/summary
# full bench line (with the CPU baseline) and the per-section cycle profile of the profiling build; the bench
# line carries the traffic just measured (bench.py reads it from profiles/, keyed by the library's SHA-256)
cp gpurun_out/rocprof/summary_pmc_traffic.json profiles/r03_pmc_traffic.json
timeout -k 10 400 python3 bench.py > gpurun_out/rocprof/bench_full.json
HOQP_DBG=1 HOQP_LIB=$PWD/quadruped_locomotion_amd/libhoqp_hip_prof.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --latency-batches 2 2> gpurun_out/rocprof/sections_reference.txt > /dev/null
HOQP_DBG=1 HOQP_LIB=$PWD/quadruped_locomotion_amd/libhoqp_hip_prof.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --latency-batches 2 --mode paper 2> gpurun_out/rocprof/sections_paper.txt > /dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --mode paper > gpurun_out/rocprof/bench_paper.json
timeout -k 10 200 python3 bench.py --no-cpu-baseline --contact trot > gpurun_out/rocprof/bench_trot.json
echo all done
