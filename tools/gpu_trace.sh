#!/bin/bash
# kernel-trace stats of the bench (per-kernel average durations)
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/trace_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OUT/log.txt 2>&1
head -4 $OUT/*/*kernel_stats.csv | cut -c1-150
