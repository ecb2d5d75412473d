#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration passes on the GPU box (one counter per rocprofv3 run).  Output: gpurun_out/calib/
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
EXE=$ROOT/tools/calib/fetch_calib
cd /tmp
for mode in 8 16 1; do
  $EXE $mode 1024 4 > $OUT/known_$mode.json
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $OUT/${ctr}_$mode -- $EXE $mode 1024 4 > $OUT/${ctr}_$mode.log 2>&1
    echo $ctr $mode done
  done
done
cd $ROOT
python3 tools/calib/fetch_calib_summarize.py $OUT > $OUT/summary.json
cat $OUT/summary.json
