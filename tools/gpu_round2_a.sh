#!/bin/bash
# GPU round trip: all GPU tests (fused entry, no-alloc entry), then a short bench of each WBC entry.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests tests/wip_no_alloc.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/bench_ref.json && cat gpurun_out/bench_ref.json
