#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_bench_quick.sh || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --skip-unread-slack > gpurun_out/bench_skip.json || exit 1
python -c "
import json; d=json.load(open('gpurun_out/bench_skip.json')); print('skip-unread-slack: %.3f M solves/s' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
