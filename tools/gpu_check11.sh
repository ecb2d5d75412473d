#!/bin/bash
# parity tests + two bench sizes (used while tuning the level-1 QP)
timeout -k 10 600 python -u -m pytest tests -m gpu --maxfail=8 -q --timeout 300 --timeout-method thread > gpurun_out/r3_suite_11.log 2>&1
rc=$?; echo rc=$rc >> gpurun_out/r3_suite_11.log; tail -8 gpurun_out/r3_suite_11.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
for b in 4096 65536; do
  timeout -k 10 100 python bench.py --steps 100 --warmup 10 --batch $b --no-cpu-baseline --latency-batches 10 > gpurun_out/x_$b.json 2>/dev/null || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/x_$b.json').read().strip().splitlines()[-1]); print($b, d['value'], d['roofline']['kernel_ms'], d['solver']['iters_hist'])"
done
HOQP_DBG=1 HOQP_LIB=$PWD/quadruped_locomotion_amd/libhoqp_hip_prof.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --latency-batches 2 2> gpurun_out/sec_stance.txt > /dev/null
grep -h "hoqp profile" gpurun_out/sec_stance.txt | tail -8 | cut -c1-700
