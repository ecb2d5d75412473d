#!/bin/bash
# two back-to-back bench runs per mode (run-to-run noise)
set -e
cd "$(dirname "$0")/.."
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref', round(d['value']), d['roofline']['kernel_ms'])"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --mode paper | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('paper', round(d['value']), d['roofline']['kernel_ms'])"
done
