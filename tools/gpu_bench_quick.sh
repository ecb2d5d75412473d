#!/bin/bash
# quick perf check: stance (headline), three, paper
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for c in "--contact stance" "--contact three" "--mode paper"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 $c > gpurun_out/bq.json || exit 1
  python - <<'PY'
import json
d=json.load(open("gpurun_out/bq.json"))
print(d["config"]["contact"], d["config"]["semantics"], "%.3f M solves/s" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"])
PY
done
