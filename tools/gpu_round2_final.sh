#!/bin/bash
# end-of-round measurements: rocprofv3 passes + section profiles (tools/profile.sh), then the other bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile.sh > gpurun_out/profile_r2.log 2>&1 || { tail -5 gpurun_out/profile_r2.log; exit 1; }
tail -3 gpurun_out/profile_r2.log
timeout -k 10 300 python bench.py --no-cpu-baseline --contact three > gpurun_out/rocprof/bench_three.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --skip-unread-slack > gpurun_out/rocprof/bench_skip.json || exit 1
timeout -k 10 500 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/rocprof/bench_c4.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 65536 --steps 50 > gpurun_out/rocprof/bench_65536.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --pipeline-streams 2 > gpurun_out/rocprof/bench_pipelined.json || exit 1
for f in full paper trot three skip c4 65536; do python - "$f" <<'PY'
import json, sys
d = json.load(open("gpurun_out/rocprof/bench_%s.json" % sys.argv[1]))
print(sys.argv[1], "%.3f M solves/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"])
PY
done
