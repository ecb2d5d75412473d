#!/bin/bash
# end-of-round measurements: rocprofv3 passes + section profiles (tools/profile.sh), the other bench lines,
# the SURVEY 8d configurations, entry rates, shim latency, host-entry rate and the parity sweep.
# Results under gpurun_out/rocprof/ and gpurun_out/configs/; tools/profile_commit.sh r03 copies them to profiles/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile.sh > gpurun_out/profile_r3.log 2>&1 || { tail -n 5 gpurun_out/profile_r3.log; exit 1; }
tail -n 3 gpurun_out/profile_r3.log
R=gpurun_out/rocprof
timeout -k 10 300 python bench.py --no-cpu-baseline --contact three > $R/bench_three.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --skip-unread-slack > $R/bench_skip.json || exit 1
timeout -k 10 500 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $R/bench_c4.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 65536 --steps 50 > $R/bench_65536.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --pipeline-streams 2 > $R/bench_pipelined.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --contact crawl > $R/bench_crawl.json || exit 1
for f in full paper trot three skip c4 65536 crawl; do python - "$f" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/rocprof/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.3f M solves/s" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"])
PY
done
bash tools/gpu_configs.sh > $R/configs.log 2>&1 || { tail -n 5 $R/configs.log; exit 1; }
cat $R/configs.log | cut -c1-200
timeout -k 10 300 python tools/entry_rates.py > $R/entry_rates.txt 2>&1 || { tail -n 5 $R/entry_rates.txt; exit 1; }
timeout -k 10 300 python tools/shim_latency.py > $R/shim_latency.txt 2>&1 || { tail -n 5 $R/shim_latency.txt; exit 1; }
timeout -k 10 300 python tools/host_entry_rate.py > $R/host_entry_rate.txt 2>&1 || { tail -n 5 $R/host_entry_rate.txt; exit 1; }
tail -n 4 $R/entry_rates.txt $R/shim_latency.txt $R/host_entry_rate.txt
timeout -k 10 900 python -u tests/stress_parity.py 16384 700 > $R/stress_parity.txt 2>&1 || { tail -n 8 $R/stress_parity.txt; exit 1; }
tail -n 12 $R/stress_parity.txt
echo final done
