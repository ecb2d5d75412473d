#!/bin/bash
# copies the summaries of the last tools/profile.sh run (gpurun_out/rocprof/) into profiles/ under a round prefix
set -e
cd "$(dirname "$0")/.."
P=${1:-r03}
S=gpurun_out/rocprof
cp $S/summary_kernel_stats.csv profiles/${P}_kernel_stats.csv
cp $S/summary_pmc.json profiles/${P}_pmc.json
cp $S/summary_pmc_traffic.json profiles/${P}_pmc_traffic.json
grep -h "hoqp profile" $S/sections_reference.txt | awk '!seen[$0]++' | tail -8 > profiles/${P}_sections_reference.txt
grep -h "hoqp profile" $S/sections_paper.txt | awk '!seen[$0]++' | tail -8 > profiles/${P}_sections_paper.txt
cp $S/bench_full.json profiles/${P}_bench.json
[ -f gpurun_out/calib/summary.json ] && cp gpurun_out/calib/summary.json profiles/${P}_fetch_calib.json
cp $S/bench_paper.json profiles/${P}_bench_paper.json
cp $S/bench_trot.json profiles/${P}_bench_trot.json
ls -la profiles/${P}_*
# the further outputs of tools/gpu_round3_final.sh, when present
for f in three skip c4 65536 pipelined crawl; do [ -f $S/bench_$f.json ] && cp $S/bench_$f.json profiles/${P}_bench_$f.json; done
for f in entry_rates shim_latency host_entry_rate stress_parity; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/${P}_$f.txt; done
if ls gpurun_out/configs/*.json > /dev/null 2>&1; then
python3 - "$P" <<'PY'
import glob, json, os, sys
out = {}
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    out[os.path.basename(f)[:-5]] = {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "config", "solver", "roofline", "cpu_baseline")}
json.dump(out, open("profiles/%s_configs.json" % sys.argv[1], "w"), indent=1)
PY
fi
ls profiles/${P}_* | wc -l
