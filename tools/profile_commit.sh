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
