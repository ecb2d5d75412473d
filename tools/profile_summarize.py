#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools_profile.sh into the small files kept under profiles/.

usage: tools_profile_summarize.py <rocprof_dir> <out_prefix>
writes <out_prefix>_kernel_stats.csv, <out_prefix>_pmc.json
"""
import csv, glob, json, sys, collections

src, out = sys.argv[1], sys.argv[2]

# kernel-trace stats: the *_kernel_stats.csv rocprofv3 wrote, copied as is
stats = glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(out + "_kernel_stats.csv", "w", newline="") as f:
        csv.writer(f).writerows(rows)

pmc = {}
for name in ("fetch", "write", "sq1", "sq2", "sqc"):
    for fn in glob.glob(f"{src}/{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fn)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                # skip the warm-up launches: keep the last 20 dispatches
                v = v[-20:]
                pmc.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(pmc, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(d["mean_per_launch"], 1) for c, d in v.items()} for k, v in pmc.items()}, indent=1))
