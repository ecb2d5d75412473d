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

# HBM traffic of the solve kernel per launch, in the form bench.py reads (profiles/rNN_pmc_traffic.json)
import hashlib, os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "quadruped_locomotion_amd", "libhoqp_hip.so")
lib_sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
# measured FETCH_SIZE / WRITE_SIZE factors for this kernel's access pattern (tools/calib/fetch_calib.sh), if present
calib = None
for fn in (os.path.join(root, "gpurun_out", "calib", "summary.json"), os.path.join(root, "profiles", "r03_fetch_calib.json")):
    if os.path.exists(fn):
        calib = json.load(open(fn))
        break
# the counter's own factor (8 B per lane, line-aligned stream: 2.0 on gfx950); the block pattern's 1.95 folds the
# line-granularity over-fetch of unaligned instance blocks into the factor, which is exactly what traffic should show
ff = calib["8"].get("FETCH_SIZE_factor", 1.0) if calib else 1.0
wf = calib["8"].get("WRITE_SIZE_factor", 1.0) if calib else 1.0
for k, cs in pmc.items():
    if ("hoqp_solve_wbc_ref_kernel" in k or "hoqp_solve_kernel" in k) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fk, wk = cs["FETCH_SIZE"]["mean_per_launch"], cs["WRITE_SIZE"]["mean_per_launch"]
        json.dump({
            "kernel": k, "batch": 4096, "semantics": "reference", "contact": "stance",
            "library_sha256": lib_sha,
            "fetch_bytes_per_launch": fk * 1024.0 * ff, "write_bytes_per_launch": wk * 1024.0 * wf,
            "fetch_size_kib": fk, "write_size_kib": wk, "fetch_factor": ff, "write_factor": wf,
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs of `bench.py --no-cpu-baseline "
                      "--steps 20 --warmup 3 --latency-batches 2`, tools/profile.sh), mean of the last 20 launches; KiB -> "
                      "bytes, times the factor measured for 8-B-per-lane loads by tools/calib/fetch_calib.sh on the same box "
                      "(profiles/r03_fetch_calib.json: FETCH_SIZE reports half the bytes, WRITE_SIZE all of them); "
                      "valid for the library whose SHA-256 is given",
        }, open(out + "_pmc_traffic.json", "w"), indent=1)
