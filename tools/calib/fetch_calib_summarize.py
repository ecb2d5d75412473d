#!/usr/bin/env python3
"""counter / known-bytes ratios of tools/calib/fetch_calib.sh -> one JSON object (kept as profiles/rNN_fetch_calib.json)."""
import csv, glob, json, sys

src = sys.argv[1]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) against the byte count a streaming kernel is known to move, "
               "1 GiB buffer (4 x the Infinity Cache), last 3 of 4 launches; factor = known bytes / reported bytes",
       "modes": {"8": "8 B per lane, 512 contiguous bytes per load instruction",
                 "16": "16 B per lane, 1024 contiguous bytes per load instruction",
                 "1": "one wavefront per 4,616-byte block (not line aligned), 8 B per lane: the fused WBC entry's pattern"}}
for mode in ("8", "16", "1"):
    known = json.load(open(f"{src}/known_{mode}.json"))
    row = dict(known)
    for ctr, key in (("FETCH_SIZE", "read_bytes_per_launch"), ("WRITE_SIZE", "write_bytes_per_launch")):
        vals = []
        for fn in glob.glob(f"{src}/{ctr}_{mode}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith("read"):
                    vals.append(float(r["Counter_Value"]))
        vals = vals[-3:]
        if vals:
            rep = sum(vals) / len(vals) * 1024.0
            row[ctr + "_bytes_reported"] = rep
            row[ctr + "_factor"] = known[key] / rep if rep else None
    res[mode] = row
print(json.dumps(res, indent=1))
