#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files (one directory per pass) into one per-kernel table.
usage: pmc_summary.py OUT.csv DIR [DIR...]"""
import collections
import csv
import glob
import sys

out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for d in dirs:
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[(k, d)].add(r["Dispatch_Id"])
counters = sorted({c for v in agg.values() for c in v})
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches_per_pass"] + counters)
    for k in sorted(agg):
        n = max(len(v) for (kk, _), v in ndisp.items() if kk == k)
        w.writerow([k, n] + [f"{agg[k].get(c, 0):.6g}" for c in counters])
print("wrote", out)
