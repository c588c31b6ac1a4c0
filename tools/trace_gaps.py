#!/usr/bin/env python3
"""Idle time between consecutive kernel dispatches of a rocprofv3 --kernel-trace CSV (one in-order queue): for every dispatch the gap between the
previous dispatch's end and its start, summed per predecessor kernel.  Gaps above --max-gap-us (host-side pauses between legs) are left out.
Usage: python tools/trace_gaps.py <..._kernel_trace.csv> [out.json] [max_gap_us=200]"""
import csv, json, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
mx = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:48]
gap_by = collections.defaultdict(lambda: [0, 0.0])
tot_gap = tot_busy = 0.0
n = 0
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None:
        g = (s - prev_end) / 1e3
        if 0 <= g <= mx:
            k = name(rows[i - 1]) + " -> " + name(r)
            gap_by[k][0] += 1; gap_by[k][1] += g
            tot_gap += g; n += 1
    tot_busy += (e - s) / 1e3
    prev_end = max(prev_end or 0, e)
out = {"dispatches": len(rows), "gaps_counted": n, "total_gap_ms": round(tot_gap / 1e3, 3), "total_busy_ms": round(tot_busy / 1e3, 3),
       "gap_frac_of_busy": round(tot_gap / tot_busy, 4), "mean_gap_us": round(tot_gap / max(n, 1), 2),
       "by_transition": [{"transition": k, "n": v[0], "mean_us": round(v[1] / v[0], 2), "total_ms": round(v[1] / 1e3, 3)}
                         for k, v in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:40]]}
print(json.dumps({k: v for k, v in out.items() if k != "by_transition"}))
for t in out["by_transition"][:30]: print(t)
if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], "w"), indent=1)
