#!/usr/bin/env python3
"""Per-dispatch durations of a rocprofv3 --kernel-trace CSV, grouped by kernel and grid size (the CNN launches of the five levels differ in
grid size or tile form): count, mean, min, total.  Usage: python tools/trace_levels.py <..._kernel_trace.csv> [out.json]"""
import csv, json, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
    key = (name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    out.append({"kernel": k[0], "grid_x": k[1], "grid_y": k[2], "wg": k[3], "n": len(v), "mean_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2), "total_ms": round(sum(v) / 1e3, 3)})
for o in out[:60]: print(o)
if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], "w"), indent=1)
