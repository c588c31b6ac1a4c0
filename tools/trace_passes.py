#!/usr/bin/env python3
"""Band-CNN launches of a rocprofv3 --kernel-trace CSV, separated into the ENCODE and the DECODE pass (`kernel_ms.encode.cnn`
10.8 ms against `decode.cnn` 10.0 ms for identical FLOPs): a launch belongs to the pass of the kernel that runs BEHIND it on the queue -- the
encoder's are followed by cdf_pairs* (or by the next band's CNN launch of the same level and then cdf_pairs_bands), the decoder's by
rans_decode_stage* / ac_* / cdf_table / cdf_anchor.  Output: per (pass, kernel, grid) count, mean, min, total, and the totals per pass.
Usage: python tools/trace_passes.py <..._kernel_trace.csv> [out.json]"""
import collections
import csv
import json
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])) for r in rows]


def pass_of(i):
    for j in range(i + 1, min(i + 8, len(rows))):
        n = names[j]
        if n.startswith("cdf_pairs"):
            return "encode"
        if n.startswith(("rans_decode_stage", "rans_tail", "ac_decode", "cdf_table", "cdf_anchor")):
            return "decode"
        if not n.startswith("band_params_kernel"):
            return "other"
    return "other"


g = collections.defaultdict(list)
for i, r in enumerate(rows):
    if "band_params_kernel" not in names[i]:
        continue
    key = (pass_of(i), names[i][:48], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"groups": [], "totals_ms": collections.defaultdict(float), "launches": collections.defaultdict(int)}
for k, v in sorted(g.items()):
    out["groups"].append({"pass": k[0], "kernel": k[1], "grid_x": k[2], "wg": k[3], "n": len(v), "mean_us": round(sum(v) / len(v), 2),
                          "min_us": round(min(v), 2), "total_ms": round(sum(v) / 1e3, 3)})
    out["totals_ms"][k[0]] += sum(v) / 1e3
    out["launches"][k[0]] += len(v)
out["totals_ms"] = {k: round(v, 3) for k, v in out["totals_ms"].items()}
out["launches"] = dict(out["launches"])
pairs = collections.defaultdict(dict)
for gr in out["groups"]:
    pairs[(gr["kernel"], gr["grid_x"], gr["wg"])][gr["pass"]] = gr["mean_us"]
out["encode_over_decode_mean"] = [{"kernel": k[0], "grid_x": k[1], "wg": k[2], **v, "ratio": round(v["encode"] / v["decode"], 4)}
                                  for k, v in sorted(pairs.items()) if "encode" in v and "decode" in v]
print(json.dumps({k: out[k] for k in ("totals_ms", "launches", "encode_over_decode_mean")}, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
