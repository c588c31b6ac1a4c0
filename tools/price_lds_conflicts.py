#!/usr/bin/env python3
"""what can the LDS bank conflicts of the coder kernels cost at most?  From the PMC passes (tools/collect_pmc.sh ->
pmc_*_summary.csv): SQ_LDS_BANK_CONFLICT counts the cycles an LDS pipe spends replaying conflicting accesses, summed over the compute units.
A coder workgroup has its compute unit (and so its LDS pipe) to itself, so conflict cycles / workgroups is the most a launch can lose on a CU
if EVERY replay cycle sits on the workgroup's critical path; divided by the launch's cycles (GRBM_GUI_ACTIVE / 8 XCDs / launches) it is an upper
bound of the kernel time a conflict-free layout could win back.  Usage: python tools/price_lds_conflicts.py <pmc_summary.csv> <out.json> <workgroups per launch>"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 240
out = {"method": __doc__.split("Usage")[0].strip(), "workgroups_per_launch": wgs, "kernels": {}}
for r in rows:
    k = r["kernel"]
    if not any(s in k for s in ("rans_decode_stage", "rans_encode", "rans_tail")):
        continue
    f = lambda c: float(r.get(c, 0) or 0)      # noqa: E731
    n = float(r["dispatches_per_pass"])
    cyc = f("GRBM_GUI_ACTIVE") / 8 / n
    conf_cu = f("SQ_LDS_BANK_CONFLICT") / n / wgs
    idx_cu = f("SQ_LDS_IDX_ACTIVE") / n / wgs
    out["kernels"][k] = {"launches_in_pass": n, "cycles_per_launch": round(cyc), "lds_active_cycles_per_cu_per_launch": round(idx_cu),
                         "bank_conflict_cycles_per_cu_per_launch": round(conf_cu), "conflict_share_of_lds_cycles": round(f("SQ_LDS_BANK_CONFLICT") / max(1.0, f("SQ_LDS_IDX_ACTIVE")), 3),
                         "lds_pipe_busy_share_of_launch": round(idx_cu / cyc, 4),
                         "upper_bound_share_of_kernel_time": round(conf_cu / cyc, 4)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
