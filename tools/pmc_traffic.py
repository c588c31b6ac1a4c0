#!/usr/bin/env python3
"""tools/pmc_traffic.py PMC_DIR OUT_DIR -- per-kernel PMC table + HBM traffic per launch.

Reads the rocprofv3 --pmc passes written by tools/collect_pmc.sh, writes OUT_DIR/pmc_summary.csv (all counters summed
over a kernel's dispatches, plus the dispatch count) and OUT_DIR/pmc_traffic.json (HBM bytes per launch of the kernels
bench.py reports a roofline for).  Units and corrections follow MI355X_MICROARCH.md "HBM": rocprofv3 reports FETCH_SIZE /
WRITE_SIZE in KiB; WRITE_SIZE is exact for wide streaming stores; FETCH_SIZE counts 128-byte requests at 64 bytes, i.e.
HALF the bytes of a wide coalesced streaming read -- it is doubled here (an upper bound for the CNN's 4-byte-per-lane
strided LDS-DMA reads, which the guide lists as uncalibrated; both raw and doubled values are kept)."""
import collections, csv, glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(f"{src}/*/*counter_collection.csv") + glob.glob(f"{src}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][f].add(r["Dispatch_Id"])
counters = sorted({c for v in agg.values() for c in v})
os.makedirs(out, exist_ok=True)
with open(os.path.join(out, "pmc_summary.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches_per_pass"] + counters)
    for k in sorted(agg):
        n = max(len(v) for v in disp[k].values())
        w.writerow([k, n] + [f"{agg[k].get(c, 0):.6g}" for c in counters])


def per_launch(prefix):
    ks = [k for k in agg if k.startswith(prefix)]
    n = sum(max(len(v) for v in disp[k].values()) for k in ks)
    fetch = sum(agg[k].get("FETCH_SIZE", 0) for k in ks) * 1024.0
    write = sum(agg[k].get("WRITE_SIZE", 0) for k in ks) * 1024.0
    d = {"launches": n, "fetch_bytes_raw_per_launch": fetch / n, "fetch_bytes_x2_per_launch": 2 * fetch / n,
         "write_bytes_per_launch": write / n, "hbm_bytes_per_launch": (2 * fetch + write) / n}
    mf, gui = sum(agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for k in ks), sum(agg[k].get("GRBM_GUI_ACTIVE", 0) for k in ks)
    bc, ba = sum(agg[k].get("SQ_LDS_BANK_CONFLICT", 0) for k in ks), sum(agg[k].get("SQ_LDS_IDX_ACTIVE", 0) for k in ks)
    wa, wc = sum(agg[k].get("SQ_WAIT_ANY", 0) for k in ks), sum(agg[k].get("SQ_WAVE_CYCLES", 0) for k in ks)
    if ba:
        d["lds_bank_conflict_frac"] = bc / ba
    if wc:
        d["wait_any_frac_of_wave_cycles"] = wa / wc
    d["note_mfma_busy_cycles_total"] = mf
    d["note_grbm_gui_active_total"] = gui
    iv, av, bz = (sum(agg[k].get(c, 0) for k in ks) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"))
    if iv:
        d["valu_insts_per_launch"] = iv / n
        if gui:
            # the chip issues one wave64 vector instruction per SIMD every 2 cycles: 1,024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            cyc = gui / 8.0 / n
            d["cycles_per_launch"] = cyc
            d["valu_issue_frac"] = (iv / n) / (cyc * 1024 * 0.5)
    return d


# a third argument merges into an existing pmc_traffic.json (the table kernel's passes are a separate command)
dst = os.path.join(out, "pmc_traffic.json")
res = json.load(open(dst)) if (len(sys.argv) > 3 and os.path.exists(dst)) else {}
res.setdefault("command", {})
if not isinstance(res["command"], dict):
    res["command"] = {"bench": res["command"]}
label = sys.argv[3] if len(sys.argv) > 3 else "bench"
res["command"][label] = ("python3 tools/bench_table.py" if label == "table" else
                         "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-ac-leg --no-pcie-legs") + " (one rocprofv3 --pmc pass per counter group)"
prefs = ("cdf_table_kernel",) if label == "table" else ("band_params_kernel", "rans_decode_stage_kernel", "rans_decode_stage_pair_kernel", "rans_decode_stage_lane_kernel", "rans_tail_kernel", "cdf_pairs_kernel", "lift_kernel", "cdf_table_kernel", "cdf_anchor_kernel",
             "ac_decode_kernel", "rans_encode_kernel", "ac_encode_pairs_kernel")
for pref in prefs:
    if any(k.startswith(pref) for k in agg):
        res[pref] = per_launch(pref)
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "command"}, indent=1)[:3000])
