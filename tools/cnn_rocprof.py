#!/usr/bin/env python3
"""rocprofv3 *_kernel_stats.csv of `bench.py --steps S --warmup W --no-cpu-baseline --no-extras --no-pcie-legs [--no-ac-leg]` -> the band-CNN time
per step (encode + decode pass), for bench.py's `roofline.frac_rocprof`.  The run holds (1 check + W warm-up + 3 + 3 timed halves ... ) encode /
decode pairs; their number is taken from the trace itself: every pass has exactly 15 band-CNN launches and one lift (encode) or unlift (decode).
Usage: python tools/cnn_rocprof.py <..._kernel_stats.csv> <out.json> "<command>" """
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
cnn_ns = sum(float(r["TotalDurationNs"]) for r in rows if "band_params_kernel" in r["Name"])
cnn_calls = sum(int(r["Calls"]) for r in rows if "band_params_kernel" in r["Name"])
assert cnn_calls % 30 == 0, cnn_calls
pairs = cnn_calls // 30
out = {"cnn_ms_per_step": cnn_ns / 1e6 / pairs, "encode_decode_pairs_in_trace": pairs, "cnn_launches": cnn_calls,
       "kernel_stats": sys.argv[1].split("profiles/")[-1] if "profiles/" in sys.argv[1] else sys.argv[1],
       "command": sys.argv[3] if len(sys.argv) > 3 else None,
       "peak_tflops": 157.3, "flop_per_step": 2.0 * 193248 * 130944 * 24 * 2}
out["tflops"] = out["flop_per_step"] / (out["cnn_ms_per_step"] * 1e-3) / 1e12
out["frac"] = out["tflops"] / out["peak_tflops"]
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
