#!/usr/bin/env python3
"""Ideal code length of the REFERENCE's own tables on full-size images of the bench workload (build container only).

The small fixtures of make_fixtures.py (1-6 kpixel) cannot show "bpp within 0.001 of reference" on the headline workload: with the
sigma-floor seed-1337 weights one probability-1/65536 symbol whose table entry moves by one count is worth a whole bit, i.e.
2e-4 .. 1e-3 bpp on such an image by itself.  This script runs the reference-owned code (same stand-ins as make_fixtures.py: the
torchac stand-in here only SUMS log2(65536 / (c_high - c_low)) of what the reference hands to the coder and keeps nothing) on
  - image 0 of bench.py's batch (768x512 uniform noise, seed 0) and
  - BASELINE.json configs[0]'s 256x256 image (seed 0),
with the seed-1337 weights, and (round 4) on
  - one full-size NATURAL-LIKE image: 768x512 "smooth" RGB (make_fixtures.make_image, seed 11) with the "trained-like" weights
    (make_fixtures.trained_like_: sigma of a few grey levels) -- the content class the sigma-floor noise workload says nothing about,
  - one full-size image DRAWN FROM THE MODEL (tests/helpers.make_sampled_image: the reference-format decoder fed random bytes, trained-like
    weights): cheap symbols over the full value range, the class a well-trained model sees,
and writes tests/golden/ref_ideal_bits.json: the 45 per-stream ideal bit counts per image.  Data only.
tests/test_oracle_golden.py::test_bpp_delta_vs_reference_tables_report compares the oracle's tables against them."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402


class BitSummer:
    def __init__(self):
        self.bits = []

    def append(self, pair):
        cdf, sym = pair
        cdf = cdf[0, 0].view(np.uint16)
        sym = sym[0, 0]
        hh, ww, Lp = cdf.shape
        n = hh * ww
        flat = cdf.reshape(n, Lp)
        s = sym.reshape(n).astype(np.int64)
        lo = flat[np.arange(n), s].astype(np.int64)
        hi = flat[np.arange(n), s + 1].astype(np.int64)
        hi[s == Lp - 2] = 0x10000
        assert (hi > lo).all()
        self.bits.append(float(np.log2(65536.0 / (hi - lo)).sum()))


def main():
    rec = BitSummer()
    mf._install_standins(rec)
    sys.path.insert(0, mf.REF)
    from graphs.models.LLICTI_nets import LLICTI  # noqa: E402  (reference-owned code)
    cfg = mf.Cfg(json.load(open(os.path.join(mf.REF, "configs", "llicti_A.json"))))
    torch.use_deterministic_algorithms(True)
    torch.set_num_threads(8)
    out = {}
    models = {}
    for wname in ("rand1337", "trainedlike"):
        torch.manual_seed(1337)
        models[wname] = LLICTI(cfg).eval()
        if wname == "trainedlike":
            mf.trained_like_(models[wname])
    for name, kind, H, W, seed, wname in (("bench_image0_768x512", "noise", 512, 768, 0, "rand1337"), ("configs0_256x256", "noise", 256, 256, 0, "rand1337"),
                                          ("natural_like_768x512", "smooth", 512, 768, 11, "trainedlike"),
                                          ("model_sampled_768x512", "sampled", 512, 768, 5, "trainedlike")):
        if kind == "sampled":
            sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
            sys.path.insert(0, os.path.dirname(HERE))
            from helpers import make_sampled_image  # noqa: E402
            rgb = make_sampled_image(H, W, seed)
        else:
            rgb = mf.make_image(kind, H, W, seed)
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255.0)).unsqueeze(0)
        rec.bits = []
        with torch.no_grad():
            bl, _ = models[wname].compress(x.clone())
        assert len(rec.bits) == 45
        hdr = sum(len(s) for s in bl[0])
        out[name] = {"H": H, "W": W, "seed": seed, "kind": kind, "weights": wname, "header_bytes": hdr,
                     "ideal_bits_per_stream": rec.bits, "ideal_bits": float(np.sum(rec.bits))}
        print(name, "ideal bpp", (out[name]["ideal_bits"] + 8 * hdr) / (H * W))
    json.dump(out, open(os.path.join(HERE, "ref_ideal_bits.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
