#!/usr/bin/env python3
"""Known-answer vectors of the rANS containers (formats of THIS build: there is no reference counterpart).

Writes tests/golden/rans_vectors.npz: for three golden cases (images + weights already committed as fixtures generated from the reference)
the container bytes the CPU oracle produces, their segment lengths and their SHA-256, for
  M1, M4   1 / 4 streams of 64 lanes   (v3 layout, unchanged since round 3)
  W3       3 WIDE streams (128 lanes)  (v3 layout, unchanged since round 3)
  X4       3 XWIDE streams (256 lanes) in the v4 layout of round 6 (the X3 vectors of rounds 4-5 held the retired xwide v3 layout)
and -- hash and segment lengths only, the images come from tests/helpers.make_image -- for two images large enough to exercise what the
small fixtures cannot, a tail that fills its payload and spills:
  X4big_smooth   256x384 "smooth" (seed 11), trained-like weights, 4 xwide streams: one-chain tails of ~1,500 symbols
  X4big_noise    96x160 "noise" (seed 3), seed-1337 weights, 2 xwide streams: two-chain tails
The formats are frozen by these bytes: tests/test_oracle_golden.py::test_rans_known_answer fails if the oracle's output for the same inputs
ever changes (an accidental format change), and the GPU suite holds the HIP path to the oracle byte for byte.
Run from the repo root:  python tests/golden/make_rans_vectors.py"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from llicti_amd.weights import pack_state_dict
from oracle import oracle as orc
from helpers import make_image

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = [("smooth_67x93_tl", "trainedlike"), ("noise_32x32_rand", "rand1337"), ("noise_33x64_tl", "trainedlike")]
BIG = [("X4big_smooth", "smooth", 256, 384, 11, "trainedlike", 4), ("X4big_noise", "noise", 96, 160, 3, "rand1337", 2)]
out = {}


def weights(wname):
    return orc.Weights(pack_state_dict(dict(np.load(os.path.join(GOLDEN, f"weights_{wname}.npz")))))


for case, wname in CASES:
    rgb = np.load(os.path.join(GOLDEN, f"case_{case}.npz"))["rgb"]
    W = weights(wname)
    for key, M, wide in (("M1", 1, 0), ("M4", 4, 0), ("W3", 3, 1), ("X4", 3, 2)):
        bl = orc.encode_image_rans(rgb, W, M, wide)
        assert np.array_equal(orc.decode_image_rans(bl, W), rgb)
        flat = b"".join(s for row in bl for s in row)
        out[f"{case}_{key}_bytes"] = np.frombuffer(flat, np.uint8)
        out[f"{case}_{key}_seglen"] = np.array([len(s) for row in bl for s in row], np.int32)
        out[f"{case}_{key}_sha256"] = np.frombuffer(hashlib.sha256(flat).digest(), np.uint8)
        print(case, key, len(flat), hashlib.sha256(flat).hexdigest()[:16])
for key, kind, H, Wd, seed, wname, M in BIG:
    rgb = make_image(kind, H, Wd, seed)
    W = weights(wname)
    bl = orc.encode_image_rans(rgb, W, M, 2)
    assert np.array_equal(orc.decode_image_rans(bl, W), rgb)
    flat = b"".join(s for row in bl for s in row)
    out[f"{key}_seglen"] = np.array([len(s) for row in bl for s in row], np.int32)
    out[f"{key}_sha256"] = np.frombuffer(hashlib.sha256(flat).digest(), np.uint8)
    print(key, len(flat), hashlib.sha256(flat).hexdigest()[:16])
np.savez_compressed(os.path.join(GOLDEN, "rans_vectors.npz"), **out)
