#!/usr/bin/env python3
"""Known-answer vectors of the LLICTI-rANS v3 container (a format of THIS build: there is no reference counterpart).

Writes tests/golden/rans_v3_vectors.npz: for three golden cases (images + weights already committed as fixtures generated
from the reference) and M in {1, 4} -- plus M = 3 WIDE streams (128 lanes each), keys ..._W3_..., and (round 4) M = 3 XWIDE streams (256 lanes each),
keys ..._X3_... -- the container bytes the CPU oracle
produces and their SHA-256.  The format is frozen by
these bytes: tests/test_oracle_golden.py::test_rans_v3_known_answer fails if the oracle's output for the same inputs ever
changes (an accidental format change), and the GPU suite holds the HIP path to the oracle byte for byte.
Run from the repo root:  python tests/golden/make_rans_v3_vectors.py"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from llicti_amd.weights import pack_state_dict
from oracle import oracle as orc

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = [("smooth_67x93_tl", "trainedlike"), ("noise_32x32_rand", "rand1337"), ("noise_33x64_tl", "trainedlike")]
out = {}
for case, wname in CASES:
    rgb = np.load(os.path.join(GOLDEN, f"case_{case}.npz"))["rgb"]
    W = orc.Weights(pack_state_dict(dict(np.load(os.path.join(GOLDEN, f"weights_{wname}.npz")))))
    for key, M, wide in (("M1", 1, 0), ("M4", 4, 0), ("W3", 3, 1), ("X3", 3, 2)):
        bl = orc.encode_image_rans(rgb, W, M, wide)
        assert np.array_equal(orc.decode_image_rans(bl, W), rgb)
        flat = b"".join(s for row in bl for s in row)
        out[f"{case}_{key}_bytes"] = np.frombuffer(flat, np.uint8)
        out[f"{case}_{key}_seglen"] = np.array([len(s) for row in bl for s in row], np.int32)
        out[f"{case}_{key}_sha256"] = np.frombuffer(hashlib.sha256(flat).digest(), np.uint8)
        print(case, key, len(flat), hashlib.sha256(flat).hexdigest()[:16])
np.savez_compressed(os.path.join(GOLDEN, "rans_v3_vectors.npz"), **out)
