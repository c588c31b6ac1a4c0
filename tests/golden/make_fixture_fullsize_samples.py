#!/usr/bin/env python3
"""Full-size pins of the reference's interpolator and tables (build container only).

All `get_params` / table fixtures of make_fixtures.py are <= 67x93 pixels: level 0 of a Kodak-size image -- interior tiles of the MFMA
kernel, tile borders, the image's borders and corners -- was only ever compared with the CPU oracle.  This script runs the reference-owned
code (same stand-ins as make_fixtures.py) on two full-size 768x512 images,
  - image 0 of bench.py's batch (uniform noise, seed 0) with the seed-1337 weights and
  - the natural-like image (make_fixtures.make_image("smooth", 512, 768, 11)) with the trained-like weights,
and stores, for levels 1 and 0 and each band, at ~160 positions (the four corners, border rows / columns, both sides of the 16-row x
32-column tile seams of the kernel, random interior positions):
  - the 60 outputs of LLICTIEntropyModel4.get_params (LLICTI_nets.py:822-825) at the position, float32, as returned (before the
    cross-channel mean update, which works in place on them),
  - per colour channel the coded symbol and 16 entries (index, value) of the int16 table row the reference handed to the coder
    (LLICTI_nets.py:938-983): the symbol's and its neighbours', both ends of the row, the rest evenly spread.
Data only (tests/golden/fullsize_samples.npz, ~0.5 MB).  tests/test_hip_parity.py::test_full_size_params_and_tables_vs_reference and
tests/test_oracle_golden.py::test_full_size_samples_vs_reference compare the HIP kernels / the oracle with them."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402

N_ENT = 16


def sample_positions(h, w, rng):
    pos = {(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1)}
    for k in range(8):
        pos.add((0, int(rng.integers(0, w)))); pos.add((h - 1, int(rng.integers(0, w))))
        pos.add((int(rng.integers(0, h)), 0)); pos.add((int(rng.integers(0, h)), w - 1))
        pos.add((1, int(rng.integers(0, w)))); pos.add((int(rng.integers(0, h)), w - 2))
    for k in range(24):          # both sides of the kernel's tile seams (16-row x 32-column tiles, also the 8- and 4-row forms' seams)
        i = min(h - 1, int(rng.integers(1, max(2, h // 4))) * 4 - int(rng.integers(0, 2)))
        j = min(w - 1, int(rng.integers(1, max(2, w // 32))) * 32 - int(rng.integers(0, 2)))
        pos.add((i, int(rng.integers(0, w)))); pos.add((int(rng.integers(0, h)), j)); pos.add((i, j))
    while len(pos) < 160:
        pos.add((int(rng.integers(0, h)), int(rng.integers(0, w))))
    return np.array(sorted(pos), dtype=np.int32)


class Recorder:
    def __init__(self):
        self.pairs = []

    def append(self, pair):
        self.pairs.append(pair)


IMAGES = (("noise0_rand1337", "noise", 0, "rand1337", 512, 768), ("smooth11_trainedlike", "smooth", 11, "trainedlike", 512, 768))
# Round 6: an ODD shape of the reference's eval set -- 577x768: every level's height is odd (577, 289, 145, 73, 37), so lazyDWT pads the
# bottom row at every level (LLICTI_nets.py:226-240), bands x11 / x10 code one row less than the band grid (:396-397) and the decoder re-pads
# (:511-530).  Its samples go to their own file (fullsize_samples_ragged.npz; `python make_fixture_fullsize_samples.py ragged`): positions on the FULL
# band grid -- the padded last row among them -- with the parameters everywhere and symbols / table entries where the position is coded.
IMAGES_RAGGED = (("smooth13_trainedlike_577x768", "smooth", 13, "trainedlike", 577, 768),)


def main():
    ragged = len(sys.argv) > 1 and sys.argv[1] == "ragged"
    rec = Recorder()
    mf._install_standins(rec)
    sys.path.insert(0, mf.REF)
    import graphs.models.LLICTI_nets as ref_nets  # noqa: E402  (reference-owned code)
    cfg = mf.Cfg(json.load(open(os.path.join(mf.REF, "configs", "llicti_A.json"))))
    torch.use_deterministic_algorithms(True)
    torch.set_num_threads(8)
    params_log = []
    orig = ref_nets.LLICTIEntropyModel4.get_params

    def get_params_rec(self, y_condition):
        p = orig(self, y_condition)
        params_log.append(p.detach().clone().numpy())
        return p
    ref_nets.LLICTIEntropyModel4.get_params = get_params_rec
    out = {}
    meta = {}
    for name, kind, seed, wname, H, W in (IMAGES_RAGGED if ragged else IMAGES):
        torch.manual_seed(1337)
        model = ref_nets.LLICTI(cfg).eval()
        if wname == "trainedlike":
            mf.trained_like_(model)
        rgb = mf.make_image(kind, H, W, seed)
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255.0)).unsqueeze(0)
        rec.pairs.clear()
        params_log.clear()
        with torch.no_grad():
            model.compress(x.clone())
        assert len(rec.pairs) == 45 and len(params_log) == 15
        rng = np.random.default_rng(99)
        for lvl in (1, 0):
            for band in range(3):
                call = (4 - lvl) * 3 + band                      # get_params calls: scale 4..0 x band 0..2
                p = params_log[call][0]                          # [60, h, w]
                h, w = p.shape[1:]
                pos = sample_positions(h, w, rng)
                tag = f"{name}_l{lvl}_b{band}"
                out[tag + "_pos"] = pos.astype(np.int16)
                out[tag + "_params"] = p[:, pos[:, 0], pos[:, 1]].T.astype(np.float32).copy()          # [N, 60]
                for clr in range(3):
                    cdf, sym = rec.pairs[(4 - lvl) * 9 + band * 3 + clr]
                    cdf = cdf[0, 0].view(np.uint16)              # [h', w', Lp]
                    sym = sym[0, 0]
                    hc, wc = cdf.shape[:2]
                    if not ragged:
                        assert (hc, wc) == (h, w), (cdf.shape, h, w)      # 768x512: no odd edge, the coded crop is the band grid
                    assert hc in (h, h - 1) and wc in (w, w - 1)
                    Lp = cdf.shape[2]
                    idx = np.zeros((len(pos), N_ENT), dtype=np.int16)
                    val = np.zeros((len(pos), N_ENT), dtype=np.uint16)
                    inside = (pos[:, 0] < hc) & (pos[:, 1] < wc)           # (an odd edge: the band grid's last row / column is not coded in bands x11 / x10, x11 / x01)
                    symv = np.full(len(pos), -1, dtype=np.int16)
                    for k, (i, j) in enumerate(pos):
                        if not inside[k]:
                            continue
                        symv[k] = sym[i, j]
                        s = int(sym[i, j])
                        want = [s, min(s + 1, Lp - 1), max(s - 1, 0), 0, 1, Lp - 2, Lp - 1]
                        want += [int(v) for v in np.linspace(2, Lp - 3, N_ENT - len(want))]
                        idx[k] = want[:N_ENT]
                        val[k] = cdf[i, j, idx[k]]
                    out[f"{tag}_c{clr}_sym"] = symv                        # (-1: the position is not coded)
                    meta[f"{tag}_c{clr}_crop"] = [int(hc), int(wc)]
                    out[f"{tag}_c{clr}_idx"] = idx
                    out[f"{tag}_c{clr}_val"] = val
                    meta[f"{tag}_c{clr}_Lp"] = int(Lp)
        meta[name] = {"kind": kind, "seed": seed, "weights": wname, "H": H, "W": W}
        print(name, "done")
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    dst = os.path.join(HERE, "fullsize_samples_ragged.npz" if ragged else "fullsize_samples.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
