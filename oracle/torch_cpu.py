"""Plain-PyTorch (CPU) restatement of the reference's compress / decompress path -- the "PyTorch-CPU baseline" BASELINE.json's
north_star asks to have measured beside the MI355X number.  TEST INFRASTRUCTURE ONLY, like everything under oracle/: only tests/
and bench.py's cpu_baseline leg may import it; the product never does.

It follows the reference's structure, not the build's: torch ops for everything the reference does in torch --
  * the interpolator CNN as nn.functional.conv2d calls with replicate pads   (LLICTI_nets.py:651-675, :695-712, :721-753)
  * the mixture CDF as a materialised [positions, 5, Lp] erfc tensor, summed  (entropy_layer_nets.py:185-204, LLICTI_nets.py:938-952)
  * integerisation round(cdf * (65536 - (Lp - 1))) -> int16 wrap -> + arange  (LLICTI_nets.py:955-983)
-- and a single-threaded range coder on the materialised int16 tables in place of torchac (absent from this image): the C
oracle's coder, which implements the same algorithm (oracle/llicti_oracle.c, Appendix A of SURVEY.md).  The 45 stages run in
the reference's order; the decoder re-runs the CNN on what it has decoded so far, as the reference's does (:415-509).
torch's erfc / conv summation order are not the numerics spec's, so its tables differ from the oracle's in an entry here and
there (hence its bytes do): it is a TIMING baseline whose own round trip is lossless and whose size agrees with the oracle's to
a few bytes (tests/test_oracle_golden.py::test_torch_cpu_path_roundtrip)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc

PREFIX = "entropymodel.entmdls_scale_band.0."
# band -> (conv name, source component (a, b), replicate pad (l, r, t, b)); kernel shapes come with the weights
LAYER0 = {
    0: [("layer0_00_11", (0, 0), (1, 2, 1, 2))],
    1: [("layer0_00_01", (0, 0), (1, 2, 1, 1)), ("layer0_11_01", (1, 1), (1, 1, 2, 1))],
    2: [("layer0_00_10", (0, 0), (1, 1, 1, 2)), ("layer0_11_10", (1, 1), (2, 1, 1, 1)), ("layer0_01_10", (0, 1), (2, 1, 1, 2))],
}
TARGET = {0: (1, 1), 1: (0, 1), 2: (1, 0)}          # band -> polyphase component it codes (x11, x01, x10)
SCALE_BOUND, WEIGHT_BOUND = 0.11 / 255.0, 1e-6


def _t(sd, key):
    v = sd[key]
    return v.detach().cpu().float() if hasattr(v, "detach") else torch.from_numpy(np.asarray(v, dtype=np.float32))


def _component(fp, lvl, a, b, h, w):
    """Polyphase component (a, b) of level lvl of the float planes [3, H, W], replicate-padded to the level's band grid h x w."""
    c = fp[:, (a << lvl)::(2 << lvl), (b << lvl)::(2 << lvl)]
    ph, pw = h - c.shape[1], w - c.shape[2]
    if ph or pw:
        c = F.pad(c.unsqueeze(0), (0, pw, 0, ph), mode="replicate")[0]
    return c


def band_params(fp, lvl, band, sd, geom):
    """-> [60, h, w]: sigma | mu | weight (Y, Co, Cg x 5 each) | a, b, d x 5."""
    _, _, h, w, _, _ = geom
    acc = None
    for name, (a, b), pad in LAYER0[band]:
        x = F.pad(_component(fp, lvl, a, b, h, w).unsqueeze(0), pad, mode="replicate")
        y = F.conv2d(x, _t(sd, f"{PREFIX}{band}.{name}.weight"), _t(sd, f"{PREFIX}{band}.{name}.bias"))
        acc = y if acc is None else acc + y
    y = F.relu(acc)
    y = F.relu(F.conv2d(y, _t(sd, f"{PREFIX}{band}.layers1toL.0.weight"), _t(sd, f"{PREFIX}{band}.layers1toL.0.bias"), groups=4))
    return F.conv2d(y, _t(sd, f"{PREFIX}{band}.layers1toL.2.weight"), _t(sd, f"{PREFIX}{band}.layers1toL.2.bias"), groups=4)[0]


def _tables(par, clr, yv, cov, minv, maxv, chunk=8192):
    """par [60, N], prior-channel target pixels yv / cov [N] -> int16 tables [N, Lp] (uint16 bit pattern), the reference's way."""
    Lp = maxv - minv + 2
    grid = torch.linspace(minv - 0.5, maxv + 0.5, Lp, dtype=torch.float64) / 255.0
    grid[0] -= 20.0 / 255.0
    grid[-1] += 20.0 / 255.0
    grid = grid.float()
    out = torch.empty((par.shape[1], Lp), dtype=torch.int16)
    ar = torch.arange(Lp, dtype=torch.int32)
    for s in range(0, par.shape[1], chunk):
        p = par[:, s:s + chunk]
        sg = torch.clamp(p[5 * clr:5 * clr + 5], min=SCALE_BOUND)
        mu = p[15 + 5 * clr:15 + 5 * clr + 5]
        if clr == 1:
            mu = mu + p[45:50] * yv[s:s + chunk]
        elif clr == 2:
            mu = mu + (p[50:55] * yv[s:s + chunk] + p[55:60] * cov[s:s + chunk])
        wk = torch.clamp(p[30 + 5 * clr:30 + 5 * clr + 5], min=WEIGHT_BOUND)
        wk = wk / (1e-9 + wk.sum(0, keepdim=True))
        z = (grid[None, None, :] - mu[:, :, None]) / sg[:, :, None]                      # [5, n, Lp]
        cdf = (wk[:, :, None] * (0.5 * torch.erfc(-(2 ** -0.5) * z))).sum(0)            # [n, Lp]
        q = torch.round(cdf * float(65536 - (Lp - 1))).to(torch.int32)
        out[s:s + chunk] = ((q + ar[None, :]) & 0xFFFF).to(torch.int16)                 # int16 wrap, then + arange
    return out


def _stage_positions(H, W, lvl, band):
    """Full-resolution (rows, cols) of the band's coded (cropped) positions."""
    a, b = TARGET[band]
    rows = np.arange((a << lvl), H, (2 << lvl))
    cols = np.arange((b << lvl), W, (2 << lvl))
    return rows, cols


def _ranges(mm6):
    """(min, max, shift) of the Y, Co, Cg alphabets from the header's six int16 (0, minCo, minCg, 255, maxCo, maxCg)."""
    return [(-127, 128, 127), (int(mm6[1]), int(mm6[4]), -int(mm6[1])), (int(mm6[2]), int(mm6[5]), -int(mm6[2]))]


def encode(rgb, sd):
    """uint8 [3, H, W] -> (streams: 45 bytes objects in coding order, meta for decode)."""
    planes, mm = orc.lift(rgb)                                   # integer lift (torch int ops in the reference; negligible either way)
    H, W = planes.shape[1:]
    fp = torch.from_numpy(planes.astype(np.float32) / np.float32(255.0))
    streams = []
    with torch.no_grad():
        for lvl in range(4, -1, -1):
            geom = orc.level_geom(H, W, lvl)
            for band in range(3):
                par = band_params(fp, lvl, band, sd, geom)
                rows, cols = _stage_positions(H, W, lvl, band)
                par = par[:, :len(rows), :len(cols)].reshape(60, -1)
                tgt = planes[:, rows][:, :, cols].reshape(3, -1)
                yv = torch.from_numpy(tgt[0].astype(np.float32) / np.float32(255.0))
                cov = torch.from_numpy(tgt[1].astype(np.float32) / np.float32(255.0))
                for clr, (minv, maxv, shift) in enumerate(_ranges(mm)):
                    tab = _tables(par, clr, yv, cov, minv, maxv)
                    streams.append(orc.ac_encode_tables(tab.numpy(), (tgt[clr] + shift).astype(np.int16)))
    dc = planes[:, ::32, ::32].copy()
    return streams, {"H": H, "W": W, "mm": mm, "dc": dc}


def decode(streams, meta, sd):
    """-> uint8 [3, H, W]"""
    H, W, mm = meta["H"], meta["W"], meta["mm"]
    planes = np.zeros((3, H, W), np.int16)
    planes[:, ::32, ::32] = meta["dc"]
    it = iter(streams)
    with torch.no_grad():
        for lvl in range(4, -1, -1):
            geom = orc.level_geom(H, W, lvl)
            for band in range(3):
                fp = torch.from_numpy(planes.astype(np.float32) / np.float32(255.0))
                par = band_params(fp, lvl, band, sd, geom)
                rows, cols = _stage_positions(H, W, lvl, band)
                par = par[:, :len(rows), :len(cols)].reshape(60, -1)
                n = len(rows) * len(cols)
                dec = np.zeros((3, n), np.int16)
                for clr, (minv, maxv, shift) in enumerate(_ranges(mm)):
                    yv = torch.from_numpy(dec[0].astype(np.float32) / np.float32(255.0))
                    cov = torch.from_numpy(dec[1].astype(np.float32) / np.float32(255.0))
                    tab = _tables(par, clr, yv, cov, minv, maxv)
                    dec[clr] = orc.ac_decode_tables(tab.numpy(), next(it), n) - shift
                planes[:, rows[:, None], cols[None, :]] = dec.reshape(3, len(rows), len(cols))
    return orc.unlift(planes)
