#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own Python.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  The reference-owned hot-path code (graphs/models/LLICTI_nets.py,
graphs/layers/entropy_layer_nets.py) is imported unmodified from /root/reference.  Its two
third-party dependencies are absent from this image (compressai==1.1.8, torchac==0.9.3,
reference README.md:11-12), so in-memory stand-ins for the handful of names the hot path
touches are registered in sys.modules first:

  compressai.ops.LowerBound                      -> torch.max(x, bound)
  compressai.entropy_models.GaussianConditional  -> ctor storing lower_bound_scale /
                                                    likelihood_lower_bound, and the static
                                                    _standardized_cumulative = 0.5*erfc(-x/sqrt 2)
  compressai.entropy_models.EntropyBottleneck    -> empty nn.Module (unused by config A)
  compressai.layers.GDN1                         -> nn.Identity placeholder (unused by config A)
  torchac.{encode,decode}_int16_normalized_cdf   -> a RECORDER: captures the (cdf, sym) pair the
                                                    reference hands to the coder and round-trips
                                                    the symbols through a trivial byte packing.

So everything pinned here is "reference-owned code + a restatement of three compressai
one-liners"; the arithmetic coder itself is NOT pinned by this script (torchac is absent:
"parity unpinned" for the coder).

Only data (inputs / outputs) is written; no reference source text is stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stand-ins
def _install_standins(recorder):
    class LowerBound(nn.Module):
        def __init__(self, bound):
            super().__init__()
            self.register_buffer("bound", torch.Tensor([float(bound)]))

        def forward(self, x):
            return torch.max(x, self.bound)

    class EntropyModel(nn.Module):
        def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
            super().__init__()
            self.use_likelihood_bound = likelihood_bound > 0
            if self.use_likelihood_bound:
                self.likelihood_lower_bound = LowerBound(likelihood_bound)

    class EntropyBottleneck(EntropyModel):
        def __init__(self, channels=1, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
            super().__init__()

        def loss(self):
            return 0.0

    class GaussianConditional(EntropyModel):
        def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
            super().__init__(*args, **kwargs)
            self.lower_bound_scale = LowerBound(scale_bound)

        @staticmethod
        def _standardized_cumulative(inputs):
            half = float(0.5)
            const = float(-(2 ** -0.5))
            return half * torch.erfc(const * inputs)

    class GDN1(nn.Identity):
        def __init__(self, in_channels=None, **kw):
            super().__init__()

    compressai = types.ModuleType("compressai")
    ops = types.ModuleType("compressai.ops")
    ops.LowerBound = LowerBound
    em = types.ModuleType("compressai.entropy_models")
    em.EntropyBottleneck = EntropyBottleneck
    em.GaussianConditional = GaussianConditional
    layers = types.ModuleType("compressai.layers")
    layers.GDN1 = GDN1
    compressai.ops, compressai.entropy_models, compressai.layers = ops, em, layers
    sys.modules.update({"compressai": compressai, "compressai.ops": ops,
                        "compressai.entropy_models": em, "compressai.layers": layers})

    torchac = types.ModuleType("torchac")

    def encode_int16_normalized_cdf(cdf_int, sym):
        assert cdf_int.dtype == torch.int16 and sym.dtype == torch.int16
        assert not cdf_int.is_cuda and not sym.is_cuda
        assert cdf_int.shape[:-1] == sym.shape
        recorder.append((cdf_int.clone().numpy(), sym.clone().numpy()))
        return sym.contiguous().numpy().tobytes()

    def decode_int16_normalized_cdf(cdf_int, byte_stream):
        shp = cdf_int.shape[:-1]
        return torch.from_numpy(np.frombuffer(byte_stream, dtype=np.int16).copy()).view(*shp)

    torchac.encode_int16_normalized_cdf = encode_int16_normalized_cdf
    torchac.decode_int16_normalized_cdf = decode_int16_normalized_cdf
    sys.modules["torchac"] = torchac


class Cfg(dict):
    __getattr__ = dict.__getitem__


def trained_like_(model, seed=7):
    """Make the random-init net behave like a trained one: sigma of a few grey levels, positive
    mixture weights, small cross-channel coefficients.  Only the last 1x1 layer is touched."""
    g = torch.Generator().manual_seed(seed)
    for bm in model.entropymodel.entmdls_scale_band[0]:
        last = bm.layers1toL[2]
        with torch.no_grad():
            last.weight.mul_(0.25)
            b = last.bias
            b[0:15] = (1.5 + 10.0 * torch.rand(15, generator=g)) / 255.0      # sigma
            b[15:30] = (torch.rand(15, generator=g) - 0.5) * 6.0 / 255.0       # mu offsets
            b[30:45] = 0.2 + torch.rand(15, generator=g)                      # weights
            b[45:60] = (torch.rand(15, generator=g) - 0.3) * 0.8              # a, b, d


def make_image(kind, H, W, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
    # smooth: low-pass noise + gradient, natural-range chroma
    base = rng.standard_normal((3, H + 16, W + 16))
    k = np.ones(9) / 9.0
    for _ in range(2):
        base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, base)
        base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 2, base)
    base = base[:, 8:8 + H, 8:8 + W]
    lum = base[0:1] * 220.0
    img = 128 + lum + base * 60.0 + np.linspace(-40, 40, W)[None, None, :]
    img = img + rng.standard_normal(img.shape) * 2.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    recorder = []
    _install_standins(recorder)
    sys.path.insert(0, REF)
    from graphs.models.LLICTI_nets import LLICTI  # noqa: E402  (reference-owned code)

    cfg = Cfg(json.load(open(os.path.join(REF, "configs", "llicti_A.json"))))
    torch.use_deterministic_algorithms(True)
    torch.set_num_threads(4)

    weights = {}
    for wname in ("rand1337", "trainedlike"):
        torch.manual_seed(1337)
        model = LLICTI(cfg).eval()
        if wname == "trainedlike":
            trained_like_(model)
        sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        weights[wname] = (model, sd)
        np.savez_compressed(os.path.join(OUT, f"weights_{wname}.npz"), **sd)
        print(wname, "params", sum(p.numel() for p in model.parameters()), "state_dict keys", len(sd))

    cases = [
        # name, kind, H, W, seed, weights, store_all_params
        ("noise_32x32_rand", "noise", 32, 32, 0, "rand1337", True),
        ("noise_67x93_rand", "noise", 67, 93, 1, "rand1337", False),
        ("smooth_64x48_tl", "smooth", 64, 48, 2, "trainedlike", True),
        ("smooth_67x93_tl", "smooth", 67, 93, 3, "trainedlike", False),
        ("noise_33x64_tl", "noise", 33, 64, 4, "trainedlike", False),
    ]
    index = {}
    for name, kind, H, W, seed, wname, store_all in cases:
        model, _ = weights[wname]
        rgb = make_image(kind, H, W, seed)
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255.0)).unsqueeze(0)
        out = {"rgb": rgb}
        with torch.no_grad():
            # integer lift (reference LLICTI_nets.py:62-74)
            ycc = LLICTI.get_YCoCg_R_from_RGB__intOps(x.clone())
            out["ycocg_int16"] = ycc.numpy()[0]
            # hook get_params to capture the 60-channel mixture parameters per (scale, band) call
            params_log = []
            bms = list(model.entropymodel.entmdls_scale_band[0])
            origs = [bm.get_params for bm in bms]

            def mk(orig):
                def f(y):
                    p = orig(y)
                    params_log.append(p.detach().clone().numpy()[0])
                    return p
                return f
            for bm, o in zip(bms, origs):
                bm.get_params = mk(o)
            del recorder[:]
            bl, x_ycocg = model.compress(x.clone())
            enc_params = list(params_log)
            enc_rec = list(recorder)
            del params_log[:]
            x_reco = model.decompres(bl, torch.device("cpu"))
            dec_params = list(params_log)
            for bm, o in zip(bms, origs):
                bm.get_params = o
        maxerr = float(((x - x_reco) * 255).abs().max())
        assert maxerr == 0.0, maxerr
        # decoder's params must be bit-identical to the encoder's in the reference (same ops, same shapes)
        for a, b in zip(enc_params, dec_params):
            assert np.array_equal(a, b)
        out["hdr0"] = np.frombuffer(bl[0][0], dtype=np.uint8)
        out["hdr_minmax"] = np.frombuffer(bl[0][1], dtype=np.int16)
        out["hdr_pad"] = np.frombuffer(bl[0][2], dtype=np.int16)
        out["hdr_dc"] = np.frombuffer(bl[0][3], dtype=np.uint8)
        out["x_ycocg_f32"] = x_ycocg.numpy()[0]          # (int16 - [127,0,0]) / 255 as float32
        out["reco_rgb"] = np.rint(x_reco.numpy()[0] * 255).astype(np.uint8)
        # per stage (coarse->fine, band, clr) what the reference handed to the coder
        assert len(enc_rec) == 45 and len(enc_params) == 15
        k = 0
        for si, scl in enumerate(range(4, -1, -1)):
            for b in range(3):
                p = enc_params[si * 3 + b]            # 60 x h x w (means of Co/Cg already updated in place)
                if store_all or scl >= 2:
                    out[f"params_s{scl}_b{b}"] = p
                for clr in range(3):
                    cdf, sym = enc_rec[k]
                    k += 1
                    cdf = cdf[0, 0].view(np.uint16)      # h' x w' x Lp
                    sym = sym[0, 0]
                    out[f"sym_s{scl}_b{b}_c{clr}"] = sym
                    hh, ww, Lp = cdf.shape
                    # keep a strided subset of table rows (full tables would be ~1 kB per symbol)
                    n = hh * ww
                    step = max(1, n // 24)
                    idx = np.arange(0, n, step)[:24]
                    out[f"cdfidx_s{scl}_b{b}_c{clr}"] = idx.astype(np.int32)
                    out[f"cdfrows_s{scl}_b{b}_c{clr}"] = cdf.reshape(n, Lp)[idx]
                    # and the two entries the encoder actually uses, for every symbol
                    flat = cdf.reshape(n, Lp)
                    s = sym.reshape(n).astype(np.int64)
                    out[f"clow_s{scl}_b{b}_c{clr}"] = flat[np.arange(n), s]
                    hi = flat[np.arange(n), s + 1].astype(np.uint32)
                    hi[s == Lp - 2] = 0x10000
                    out[f"chigh_s{scl}_b{b}_c{clr}"] = hi
        np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), **out)
        index[name] = {"kind": kind, "H": H, "W": W, "seed": seed, "weights": wname,
                       "minmax": [int(v) for v in out["hdr_minmax"]], "pad": int(out["hdr_pad"][0])}
        print(name, index[name], "maxerr", maxerr)
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
