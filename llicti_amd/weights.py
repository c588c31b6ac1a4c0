"""Canonical packing of the interpolator-CNN weights (host logic; numpy only).

Maps the reference's `state_dict` (33 entries, key names as in /root/reference
graphs/models/LLICTI_nets.py:651-675, :697, :711) to the per-band arrays the
numerics spec is written in:

    w0 [352][K0]  concatenation along K of the band's layer-0 conv weights, sources in lazyDWT order
                  x00, x11, x01 (K0 = 48/72/120).  A conv with a 4-wide kernel (4x4, 3x4) is flattened
                  (ci, ky, kx); a conv with a 4-high, 3-wide kernel (4x3) is flattened (ci, kx, ky).  Either
                  way four consecutive k walk the kernel's length-4 axis, which is what one fp32 MFMA
                  k-step (k = 4) consumes.  The K order is the order of the fmaf chain, i.e. part of the spec.
    b0 [352]      fp32 sum of those convs' biases, left to right
    w1 [352][88], b1 [352]    grouped 1x1 (4 heads of 88)
    w2 [60][88],  b2 [60]     grouped 1x1 (4 heads of 15 outputs)

compressai's extra buffers in a real checkpoint (`*_bound.bound`, `scale_table`, `_offset`,
`_quantized_cdf`, `_cdf_length` ...) are ignored: the bounds 0.11/255 and 1e-6 are constants of the
spec (entropy_layer_nets.py:149-158).
"""
from __future__ import annotations

import numpy as np

PREFIX = "entropymodel.entmdls_scale_band.0."
# band -> layer-0 conv names in source order x00, x11, x01 (LLICTI_nets.py:651-675)
LAYER0 = {
    0: ["layer0_00_11"],
    1: ["layer0_00_01", "layer0_11_01"],
    2: ["layer0_00_10", "layer0_11_10", "layer0_01_10"],
}
K0 = {0: 48, 1: 72, 2: 120}
NCH, HEAD, NPAR = 352, 88, 60


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def pack_state_dict(sd) -> dict:
    """state_dict -> {band: {"K0", "w0", "b0", "w1", "b1", "w2", "b2"}} (all float32, C-contiguous)."""
    out = {}
    for b in range(3):
        p = f"{PREFIX}{b}."
        ws, bs = [], []
        for name in LAYER0[b]:
            w = _np(sd[p + name + ".weight"])
            assert w.shape[0] == NCH and w.shape[1] == 3, w.shape
            if w.shape[3] == 3:                      # (kh, kw) = (4, 3): walk ky fastest
                assert w.shape[2] == 4
                w = np.ascontiguousarray(w.transpose(0, 1, 3, 2))
            else:
                assert w.shape[3] == 4
            ws.append(w.reshape(NCH, -1))
            bs.append(_np(sd[p + name + ".bias"]))
        w0 = np.ascontiguousarray(np.concatenate(ws, axis=1))
        assert w0.shape == (NCH, K0[b]), w0.shape
        b0 = bs[0].copy()
        for extra in bs[1:]:
            b0 = (b0 + extra).astype(np.float32)
        w1 = _np(sd[p + "layers1toL.0.weight"]).reshape(NCH, HEAD)
        b1 = _np(sd[p + "layers1toL.0.bias"])
        w2 = _np(sd[p + "layers1toL.2.weight"]).reshape(NPAR, HEAD)
        b2 = _np(sd[p + "layers1toL.2.bias"])
        out[b] = {"K0": K0[b], "w0": w0, "b0": np.ascontiguousarray(b0), "w1": np.ascontiguousarray(w1),
                  "b1": b1, "w2": np.ascontiguousarray(w2), "b2": b2}
    return out


def expected_keys():
    keys = []
    for b in range(3):
        p = f"{PREFIX}{b}."
        for name in LAYER0[b]:
            keys += [p + name + ".weight", p + name + ".bias"]
        keys += [p + "layers1toL.0.weight", p + "layers1toL.0.bias",
                 p + "layers1toL.2.weight", p + "layers1toL.2.bias"]
    return keys


# Buffers compressai's GaussianConditional / EntropyModel register on `conditional_prob_model` in a real checkpoint
# (compressai==1.1.8: entropy_models.py) that this build's parameter containers do not carry: they are tables of the
# quantised-CDF coder the reference never uses on this path (it calls torchac with its own get_cdfs tables).
COMPRESSAI_EXTRA_SUFFIXES = ("._offset", "._quantized_cdf", "._cdf_length", ".scale_table", ".scale_bound")


def load_reference_state_dict(model, sd):
    """Strict load of a reference `state_dict` (agents/base.py:60 loads strictly too): only the known compressai
    buffers above are dropped; a missing key, an unexpected key (e.g. a `module.` prefix) or a shape mismatch raises
    instead of silently leaving the seeded default init in place."""
    kept = {k: v for k, v in sd.items() if not k.endswith(COMPRESSAI_EXTRA_SUFFIXES)}
    missing = [k for k in expected_keys() if k not in kept]
    if missing:
        raise KeyError(f"checkpoint state_dict lacks {len(missing)} of the 24 weight entries of LLICTI (config A), e.g. {missing[:3]}; "
                       f"first keys present: {list(sd)[:3]}")
    model.load_state_dict(kept, strict=True)
    return model
