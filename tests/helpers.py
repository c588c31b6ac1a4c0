"""Shared test helpers (synthetic inputs; no reference code)."""
import numpy as np


def make_image(kind, H, W, seed):
    """Same generators as tests/golden/make_fixtures.py: i.i.d. uniform RGB, or low-pass 'smooth' RGB."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
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


def make_batch(kind, B, H, W, seed0=0):
    return np.stack([make_image(kind, H, W, seed0 + i) for i in range(B)])


def make_sampled_image(H, W, seed):
    """An image DRAWN FROM THE MODEL (trained-like weights): the reference-format decoder run on streams of random bytes (an arithmetic decoder
    fed random bits emits symbols with the model's own probabilities), behind the header -- size, value ranges, coarsest pixels -- of the smooth
    image of the same size.  The content class of a model that predicts its data well: cheap symbols (about 4.8 bits each) over the full value
    range -- what the sigma-floor noise batch and the 21-bpp smooth image are not.  Test infrastructure (uses the CPU oracle); deterministic."""
    import os
    from oracle import oracle as orc
    from llicti_amd.weights import pack_state_dict
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    W_o = orc.Weights(pack_state_dict(dict(np.load(os.path.join(gold, "weights_trainedlike.npz")))))
    bl = orc.encode_image(make_image("smooth", H, W, 11), W_o)
    rng = np.random.default_rng(seed)
    bl = [list(bl[0])] + [[rng.integers(0, 256, len(s), dtype=np.uint8).tobytes() for s in row] for row in bl[1:]]
    return orc.decode_image(bl, W_o)


def xwide_stream_header(stream: bytes):
    """(T field = ceil(T / 32), one-chain flag, bits of the main region below the header field) of an xwide v4 stream (oracle/llicti_oracle.h, "stream"):
    bit region | 992 bytes of states; the region's highest set bit is its end marker, the 9 bits below it the header field."""
    nbytes = len(stream) - 992
    assert nbytes >= 2 and stream[nbytes - 1] != 0
    top = 8 * (nbytes - 1) + stream[nbytes - 1].bit_length() - 1
    v = int.from_bytes(stream[:nbytes], "little")
    f9 = (v >> (top - 9)) & 0x1FF
    return f9 & 0xFF, f9 >> 8, top - 9
