"""Synthetic test images of the bench and the parity tests: i.i.d. uniform RGB
("noise"), or low-pass noise + gradient with natural-range chroma ("smooth").  NumPy default_rng(seed): the same bytes everywhere, so
a fixture generated from the reference's Python in the build container (tests/golden/make_fixture_ideal_bits.py) names an image
by (kind, H, W, seed) only."""
from __future__ import annotations

import numpy as np


def make_image(kind: str, H: int, W: int, seed: int) -> np.ndarray:
    """uint8 [3, H, W]."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
    if kind != "smooth":
        raise ValueError(f"unknown image kind {kind!r}")
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
