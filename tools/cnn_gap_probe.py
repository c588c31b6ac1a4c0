#!/usr/bin/env python3
"""why are the encoder's band-CNN launches ~7 % slower than the decoder's (same kernels, tiles, FLOPs)?  Timing experiments on
the level-0 launches of bench.py's batch (24 x 768x512), HIP events on the launch stream, each variant N launches:
  back_to_back      the three bands' launches in a row, nothing between them (the matrix pipe never rests)
  pairs_between     ... with the encoder's cdf_pairs launch behind each (the encoder's schedule)
  idle_between      ... with ~0.6 ms of an idle queue between them (what the decoder's ~0.6 ms rANS stage launch is to the matrix pipe)
  light_between     ... with a bandwidth-light, VALU-only kernel of the same length between them
If `idle_between` / `light_between` run the CNN launches faster than `back_to_back` / `pairs_between`, the difference is the clock the chip holds
under sustained fp32-MFMA load (power management), not something the encoder's code does.
Usage: python tools/cnn_gap_probe.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from llicti_amd.codec import HipCodec  # noqa: E402
from llicti_amd.config import default_config  # noqa: E402
from llicti_amd.graphs.models.LLICTI_nets import LLICTI  # noqa: E402

B, H, W = 24, 512, 768
dev = torch.device("cuda", 0)
torch.manual_seed(1337)
codec = HipCodec(dev)
codec.load_state_dict(LLICTI(default_config()).state_dict())
rgb = torch.from_numpy(np.stack([np.random.default_rng(i).integers(0, 256, (3, H, W), dtype=np.uint8) for i in range(B)])).to(dev)
planes, fplanes, mm = codec.lift(rgb)
L, Cc = codec.L, __import__("ctypes")
from llicti_amd import _lib  # noqa: E402
from llicti_amd.codec import _ptr, _stream_ptr  # noqa: E402
h, w = H // 2, W // 2
params = torch.empty((B, 64, h, w), dtype=torch.float32, device=dev)
pairs = torch.empty((3, B, h * w), dtype=torch.int32, device=dev)
filler = torch.empty(1 << 20, device=dev)


def cnn(band):
    _lib.check(L.llicti_band_params_f32(codec.ctx, _ptr(fplanes), B, H, W, 0, band, _ptr(params), _stream_ptr(dev)))


def pairs_k(band):
    _lib.check(L.llicti_cdf_pairs_u32(codec.ctx, _ptr(planes), _ptr(params), _ptr(mm), B, H, W, 0, band, _ptr(pairs), _stream_ptr(dev)))


def idle():
    torch.cuda._sleep(int(0.6e-3 * 2.4e9))          # ~0.6 ms of a spinning single wavefront: the chip is idle


def light():
    x = filler
    for _ in range(40):
        x = x * 1.0001 + 0.5                         # ~15 us each on 4 MB: VALU + a little L2 traffic


def run(between, reps=8):
    ms = {0: [], 1: [], 2: []}
    for _ in range(2):
        for band in range(3):
            cnn(band)
    torch.cuda.synchronize()
    for _ in range(reps):
        for band in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cnn(band)
            e1.record()
            if between is not None:
                between(band)
            ms[band].append((e0, e1))
    torch.cuda.synchronize()
    return {f"band{b}_ms": round(float(np.median([a.elapsed_time(c) for a, c in v])), 4) for b, v in ms.items()}


out = {"workload": f"{B}x{W}x{H}, level 0, three band-CNN launches per round, median of 8 rounds, HIP events around each launch"}
for rep in range(2):        # twice, interleaved: the order of the variants must not matter
    for name, fn in (("back_to_back", None), ("pairs_between", lambda b: pairs_k(b)), ("idle_between", lambda b: idle()), ("light_between", lambda b: light())):
        r = run(fn)
        r["sum_ms"] = round(sum(r.values()), 4)
        out.setdefault(name, []).append(r)
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
