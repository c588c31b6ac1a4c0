#!/usr/bin/env python3
"""Golden fixtures for the rows SURVEY.md section 8(f) lists as "next": the training / validation likelihood
path (`LLICTI.forward`, reference LLICTI_nets.py:101-123, :318-342, :802-811, :827-935 and
entropy_layer_nets.py:117-183) and the reporting helpers (`loggers/rate.py:120-168`,
`graphs/losses/rate_dist.py:125-135`).

Like make_fixtures.py this runs ONLY in the build container: it imports the reference-owned modules unmodified
from /root/reference (with the same in-memory stand-ins for the absent compressai / torchac packages) and
writes data only -- inputs and the reference's outputs."""
import json
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402  (stand-ins, image generators, trained-like weights)

REF = mf.REF


def main():
    mf._install_standins([])
    sys.path.insert(0, REF)
    from graphs.models.LLICTI_nets import LLICTI            # reference-owned
    from loggers.rate import RateLogger                      # reference-owned, no third-party imports
    from graphs.losses.rate_dist import CompressionRLossList

    cfg = mf.Cfg(json.load(open(os.path.join(REF, "configs", "llicti_A.json"))))
    torch.use_deterministic_algorithms(True)
    torch.set_num_threads(4)

    # ---- forward(): self-information maps (pad=False: H, W multiples of 32, as the agent's validate() pads)
    for name, kind, H, W, seed, wname in (("fwd_smooth_64x96_tl", "smooth", 64, 96, 11, "trainedlike"),
                                          ("fwd_noise_32x64_rand", "noise", 32, 64, 12, "rand1337")):
        torch.manual_seed(1337)
        model = LLICTI(cfg).eval()
        if wname == "trainedlike":
            mf.trained_like_(model)
        rgb = mf.make_image(kind, H, W, seed)
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255.0)).unsqueeze(0)
        with torch.no_grad():
            ycc = model.get_YCoCg_R_from_RGB(x.clone())                   # float lift, round-half-even (:40-49)
            infos = model.forward(x.clone())                              # list of 5: 1 x 9 x h x w, scale 0 first
        out = {"rgb": rgb, "ycocg_train_f32": ycc.numpy()[0]}
        for s, t in enumerate(infos):
            out[f"selfinfo_s{s}"] = t.numpy()[0]
        out["total_bits"] = np.array([float(sum(t.double().sum() for t in infos))])
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, [tuple(t.shape) for t in infos], "total bits", float(out["total_bits"][0]),
              "bpp", float(out["total_bits"][0]) / (H * W))

    # ---- reporting: CompressionRLossList + RateLogger table text
    rng = np.random.default_rng(5)
    lens = [[3, 12, 2, 192, 0, 0, 0, 0, 0]] + [[int(v) for v in rng.integers(10, 4000, size=9)] for _ in range(5)]
    bl = [[bytes(n) for n in row] for row in lens]
    numel = 3 * 64 * 96
    rates = CompressionRLossList().forward(numel, bl)
    rates2 = [[r * 0.5 + 0.01 for r in row] for row in rates]

    class Grab(logging.Handler):
        def __init__(self):
            super().__init__()
            self.lines = []

        def emit(self, record):
            self.lines.append(record.getMessage())
    grab = Grab()
    lg = logging.getLogger("Rate Loss")
    lg.addHandler(grab)
    lg.setLevel(logging.INFO)
    rl = RateLogger()
    rl._get_time_now_str = lambda: "12:34:56"
    rl(rates)
    rl(rates2)
    tot, zero = rl.display(typ="te")
    texts = {"te": grab.lines[-1]}
    rl(rates)
    rl.display(lr=0.0001, typ="va")
    texts["va"] = grab.lines[-1]
    json.dump({"stream_lengths": lens, "numel": numel, "rates": rates, "rates2": rates2, "display_sum": float(tot),
               "text": texts}, open(os.path.join(HERE, "rate_table.json"), "w"), indent=1)
    print(texts["te"])


if __name__ == "__main__":
    main()
