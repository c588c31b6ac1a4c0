#!/usr/bin/env python3
"""How long the serial tail of the rANS v3 container gets when symbols are cheap (GPU box).  The tail coder fills the payload
of a stream's lane states (64 x 31 bits) with the stream's last T symbols, so T grows as bits per symbol fall: ~155 for the
bench's uniform noise (12.8 bits per symbol), up to the format's 2047 for a near-deterministic source.  Cases: the bench batch;
a smooth batch with the trained-like weights; a flat grey batch with all-zero weights (sigma at its bound, mu exact: the
cheapest symbols the model can code).  Prints decode time, the tail kernel's share and bits per symbol."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from llicti_amd.codec import HipCodec, mode_of_name
from helpers import make_batch
def load_state_dict(w):
    return dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', f'weights_{w}.npz')))

B, H, W = 24, 512, 768
out = []
for name, kind, wname in (("noise / seed-1337-like weights", "noise", "rand1337"), ("smooth / trained-like weights", "smooth", "trainedlike"),
                          ("flat grey 127 / all-zero weights", "flat", "zero")):
    sd = load_state_dict("rand1337" if wname == "zero" else wname)
    if wname == "zero":
        sd = {k: np.zeros_like(v) for k, v in sd.items()}
    rgb = np.full((B, 3, H, W), 127, np.uint8) if kind == "flat" else make_batch(kind, B, H, W, seed0=5)
    x = torch.from_numpy(rgb).cuda()
    for cname in sys.argv[1:] or ["rans10", "wrans10"]:
        mode = mode_of_name(cname)
        c = HipCodec("cuda:0"); c.load_state_dict(sd)
        cont, seg = c.encode(x, mode=mode); c.check()
        nbytes = int(seg.sum().item())
        c.set_profiling(True)
        for _ in range(2):
            rec = c.decode(cont, seg, H, W, mode=mode)
        c.check()
        assert torch.equal(rec, x)
        cat, _ = c.last_timing_detail()
        ms, _ = c.last_timing()
        c.set_profiling(False)
        r = {"case": name, "container": cname, "bits_per_symbol": round(8.0 * nbytes / (B * 3 * H * W), 3),
             "decode_ms": round(ms[0], 3), "kernel_ms": {k: round(v, 3) for k, v in cat.items() if v > 0}}
        print(json.dumps(r), flush=True)
        out.append(r)
        c.close()
if os.environ.get("PROBE_OUT"):
    json.dump({"tool": "tools/probe_tail.py", "runs": out}, open(os.environ["PROBE_OUT"], "w"), indent=1)
