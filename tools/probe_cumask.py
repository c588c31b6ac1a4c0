#!/usr/bin/env python3
"""A/B of the CU-partitioned decode: decode of 24 x 768x512 as S sub-batches whose CNN launches run on HIP streams
masked to n_cu - R compute units (hipExtStreamCreateWithCUMask) and whose rANS stages run on streams masked to the other R, against the
plain decode and the unmasked sub-batch pipeline.  Prints / writes JSON: ms per decode for M in --ms."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llicti_amd.codec import HipCodec, MODE_RANS
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="2,8")
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--out", default="")
a = ap.parse_args()
B, H, W = a.batch, 512, 768
torch.manual_seed(1337)
sd = LLICTI(default_config()).state_dict()
rgb = torch.from_numpy(np.stack([np.random.default_rng(i).integers(0, 256, size=(3, H, W), dtype=np.uint8) for i in range(B)])).cuda()
configs = [("plain", {})] + [(f"S{S}", {"LLICTI_PIPELINE": str(S)}) for S in (2,)] + \
          [(f"S{S}_R{R}", {"LLICTI_PIPELINE": str(S), "LLICTI_CUMASK": str(R)}) for S in (2,) for R in (48, 64, 80)]
res = []
for name, env in configs:
    for k in ("LLICTI_PIPELINE", "LLICTI_CUMASK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = HipCodec("cuda:0")
    c.load_state_dict(sd)
    row = {"config": name}
    for M in [int(v) for v in a.ms.split(",")]:
        mode = MODE_RANS(M)
        cont, seg = c.encode(rgb, mode=mode)
        c.check()
        c.poison_workspace()
        rec = c.decode(cont, seg, H, W, mode=mode)
        c.check()
        assert torch.equal(rec, rgb), name
        for _ in range(2):
            c.decode(cont, seg, H, W, mode=mode, out=rec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            c.decode(cont, seg, H, W, mode=mode, out=rec)
        torch.cuda.synchronize()
        row[f"dec_ms_M{M}"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        assert torch.equal(rec, rgb), name
    c.close()
    res.append(row)
    print(row, flush=True)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
