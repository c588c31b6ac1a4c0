#!/usr/bin/env python3
"""Where does each container saturate?  Encode / decode MPix/s of the AC (reference-format) container over the
batch size, and of the rANS container over M (streams per image), on one MI355X.  Writes a JSON summary
(default gpurun_out/probe_scaling.json; the copy that is judged lives under profiles/<round>/)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llicti_amd.codec import HipCodec, MODE_AC, MODE_RANS
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/probe_scaling.json")
ap.add_argument("--ac-batches", default="1,24,64,128,256")
ap.add_argument("--rans-m", default="1,2,4,8,16,32")
ap.add_argument("--rans-batches", default="1,24")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()

torch.manual_seed(1337)
codec = HipCodec(torch.device("cuda", 0))
codec.load_state_dict(LLICTI(default_config()).state_dict())
H, W = 512, 768
res = {"shape": [H, W], "ac": [], "rans": []}


def run(B, mode, reps):
    g = torch.Generator(device="cuda").manual_seed(B)
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
    cont, seg = codec.encode(rgb, mode=mode)
    rec = codec.decode(cont, seg, H, W, mode=mode)
    codec.check()
    assert torch.equal(rec, rgb)

    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    te = timed(lambda: codec.encode(rgb, mode=mode, out=cont, seg_len=seg))
    td = timed(lambda: codec.decode(cont, seg, H, W, mode=mode, out=rec))
    mp = B * H * W / 1e6
    r = {"B": B, "enc_ms": round(te * 1e3, 3), "dec_ms": round(td * 1e3, 3), "enc_mpix_s": round(mp / te, 1), "dec_mpix_s": round(mp / td, 1),
         "encdec_mpix_s": round(mp / (te + td), 1), "bytes_per_image": float(seg.sum().item()) / B,
         "workspace_GiB": round(codec._ws.numel() / 2**30, 2)}
    del rgb, cont, seg, rec
    codec._ws = None
    torch.cuda.empty_cache()
    return r


for B in [int(v) for v in args.ac_batches.split(",") if v]:
    r = run(B, MODE_AC, args.reps if B < 200 else 1)
    res["ac"].append(r)
    print("AC", r, flush=True)
ac_bytes = {r["B"]: r["bytes_per_image"] for r in res["ac"]}
for B in [int(v) for v in args.rans_batches.split(",") if v]:
    for M in [int(v) for v in args.rans_m.split(",") if v]:
        r = run(B, MODE_RANS(M), args.reps + 1)
        r["M"] = M
        if B in ac_bytes:
            r["bpp_delta_vs_ac"] = round(8.0 * (r["bytes_per_image"] - ac_bytes[B]) / (H * W), 5)
        res["rans"].append(r)
        print("rANS", r, flush=True)
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
json.dump(res, open(args.out, "w"), indent=1)
