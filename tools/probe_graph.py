#!/usr/bin/env python3
"""Would a HIP graph over the decode chain (45 rANS stages + 15 CNN launches) pay?  Captures HipCodec.decode / .encode into a
torch.cuda.CUDAGraph (the library launches on the capturing stream) and compares replay with plain launches."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llicti_amd.codec import HipCodec, MODE_RANS
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI
torch.manual_seed(1337)
codec = HipCodec(torch.device("cuda", 0))
codec.load_state_dict(LLICTI(default_config()).state_dict())
H, W = 512, 768
res = []
for B, M in ((1, 32), (1, 16), (4, 16), (24, 8)):
    mode = MODE_RANS(M)
    g = torch.Generator(device="cuda").manual_seed(B)
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
    cont, seg = codec.encode(rgb, mode=mode)
    rec = codec.decode(cont, seg, H, W, mode=mode)
    codec.check()
    assert torch.equal(rec, rgb)

    def timed(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    n = 50 if B == 1 else 10
    plain_d = timed(lambda: codec.decode(cont, seg, H, W, mode=mode, out=rec), n)
    plain_e = timed(lambda: codec.encode(rgb, mode=mode, out=cont, seg_len=seg), n)
    row = {"B": B, "M": M, "dec_ms_plain": round(plain_d, 3), "enc_ms_plain": round(plain_e, 3)}
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            codec.decode(cont, seg, H, W, mode=mode, out=rec)      # warm on the side stream
            codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
        torch.cuda.synchronize()
        gd, ge = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gd, stream=s):
            codec.decode(cont, seg, H, W, mode=mode, out=rec)
        with torch.cuda.graph(ge, stream=s):
            codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
        rec.zero_()
        ge.replay(); gd.replay()
        torch.cuda.synchronize()
        assert torch.equal(rec, rgb)
        row["dec_ms_graph"] = round(timed(gd.replay, n), 3)
        row["enc_ms_graph"] = round(timed(ge.replay, n), 3)
    except Exception as e:
        row["graph_error"] = repr(e)[:300]
    res.append(row)
    print(row, flush=True)
json.dump(res, open("gpurun_out/probe_graph.json", "w"), indent=1)
