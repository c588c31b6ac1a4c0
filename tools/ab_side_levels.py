#!/usr/bin/env python3
"""A/B of llicti_set_tuning("enc_side_levels"): encoder levels 4..1 on a side stream next to level 0 (0 = one queue, 1 = side stream).
python tools/ab_side_levels.py > gpurun_out/ab_side_levels.json"""
import json, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from llicti_amd.codec import HipCodec, mode_of_name
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

dev = torch.device("cuda", 0)
torch.manual_seed(1337)
codec = HipCodec(dev)
codec.load_state_dict(LLICTI(default_config()).state_dict())
out = {}
for B, H, W, name in ((1, 512, 768, "rans128"), (1, 512, 768, "xrans64"), (2, 512, 768, "rans128"), (4, 512, 768, "rans32"), (24, 512, 768, "xrans9"), (1, 2160, 3840, "xrans64")):
    mode = mode_of_name(name)
    rgb = torch.from_numpy(bench.make_batch(B, H, W, 0)).to(dev)
    codec.set_tuning("enc_side_levels", 0)
    cont, seg = codec.encode(rgb, mode=mode)
    codec.check()
    ref, refseg = cont.clone(), seg.clone()
    res = {}
    for rep in range(3):
        for v in (0, 1):
            codec.set_tuning("enc_side_levels", v)
            cont.zero_()
            codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
            torch.cuda.synchronize()
            used = torch.arange(cont.shape[1], device=dev)[None, :] < refseg.sum(dim=1, keepdim=True)
            assert torch.equal(seg, refseg) and torch.equal(cont * used, ref * used), (B, name, v)
            t0 = time.perf_counter()
            for _ in range(20):
                codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
            torch.cuda.synchronize()
            res.setdefault(str(v), []).append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
    codec.check()
    out[f"{B}x{W}x{H}_{name}"] = {k: statistics.median(v) for k, v in res.items()}
codec.set_tuning("enc_side_levels", 0)
print(json.dumps(out, indent=1))
