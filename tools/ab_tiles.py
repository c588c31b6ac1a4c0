#!/usr/bin/env python3
"""A/B of the band CNN's tile-height choice: per-level CNN time of one encode with round 3's rule (16 or 4 rows: set_tuning
cnn_tile_rows = -1), with the round-4 model (0: 16 / 8 / 4 rows) and with each form forced.  python tools/ab_tiles.py > gpurun_out/ab_tiles.json"""
import json, os, sys, statistics
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
for B, H, W in ((24, 512, 768), (32, 512, 768), (1, 512, 768), (1, 2160, 3840)):
    mode = mode_of_name(bench.default_container(H, W))
    rgb = torch.from_numpy(bench.make_batch(B, H, W, 0)).to(dev)
    cont, seg = codec.encode(rgb, mode=mode)
    ref = cont.clone()
    res = {}
    for rows in (-1, 0, 16, 8, 4):
        codec.set_tuning("cnn_tile_rows", rows)
        codec.set_profiling(False)
        codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
        torch.cuda.synchronize()
        assert torch.equal(cont, ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(); codec.encode(rgb, mode=mode, out=cont, seg_len=seg); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        codec.set_profiling(True)
        lv = []
        for _ in range(3):
            codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
            torch.cuda.synchronize()
            lv.append(codec.last_cnn_level_ms())
        res[str(rows)] = {"encode_ms": round(statistics.median(ts), 4), "cnn_level_ms": [round(statistics.median(x[l] for x in lv), 4) for l in range(5)]}
    codec.set_tuning("cnn_tile_rows", 0)
    codec.set_profiling(False)
    out[f"{B}x{W}x{H}"] = res
print(json.dumps(out, indent=1))
