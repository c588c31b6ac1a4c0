#!/usr/bin/env python3
"""A/B of llicti_set_tuning("enc_chunk_images"): encode time of bench.py's batch with the level-0 launches whole or in sub-batches
(interleaved repeats on one box).  python tools/ab_chunk.py > gpurun_out/ab_chunk.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from llicti_amd.codec import HipCodec, mode_of_name
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

B, H, W = 24, 512, 768
dev = torch.device("cuda", 0)
torch.manual_seed(1337)
codec = HipCodec(dev)
codec.load_state_dict(LLICTI(default_config()).state_dict())
mode = mode_of_name(bench.default_container(H, W))
rgb = torch.from_numpy(bench.make_batch(B, H, W, 0)).to(dev)
cont, seg = codec.encode(rgb, mode=mode)
ref = cont.clone()
res = {}
for rep in range(4):
    for chunk in (0, 4, 6, 8, 12):
        codec.set_tuning("enc_chunk_images", chunk)
        codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
        torch.cuda.synchronize()
        assert torch.equal(cont[:, :1000000], ref[:, :1000000])
        t0 = time.perf_counter()
        for _ in range(5):
            codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
        torch.cuda.synchronize()
        res.setdefault(chunk, []).append(round((time.perf_counter() - t0) / 5 * 1e3, 4))
codec.set_profiling(True)
detail = {}
for chunk in (0, 8):
    codec.set_tuning("enc_chunk_images", chunk)
    codec.encode(rgb, mode=mode, out=cont, seg_len=seg)
    torch.cuda.synchronize()
    cat, per = codec.last_timing_detail()
    detail[chunk] = {k: round(v, 3) for k, v in cat.items() if v > 0}
print(json.dumps({"workload": f"{B}x{W}x{H} encode, ms per call by enc_chunk_images (0 = whole batch per launch)", "encode_ms": res, "profiled": detail}))
