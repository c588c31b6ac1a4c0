#!/usr/bin/env python3
"""Micro-benchmark of the band CNN kernel alone (A/B work on variants: LLICTI_HIP_SO=... tools/bench_cnn.py).
Prints ms and TFLOP/s per (level, band) for B x H x W planes of noise; no correctness check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llicti_amd.codec import HipCodec
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

B, H, W = int(os.environ.get("B", 24)), 512, 768
torch.manual_seed(1337)
codec = HipCodec(torch.device("cuda", 0))
codec.load_state_dict(LLICTI(default_config()).state_dict())
fpl = (torch.randint(-255, 256, (B, 3, H, W), device="cuda").float() / 255.0).contiguous()
MAC = {0: 53152, 1: 61600, 2: 78496}     # per position: layer 0 (K0 x 352) + 4 x 88 x 88 + 4 x 88 x 15; sum = 193,248
tot_ms, tot_fl = 0.0, 0.0
for lvl in (0, 1, 2, 3, 4):
    for band in (0, 1, 2):
        codec.band_params(fpl, lvl, band)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            out = codec.band_params(fpl, lvl, band)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        pos = B * out.shape[2] * out.shape[3]                   # [B, 64 planes, h, w]
        fl = 2.0 * MAC[band] * pos
        tot_ms += ms; tot_fl += fl
        print(f"lvl {lvl} band {band}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
print(f"total {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
