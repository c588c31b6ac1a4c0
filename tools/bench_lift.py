#!/usr/bin/env python3
"""Micro-benchmark of the lift kernel (A/B of variants: LLICTI_HIP_SO=... tools/bench_lift.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llicti_amd.codec import HipCodec
B, H, W = 24, 512, 768
codec = HipCodec(torch.device("cuda", 0))
rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda")
codec.lift(rgb); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    codec.lift(rgb)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"lift {ms*1e3:.1f} us per call (incl. torch.empty + minmax init), {B*H*W*21/ms/1e6:.0f} GB/s algorithmic")
