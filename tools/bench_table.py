#!/usr/bin/env python3
"""The full-table CDF kernel alone on one 3840x2160 image, level 0 (BASELINE.json configs[3]); for A/B and PMC runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llicti_amd.codec import HipCodec
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI
import bench
torch.manual_seed(1337)
codec = HipCodec(torch.device("cuda", 0))
wname = os.environ.get("WEIGHTS", "rand")
if wname == "rand":
    codec.load_state_dict(LLICTI(default_config()).state_dict())
else:
    codec.load_state_dict({k: v for k, v in np.load(os.path.join(bench.ROOT, "tests", "golden", "weights_trainedlike.npz")).items()})
r = bench.table_kernel_roofline(codec, torch)
print(wname, r["achieved"], "GB/s", r["per_channel"])
