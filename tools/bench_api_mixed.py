#!/usr/bin/env python3
"""bench.py's `api_path_mixed` leg alone (the drop-in API on the reference's own eval-set sizes): a quick way to time it while working on the
host side of the batched path.  Usage: python tools/bench_api_mixed.py [eval_batch] [n_images]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(1337)
for rep in range(2):
    r = bench.api_path_mixed_leg(torch, dev, B, n)
    print(json.dumps({"mpix_s": r["mixed_batches"]["mpix_s"], "repeats": r["repeats_mpix_s"], "gpu_enc_ms_per_image": r["mixed_batches"]["gpu_enc_ms_per_image"],
                      "gpu_dec_ms_per_image": r["mixed_batches"]["gpu_dec_ms_per_image"], "bpsp": r["mixed_batches"]["bpsp"]}), flush=True)
