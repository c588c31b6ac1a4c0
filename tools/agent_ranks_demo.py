#!/usr/bin/env python3
"""LLICTIAgent.eval_model as one rank of several (SURVEY section 8e through the API).  Start it with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P tools/agent_ranks_demo.py OUT.json [eval_batch]
(the launcher starts the ranks before anything touches the GPU) or alone (G = 1).  Rank 0 writes what it logged -- the per-image lines of ALL
images in index order and the rate table -- and the gathered records to OUT.json; with LLICTI_DIST_BACKEND=gloo the ranks may share one GPU
(a rehearsal of the plumbing, not a scaling measurement)."""
import io
import json
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from llicti_amd.agents.llicti_agent import LLICTIAgent  # noqa: E402
from llicti_amd.config import default_config  # noqa: E402

buf = io.StringIO()
h = logging.StreamHandler(buf)
h.setFormatter(logging.Formatter("%(name)s|%(message)s"))
for n in ("Agent", "Rate Loss"):
    logging.getLogger(n).setLevel(logging.INFO)
    logging.getLogger(n).addHandler(h)
sizes = [(96, 128), (67, 93), (128, 96), (96, 128), (150, 131), (97, 351), (64, 80), (96, 128), (33, 64)]
imgs = [np.random.default_rng(70 + i).integers(0, 256, size=(3, hh, ww), dtype=np.uint8) for i, (hh, ww) in enumerate(sizes)]
eb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
container = sys.argv[3] if len(sys.argv) > 3 else "xrans1"
if container == "auto":
    # container "auto": an image's container is a function of the image (its size, and -- picked by the encoder on the device -- what its last stage
    # costs), so rank 0's log of a sharded run is the one-rank log on ANY content: flat, natural-like and noise images of several sizes, interleaved
    from helpers import make_image  # noqa: E402
    sizes = [(256, 384), (256, 384), (192, 256), (256, 384), (321, 481), (256, 384), (160, 352), (256, 384), (192, 256)]
    kinds = ["flat", "smooth", "noise", "noise", "smooth", "flat", "noise", "smooth", "flat"]
    imgs = [np.full((3, hh, ww), 60 + 20 * i, np.uint8) if k == "flat" else make_image(k, hh, ww, 70 + i) for i, ((hh, ww), k) in enumerate(zip(sizes, kinds))]
cfg = default_config(test_data=imgs, container=container, **({"eval_batch": eb} if eb > 1 else {"eval_batch": 1}))
agent = LLICTIAgent(cfg)
res = agent.run()
import torch.distributed as dist  # noqa: E402
if agent.rank == 0:
    rows = agent.all_results if agent.world > 1 else res
    json.dump({"world": agent.world, "log": buf.getvalue(), "records": [[r["idx"], r["H"], r["W"], r["bpsp"], r["max_abs_err"]] for r in rows],
               "own": [r["idx"] for r in res], "container": container}, open(sys.argv[1], "w"))
if dist.is_available() and dist.is_initialized():
    dist.destroy_process_group()
