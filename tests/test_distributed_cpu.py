"""N > 1 path on CPU: two gloo processes shard a list of images exactly as bench.py shards them over GPUs
(no data-path collective), code their shards (with the CPU oracle standing in for the device codec) and
aggregate with llicti_amd.shard.aggregate."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_state_dict
from helpers import make_image

N_IMAGES = 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from llicti_amd import shard
    from llicti_amd.weights import pack_state_dict
    from oracle import oracle as orc
    orc.set_threads(2)
    W = orc.Weights(pack_state_dict(load_state_dict("trainedlike")))
    r, w = shard.world_info()
    assert (r, w) == (rank, world)
    mine = shard.shard_indices(N_IMAGES, rank, world)
    nbytes = npix = 0
    shard.barrier()
    for i in mine:
        rgb = make_image("smooth", 40 + 8 * i, 48, seed=100 + i)
        bl = orc.encode_image(rgb, W)
        assert np.array_equal(orc.decode_image(bl, W), rgb)
        nbytes += sum(len(s) for row in bl for s in row)
        npix += rgb.shape[1] * rgb.shape[2]
    shard.barrier()
    agg = shard.aggregate(elapsed_s=1.0 + rank, n_bytes=nbytes, n_pixels=npix)
    np.save(os.path.join(outdir, f"rank{rank}.npy"), np.array([agg["elapsed_s"], agg["bytes"], agg["pixels"], nbytes, len(mine)]))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        res = [np.load(os.path.join(d, f"rank{r}.npy")) for r in range(world)]
    # single-process truth
    from llicti_amd.weights import pack_state_dict
    from oracle import oracle as orc
    W = orc.Weights(pack_state_dict(load_state_dict("trainedlike")))
    tot_b = tot_p = 0
    for i in range(N_IMAGES):
        rgb = make_image("smooth", 40 + 8 * i, 48, seed=100 + i)
        tot_b += sum(len(s) for row in orc.encode_image(rgb, W) for s in row)
        tot_p += rgb.shape[1] * rgb.shape[2]
    for r in res:
        assert r[0] == 2.0                       # MAX over ranks of (1.0, 2.0)
        assert r[1] == tot_b and r[2] == tot_p   # SUM over ranks == whole job
    assert sum(r[4] for r in res) == N_IMAGES and res[0][3] + res[1][3] == tot_b


def test_shard_indices_partition():
    from llicti_amd.shard import shard_indices, world_info, aggregate
    for n, world in [(256, 8), (24, 4), (5, 2), (3, 8)]:
        parts = [shard_indices(n, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert world_info() == (0, 1)
    a = aggregate(2.0, 1000, 4_000_000)
    assert a["mpix_s"] == 2.0 and a["bpp"] == 0.002
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)
