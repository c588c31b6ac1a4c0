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


def test_bench_launches_itself_for_n_gt_1():
    """`python bench.py --gpus 2 ...` from a bare shell (no WORLD_SIZE): the parent must start the two ranks as CHILD
    processes under torch.distributed.run, the ranks must rendezvous on 127.0.0.1, and rank 0's line must come back
    through the parent.  --dry-run keeps the GPU out of it (this box has none): rendezvous, barrier and the MAX / SUM
    aggregation are the real code path of the N > 1 bench."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["metric"] == "dry_run" and out["value"] is None
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["elapsed_max_s"] == 2.0 and out["bytes"] == 3000.0          # MAX of (1, 2); SUM of (1000, 2000)
    assert out["pixels"] == 2 * out["batch_per_gpu"] * 512 * 768


def test_bench_defaults_match_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert a.gpus == 1 and a.batch == 0 and a.container == "auto"
    # streams per image of the timed container: one decoder workgroup per stream on its own compute unit, at most 10 (the bpp budget)
    assert bench.default_container(24) == "xrans10" and bench.default_container(32) == "xrans8" and bench.default_container(3) == "xrans10"
    assert bench.default_container(512) == "xrans1" and bench.default_container(24, n_cu=128) == "xrans5"
    # a launcher that started the wrong number of ranks is an error, not a silent single-rank run
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "started 3 ranks" in (p.stderr + p.stdout)


def test_bench_eight_rank_dry_run_configs4():
    """the 8-rank launch of configs[4] rehearsed without hardware -- rendezvous of 8 gloo ranks on 127.0.0.1, B = 32 per rank,
    the timed container for that batch (xrans8), MAX / SUM aggregation, the per-rank all_gather and the straggler ratio."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["batch_per_gpu"] == 32 and out["container"] == "xrans8"
    assert "configs[4]" in out["config"]["workload"]
    assert out["pixels"] == 256 * 512 * 768 and out["elapsed_max_s"] == 8.0 and out["bytes"] == 1000.0 * 36
    assert [r["rank"] for r in out["per_rank"]] == list(range(8)) and out["distinct_devices"] == 8
    # rank r reported (1 + r) s for one step of 32 images: its own rate, and slowest / fastest = 8
    for r in out["per_rank"]:
        assert abs(r["mpix_s"] - 32 * 512 * 768 / (1.0 + r["rank"]) / 1e6) < 1e-3
        assert abs(r["pcie_inclusive_mpix_s"] - 32 * 512 * 768 / (2.0 + r["rank"]) / 1e6) < 1e-3
    assert out["straggler_ratio"] == 8.0


def test_distinct_device_accounting():
    from llicti_amd import shard
    assert shard.distinct_devices([0x100, 0x200, 0x300]) == 3
    assert shard.distinct_devices([0x100, 0x100, 0x300]) == 2          # two ranks on one GPU: bench.py refuses without --allow-shared-gpu
    assert shard.distinct_devices([-1, -1]) == 2                        # unknown identities are not taken for the same device
    rows, strag = shard.per_rank_report([2.0, 4.0], [1.0, 1.0], 1_000_000, 2, [0x0101, -1], [0, 1])
    assert rows[0]["mpix_s"] == 1.0 and rows[1]["mpix_s"] == 0.5 and strag == 2.0
    assert rows[0]["pci"] == "0000:01:01" and rows[1]["pci"] is None
    assert shard.gather_per_rank([1.5, 2.5]) == [[1.5, 2.5]]           # world = 1: no process group needed
