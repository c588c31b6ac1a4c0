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
    # the timed container is a function of the image SIZE alone (llicti_amd.codec.image_streams): the same for every batch size and rank count
    assert bench.default_container(512, 768) == "xauto15" and bench.default_streams(512, 768) == 15
    assert bench.default_container(2160, 3840) == "xrans64" and bench.default_container(256, 256) == "xauto2" and bench.default_container(64, 64) == "rans1"
    # a launcher that started the wrong number of ranks is an error, not a silent single-rank run
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "started 3 ranks" in (p.stderr + p.stdout)


def test_bench_eight_rank_dry_run_configs4():
    """the 8-rank launch of configs[4] rehearsed without hardware -- rendezvous of 8 gloo ranks on 127.0.0.1, B = 32 per rank,
    the timed container (xrans15 for 768x512 whatever the batch), MAX / SUM aggregation, the per-rank all_gather and the straggler ratio."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["batch_per_gpu"] == 32 and out["container"] == "xauto15"
    assert "configs[4]" in out["config"]["workload"]
    # the like-for-like efficiency (against rank 0's solo run of the SAME per-GPU batch and container) and the ranks' NUMA placement are in the line
    assert out["n1_companion"]["batch"] == 32 and abs(out["efficiency_like_for_like"] - out["pixels"] / out["elapsed_max_s"] / 1e6 / (8 * out["n1_companion"]["value"])) < 1e-3
    assert [r["numa_node"] for r in out["per_rank"]] == [0, 0, 0, 0, 1, 1, 1, 1] and all(r["cpus_bound"] == 16 for r in out["per_rank"])
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


# ------------------------------------------------------------------------------------------------ LLICTIAgent.eval_model over several ranks
class _StubModel(torch.nn.Module):
    """Stands in for the device codec in the agent's multi-rank plumbing (no GPU here): stream lengths are a deterministic function of the
    image AND of a weight, so a rank that coded with other weights than rank 0's, or an image booked under the wrong index, changes the log."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(4))
        self.register_buffer("bound", torch.zeros(1))
        self.mode = 0
        self._last = None

    def compress(self, x):
        key = int(float(x.double().sum()) * 255) + int(float(self.w.sum()))
        h4, w4 = (x.shape[2] + 31) // 32, (x.shape[3] + 31) // 32
        rows = [[bytes([5, h4, w4]), bytes(12), bytes(2), bytes(3 * h4 * w4), b"", b"", b"", b"", b""]]
        for s in range(5):
            rows.append([bytes((key * (9 * s + k + 3)) % 97 + 1) for k in range(9)])
        self._last = x
        return rows, None

    def decompres(self, bytestream_list, devc=None, xorg=None):
        return self._last.clone()


def _agent_log_lines(agent_logger_names=("Agent", "Rate Loss")):
    import io
    import logging
    buf = io.StringIO()
    h = logging.StreamHandler(buf)
    h.setFormatter(logging.Formatter("%(name)s|%(message)s"))
    for n in agent_logger_names:
        lg = logging.getLogger(n)
        lg.setLevel(logging.INFO)
        lg.addHandler(h)
    return buf


def _strip_times(text):
    import re
    text = re.sub(r"Enc/Dec-Times:[0-9.]+/[0-9.]+", "Enc/Dec-Times:T/T", text)
    return re.sub(r"\(\d\d:\d\d:\d\d\)", "(clock)", text)


def _agent_images():
    return [make_image("noise", 32 + 8 * (i % 3), 40 + 4 * (i % 4), seed=300 + i) for i in range(7)]


def _agent_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    buf = _agent_log_lines()
    model = _StubModel()
    if rank == 0:
        with torch.no_grad():
            model.w.fill_(2.5)                       # "the checkpoint", read by rank 0 only
    agent = LLICTIAgent(default_config(test_data=_agent_images()), model=model)
    assert (agent.rank, agent.world) == (rank, world)
    assert float(model.w.sum()) == 10.0              # every rank codes with rank 0's weights (one broadcast)
    res = agent.run()
    assert [r["idx"] for r in res] == list(range(rank, 7, world))
    with open(os.path.join(outdir, f"log{rank}.txt"), "w") as f:
        f.write(buf.getvalue())
    if rank == 0:
        np.save(os.path.join(outdir, "all.npy"), np.array([[r["idx"], r["H"], r["W"], r["bpsp"]] for r in agent.all_results]))
    dist.destroy_process_group()


def test_agent_two_ranks_equal_one_rank():
    """SURVEY section 8(e) through the API: LLICTIAgent.eval_model started as two ranks (gloo here, RCCL on GPUs) shards the test set image i ->
    rank i mod 2, takes rank 0's weights by one broadcast, gathers the per-image records once, and rank 0 logs exactly what a one-rank run logs
    -- the reference's per-image lines in index order and the rate table (but for the wall-clock fields); the other rank logs no image line."""
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    buf = _agent_log_lines()
    model = _StubModel()
    with torch.no_grad():
        model.w.fill_(2.5)
    one = LLICTIAgent(default_config(test_data=_agent_images()), model=model)
    res1 = one.run()
    for n in ("Agent", "Rate Loss"):
        import logging
        logging.getLogger(n).handlers = [h for h in logging.getLogger(n).handlers if getattr(h, "stream", None) is not buf]
    log1 = _strip_times(buf.getvalue())
    assert len(res1) == 7 and log1.count("Check: Decoded img matches original") == 7
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_agent_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        log_r0 = _strip_times(open(os.path.join(d, "log0.txt")).read())
        log_r1 = open(os.path.join(d, "log1.txt")).read()
        allr = np.load(os.path.join(d, "all.npy"))
    pick = lambda t: [ln for ln in t.splitlines() if ln.startswith("Agent|") and "bpsp=" in ln] + [t[t.index("Rate Loss|"):]]      # noqa: E731
    assert pick(log_r0) == pick(log1)                   # lines in index order + the table: identical
    assert "bpsp=" not in log_r1 and "Rate Loss|" not in log_r1
    assert [int(v) for v in allr[:, 0]] == list(range(7))
    assert np.allclose(allr[:, 3], [r["bpsp"] for r in res1])


def test_numa_binding_from_pci_locality(tmp_path):
    """One process per GPU: a rank binds its host threads to the CPUs of the NUMA node its GPU hangs off before it allocates pinned staging
    buffers (llicti_amd.shard.bind_to_gpu_numa; bench.py at N > 1 and LLICTIAgent under the launcher).  Against a made-up sysfs tree: the PCI
    identity finds the node, the node its cpulist, the affinity is the intersection with what the process may use -- and nothing is touched
    where sysfs does not say."""
    from llicti_amd import shard
    ident = (0 << 16) | (0x43 << 8) | 0x00
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:43:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    allowed = sorted(os.sched_getaffinity(0))
    lo, hi = allowed[0], allowed[min(len(allowed) - 1, 3)]
    (node / "cpulist").write_text(f"{lo}-{hi},4000-4003\n")
    assert shard._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert shard.numa_of_pci(ident, str(tmp_path)) == (1, list(range(lo, hi + 1)) + [4000, 4001, 4002, 4003])
    calls = []
    out = shard.bind_to_gpu_numa(identity=ident, sysfs_root=str(tmp_path), setaffinity=lambda pid, cpus: calls.append((pid, list(cpus))))
    want = [c for c in range(lo, hi + 1) if c in allowed]
    assert calls == [(0, want)] and out["numa_node"] == 1 and out["cpus_bound"] == len(want)
    # no sysfs entry, an unknown identity, a single-node host (-1): the affinity is left alone
    for ident2, root in ((ident + 1, str(tmp_path)), (-1, str(tmp_path)), (ident, str(tmp_path / "nowhere"))):
        calls.clear()
        o = shard.bind_to_gpu_numa(identity=ident2, sysfs_root=root, setaffinity=lambda pid, cpus: calls.append(1))
        assert o["numa_node"] is None and o["cpus_bound"] == 0 and not calls
    (dev / "numa_node").write_text("-1\n")
    assert shard.bind_to_gpu_numa(identity=ident, sysfs_root=str(tmp_path), setaffinity=lambda pid, cpus: calls.append(1))["cpus_bound"] == 0 and not calls


def test_auto_container_modes_do_not_depend_on_the_partition():
    """Container "auto" through the model's own chooser (LLICTI.mode_for_batch -> llicti_amd.codec.auto_modes) on the reference's eval-set sizes: whatever
    the rank count and eval_batch -- i.e. whichever images end up in a call together -- every image gets the same ENCODER mode, the one its own size
    gives (the rest of the choice, the count adjusted by the image's last-stage cost, is made per image on the device: -m gpu,
    test_agent_auto_container_is_a_function_of_the_image / test_agent_two_ranks_on_gpu_equal_one_rank).  Round 5's chooser looked at the batch size, the
    CU count and a running mean of past content: an image's bytes depended on the partition."""
    import json
    from llicti_amd.codec import MODE_RANS_AUTO, image_mode, image_streams
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    sizes = [tuple(s) for s in json.load(open(os.path.join(ROOT, "tests", "golden", "eval_shapes.json")))["shapes"][:96]]
    model = LLICTI(default_config(container="auto"))

    def modes(world, eb):
        got = {}
        for rank in range(world):
            ids = list(range(rank, len(sizes), world))
            for k in range(0, len(ids), eb):
                batch = ids[k:k + eb]
                m = model.mode_for_batch(len(batch), None, sizes=[sizes[i] for i in batch])
                for j, i in enumerate(batch):
                    got[i] = m if isinstance(m, int) else m[j]
        return [got[i] for i in range(len(sizes))]
    base = modes(1, 1)
    assert base == [image_mode(h, w) for h, w in sizes] == [MODE_RANS_AUTO(image_streams(h, w)) for h, w in sizes]
    for world, eb in ((1, 24), (2, 24), (2, 5), (3, 7), (8, 24), (8, 3)):
        assert modes(world, eb) == base, (world, eb)
