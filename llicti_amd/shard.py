"""Multi-GPU harness helpers: images shard across ranks, nothing else does.

Image i goes to rank i mod world (every image is an independent unit: own header, own min/max, own
streams), weights are replicated, and the hot path needs **no collective**.  The only communication is the
timing barrier and one MAX / SUM all-reduce of scalars at the end (RCCL on GPUs: backend "nccl"; "gloo" in
the CPU tests).  Works unchanged when torch.distributed is not initialised (world = 1).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items: int, rank: int, world: int):
    """Indices of the images rank `rank` codes: i = rank, rank + world, ..."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def aggregate(elapsed_s: float, n_bytes: int, n_pixels: int, device=None):
    """Whole-job figures from per-rank ones: time = MAX over ranks, bytes / pixels = SUM over ranks."""
    dev = device if device is not None else "cpu"
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    s = torch.tensor([float(n_bytes), float(n_pixels)], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_bytes, total_pixels = float(s[0].item()), float(s[1].item())
    return {"elapsed_s": elapsed, "bytes": total_bytes, "pixels": total_pixels,
            "mpix_s": total_pixels / elapsed / 1e6 if elapsed > 0 else 0.0,
            "bpp": 8.0 * total_bytes / total_pixels if total_pixels > 0 else 0.0}


def gather_per_rank(values, device=None):
    """all_gather of a few scalars per rank -> list (rank-major) of lists of floats.  The per-rank detail of the N > 1 bench line:
    a straggler or a NUMA-far rank is invisible in SUM pixels / MAX time alone."""
    dev = device if device is not None else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized():
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)
    else:
        outs = [t]
    return [[float(v) for v in o.cpu()] for o in outs]


def device_identity(torch_device):
    """(pci domain, bus, device) of a CUDA/HIP device as one integer, or -1 when the runtime does not say.  Two ranks with the same
    non-negative identity share a GPU whatever their local device indices are (HIP_VISIBLE_DEVICES set per rank, containers)."""
    try:
        p = torch.cuda.get_device_properties(torch_device)
        dom, bus, devn = (getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is None:
            return -1
        return (int(dom or 0) << 16) | (int(bus) << 8) | int(devn or 0)
    except Exception:
        return -1


def distinct_devices(identities):
    """Number of distinct physical devices among the ranks' identities (unknown identities, -1, count as distinct)."""
    known = [int(i) for i in identities if int(i) >= 0]
    return len(set(known)) + sum(1 for i in identities if int(i) < 0)


def per_rank_report(elapsed_s, pcie_elapsed_s, pixels_per_step, steps, identities, devices):
    """[{rank, device, mpix_s, pcie_inclusive_mpix_s}] and the straggler ratio (slowest / fastest rank's time of the timed steps)."""
    rows = []
    for r, (t, tp) in enumerate(zip(elapsed_s, pcie_elapsed_s)):
        rows.append({"rank": r, "device": int(devices[r]), "pci": (None if identities[r] < 0 else "%04x:%02x:%02x" % (int(identities[r]) >> 16, (int(identities[r]) >> 8) & 0xFF, int(identities[r]) & 0xFF)),
                     "mpix_s": round(pixels_per_step * steps / t / 1e6, 3) if t > 0 else None,
                     "pcie_inclusive_mpix_s": round(pixels_per_step / tp / 1e6, 3) if tp > 0 else None})
    ts = [t for t in elapsed_s if t > 0]
    return rows, (round(max(ts) / min(ts), 4) if ts else None)


# ---------------------------------------------------------------------- the eval driver over several ranks
def init_from_env(device=None, backend=None):
    """Join the job `python -m torch.distributed.run` started (RANK / WORLD_SIZE / MASTER_* in the environment) unless a process group exists
    already; returns (rank, world).  Call it BEFORE anything touches the GPU in this process only for the launch itself -- the launcher has
    done that; here the group is created on the device the rank will use (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU)."""
    import os
    if dist.is_available() and dist.is_initialized():
        return world_info()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    on_gpu = device is not None and torch.device(device).type == "cuda"
    backend = backend or os.environ.get("LLICTI_DIST_BACKEND") or ("nccl" if on_gpu else "gloo")      # (the variable: rehearsals of several ranks on ONE GPU need gloo)
    dist.init_process_group(backend, **({"device_id": torch.device(device)} if backend == "nccl" else {}))
    return world_info()


def broadcast_module_state(module, device=None):
    """One broadcast from rank 0 of everything `module.state_dict()` holds (0.79 MB for LLICTI: parameters and buffers flattened into one float32
    buffer): a checkpoint is read once, by rank 0, and every rank codes with exactly its weights.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    sd = module.state_dict()
    keys = sorted(k for k, v in sd.items() if torch.is_tensor(v) and v.numel() > 0)
    if not keys:
        return 0
    dev = device if device is not None else sd[keys[0]].device
    if dist.get_backend() == "gloo":
        dev = "cpu"
    flat = torch.cat([sd[k].detach().to(device=dev, dtype=torch.float32).reshape(-1) for k in keys])
    dist.broadcast(flat, src=0)
    pos = 0
    with torch.no_grad():
        for k in keys:
            n = sd[k].numel()
            sd[k].copy_(flat[pos:pos + n].reshape(sd[k].shape).to(sd[k].dtype))
            pos += n
    return int(flat.numel())


def gather_records(rows, width, device=None):
    """One all_gather of the ranks' per-image records: `rows` = list of `width` floats per image this rank coded -> the records of ALL ranks
    (float64 [n_total, width], rank-major; the caller sorts by its index column).  Ranks may hold different numbers of images."""
    dev = device if device is not None else "cpu"
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return torch.tensor(rows, dtype=torch.float64).reshape(-1, width)
    world = dist.get_world_size()
    n = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    cap = max(int(c.item()) for c in counts)
    mine = torch.zeros((max(cap, 1), width), dtype=torch.float64, device=dev)
    if rows:
        mine[:len(rows)] = torch.tensor(rows, dtype=torch.float64, device=dev).reshape(-1, width)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    return torch.cat([o[:int(c.item())].cpu() for o, c in zip(outs, counts)], dim=0)
