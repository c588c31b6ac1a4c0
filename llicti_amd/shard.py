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
