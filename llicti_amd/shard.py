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
