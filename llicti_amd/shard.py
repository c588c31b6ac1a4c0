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


def _parse_cpulist(text):
    """"0-15,128-143" -> [0, ..., 15, 128, ..., 143] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def numa_of_pci(identity, sysfs_root="/sys"):
    """(NUMA node, its CPUs) of the PCI device `identity` (device_identity(): domain << 16 | bus << 8 | device), read from sysfs:
    <root>/bus/pci/devices/DDDD:BB:DD.0/numa_node and <root>/devices/system/node/node<N>/cpulist.  (None, []) where the platform does not say
    (a single-node host reports -1; containers may hide sysfs)."""
    import os
    if identity is None or int(identity) < 0:
        return None, []
    ident = int(identity)
    name = "%04x:%02x:%02x.0" % (ident >> 16, (ident >> 8) & 0xFF, ident & 0xFF)
    try:
        with open(os.path.join(sysfs_root, "bus", "pci", "devices", name, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, []
        with open(os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            return node, _parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, []


def bind_to_gpu_numa(torch_device=None, identity=None, sysfs_root="/sys", setaffinity=None):
    """One process per GPU on an 8-GPU node: run this rank's host threads on the CPUs of the NUMA node its GPU hangs off, BEFORE the pinned staging
    buffers are allocated -- pinned pages are placed by first touch, so they then lie on the GPU's own node and the uint8 RGB / container copies of
    the PCIe-inclusive and agent paths do not cross the socket interconnect (MI355X nodes: two sockets, four GPUs each).  Returns
    {"numa_node", "cpus_bound", "cpus"}; leaves the affinity alone (cpus_bound 0) where sysfs does not say or the node's CPUs are not in the
    process's allowed set.  `identity` / `sysfs_root` / `setaffinity` are for tests."""
    import os
    ident = device_identity(torch_device) if identity is None else identity
    node, cpus = numa_of_pci(ident, sysfs_root)
    out = {"numa_node": node, "cpus_bound": 0, "cpus": None}
    if node is None or not cpus:
        return out
    try:
        allowed = set(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return out
    mine = sorted(set(cpus) & allowed)
    if not mine:
        return out
    try:
        (setaffinity or os.sched_setaffinity)(0, mine)
    except OSError:
        return out
    out.update({"cpus_bound": len(mine), "cpus": "%d-%d" % (mine[0], mine[-1]) if mine == list(range(mine[0], mine[-1] + 1)) else ",".join(map(str, mine[:8])) + ("..." if len(mine) > 8 else "")})
    return out


def distinct_devices(identities):
    """Number of distinct physical devices among the ranks' identities (unknown identities, -1, count as distinct)."""
    known = [int(i) for i in identities if int(i) >= 0]
    return len(set(known)) + sum(1 for i in identities if int(i) < 0)


def per_rank_report(elapsed_s, pcie_elapsed_s, pixels_per_step, steps, identities, devices, numa=None):
    """[{rank, device, mpix_s, pcie_inclusive_mpix_s, numa_node, cpus_bound}] and the straggler ratio (slowest / fastest rank's time of the timed
    steps).  numa: per rank (node or -1, CPUs the rank bound itself to) from bind_to_gpu_numa()."""
    rows = []
    for r, (t, tp) in enumerate(zip(elapsed_s, pcie_elapsed_s)):
        rows.append({"rank": r, "device": int(devices[r]), "pci": (None if identities[r] < 0 else "%04x:%02x:%02x" % (int(identities[r]) >> 16, (int(identities[r]) >> 8) & 0xFF, int(identities[r]) & 0xFF)),
                     "mpix_s": round(pixels_per_step * steps / t / 1e6, 3) if t > 0 else None,
                     "pcie_inclusive_mpix_s": round(pixels_per_step / tp / 1e6, 3) if tp > 0 else None})
        if numa is not None:
            rows[-1]["numa_node"] = (None if int(numa[r][0]) < 0 else int(numa[r][0]))
            rows[-1]["cpus_bound"] = int(numa[r][1])
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
