#!/usr/bin/env python3
"""bench.py -- LLICTI encode+decode throughput on MI355X (BASELINE.json metric: MPix/s encode+decode).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input on every rank: encode the batch
(uint8 RGB already resident in HBM -> containers in HBM) and decode it again (containers in HBM -> uint8
RGB in HBM).  Workload: 24 x 768x512 RGB per GPU in the rANS container (BASELINE.json configs[2]: "Batch
of 24x 768x512 RGB, HIP rANS replacing torchac end-to-end", the shape the north-star target is quoted on;
--batch 1 gives configs[1]'s single image, --container ac the torchac-compatible container), i.i.d.
uniform noise, seeds 0..B-1 per rank, weights = seed-1337 default init (the
reference does the same when its checkpoint is missing).  Images shard across ranks with no data-path collective ("weak" scaling): the
only collectives are the timing barrier and a MAX / SUM of scalars at the end.

One JSON line is printed by rank 0.  `value` = pixels of all ranks x K / max-over-ranks time of the K
steps.  `roofline` is for the dominant kernel (the fp32-MFMA interpolator CNN): algorithmic FLOPs of the
launches in one encode+decode / their summed HIP-event durations, measured live in extra profiled steps
after the timed region.  `cpu_baseline` times the CPU oracle in the reference's structure (materialised
Lp-entry tables, single-thread coder) on the host cores, on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAC_PER_POSITION = 193248            # 84,480 layer-0 + 92,928 mid + 15,840 out
PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0


def positions_per_image(H, W):
    n = 0
    for lvl in range(5):
        st = 1 << lvl
        Hl, Wl = (H + st - 1) // st, (W + st - 1) // st
        n += ((Hl + 1) // 2) * ((Wl + 1) // 2)
    return n


def make_batch(B, H, W, seed0):
    return np.stack([np.random.default_rng(seed0 + i).integers(0, 256, size=(3, H, W), dtype=np.uint8) for i in range(B)])


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes of this same command (FETCH_SIZE and
    WRITE_SIZE in separate passes; collected by tools/collect_pmc.sh and committed as profiles/<round>/pmc_traffic.json --
    counters cannot be read from inside this process).  None when no such file is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            return json.load(fh)["band_params_kernel"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(H, W):
    """Oracle ("port"), reference structure, on a bounded sample: ONE H x W image of the same workload."""
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    from llicti_amd.weights import pack_state_dict
    from oracle import oracle as orc
    import torch
    torch.manual_seed(1337)
    sd = LLICTI(default_config()).state_dict()
    Wt = orc.Weights(pack_state_dict(sd))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)        # the oracle's OpenMP loops stop scaling well before that
    orc.set_threads(cores)
    # bounded sample: whole images of the same workload until >= 10 s of CPU work (at most 8 images)
    n_img, t_enc, t_dec, bl0 = 0, 0.0, 0.0, None
    while n_img < 8 and t_enc + t_dec < 10.0:
        rgb = make_batch(1, H, W, n_img)[0]
        t0 = time.time()
        bl = orc.encode_image(rgb, Wt, full_tables=True)
        t1 = time.time()
        rec = orc.decode_image(bl, Wt, full_tables=True)
        t2 = time.time()
        assert np.array_equal(rec, rgb)
        t_enc += t1 - t0
        t_dec += t2 - t1
        if bl0 is None:
            bl0 = bl
        n_img += 1
    return {"value": round(n_img * H * W / 1e6 / (t_enc + t_dec), 5), "unit": "MPix/s", "cores": cores, "kind": "port",
            "sample": f"{n_img} images {W}x{H} uniform-noise RGB (seeds 0..{n_img - 1}), encode {t_enc:.2f}s + decode {t_dec:.2f}s, "
                      "materialised Lp-entry tables (OpenMP) + single-thread range coder",
            "enc_s": round(t_enc, 3), "dec_s": round(t_dec, 3)}, bl0


def natural_like_leg(torch, dev, B, H, W, mode):
    """uniform noise + sigma-floor random weights is the worst case for alphabet width and far
    from natural statistics, so the same shapes are also run on a SMOOTH set (low-pass noise + gradient, seed-fixed,
    generated on the GPU) with the "trained-like" weights of tests/golden (sigma of a few grey levels): Lp and bpp
    in a natural range.  Informational; the headline stays on BASELINE.json's uniform-noise workload."""
    from llicti_amd.codec import HipCodec
    wfile = os.path.join(ROOT, "tests", "golden", "weights_trainedlike.npz")
    if not os.path.exists(wfile):
        return None
    codec = HipCodec(dev)
    codec.load_state_dict({k: v for k, v in np.load(wfile).items()})
    g = torch.Generator(device=dev).manual_seed(2024)
    x = torch.randn((B, 3, H + 32, W + 32), device=dev, generator=g)
    k = torch.ones((3, 1, 9, 9), device=dev) / 81.0
    for _ in range(2):
        x = torch.nn.functional.conv2d(x, k, padding=4, groups=3)
    x = x[:, :, 16:16 + H, 16:16 + W]
    lum = x[:, 0:1] * 900.0
    ramp = torch.linspace(-40, 40, W, device=dev)[None, None, None, :]
    img = 128 + lum + x * 250.0 + ramp + torch.randn((B, 3, H, W), device=dev, generator=g) * 2.0
    rgb = img.round().clamp(0, 255).to(torch.uint8).contiguous()
    stride = codec.max_container_bytes(H, W)
    cont = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    seg = torch.zeros((B, 49), dtype=torch.int32, device=dev)
    rec = torch.empty_like(rgb)

    def e():
        codec.encode(rgb, mode=mode, out=cont, seg_len=seg)

    def d():
        codec.decode(cont, seg, H, W, mode=mode, out=rec)
    e(); d(); codec.check()
    assert torch.equal(rec, rgb)

    def timed(fn, n=3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    te, td = timed(e), timed(d)
    mp = B * H * W / 1e6
    mm = None
    try:
        _, _, mmt = codec.lift(rgb[:1])
        mm = [int(v) for v in mmt[0].cpu().numpy()]
    except Exception:
        pass
    codec.close()
    return {"workload": f"{B}x{W}x{H} smooth synthetic RGB, trained-like weights (tests/golden)", "encdec_mpix_s": round(mp / (te + td), 3),
            "enc_mpix_s": round(mp / te, 3), "dec_mpix_s": round(mp / td, 3), "bpp": round(8.0 * float(seg.sum().item()) / (B * H * W), 4),
            "chroma_range_image0": mm}


def table_kernel_roofline(codec, torch, H=2160, W=3840):
    """BASELINE.json configs[3]: one 3840x2160 image, the full-table CDF kernel (the reference's get_cdfs +
    _convert_to_int_and_normalize, LLICTI_nets.py:938-983) at level 0 -- HBM-write bound by construction.
    Algorithmic bytes per coded symbol: 2 * row_stride written (264 entries for Y, 512 for Co/Cg) + 256 B of
    CNN outputs read; duration from events on the launch stream."""
    rgb = torch.from_numpy(make_batch(1, H, W, 0)).cuda()
    planes, fplanes, mm = codec.lift(rgb)
    params = codec.band_params(fplanes, 0, 0)
    out = {}
    tot_b, tot_ms = 0.0, 0.0
    for clr, stride in ((0, 264), (1, 512), (2, 512)):
        codec.cdf_tables(planes, params, mm, 0, 0, clr, row_stride=stride)       # warm-up
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3
        tabs = None
        e0.record()
        for _ in range(n):
            tabs = None
            tabs = codec.cdf_tables(planes, params, mm, 0, 0, clr, row_stride=stride)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rows = tabs.shape[1]
        nbytes = rows * (2.0 * stride + 256.0)
        out[("Y", "Co", "Cg")[clr]] = {"ms": round(ms, 3), "GB_s": round(nbytes / ms / 1e6, 1)}
        tot_b += nbytes
        tot_ms += ms
        del tabs
    ach = tot_b / tot_ms / 1e6
    return {"bound": "hbm", "kernel": "cdf_table_kernel", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
            "workload": f"{W}x{H} image, level 0 band x11: {rows} rows x (Y 264 | Co 512 | Cg 512) uint16 entries; BASELINE.json configs[3]",
            "per_channel": out, "bytes_per_launch_avg": tot_b / 3.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=24, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--container", default="rans16", help="rans<M> (M streams per image) or ac (torchac-compatible)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed AC-container and 4K table-kernel legs (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    local_dev = local_rank % max(1, torch.cuda.device_count())   # identity on a full node
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        # RCCL over xGMI; used for the barrier + two scalar all-reduces only (no data-path collective)
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)                      # forces communicator set-up now, outside any timed region
                torch.cuda.synchronize()
            except Exception as e:                          # the collectives carry three scalars: gloo is a safe stand-in
                print(f"[bench] RCCL init failed on rank {rank} ({e!r}); falling back to gloo", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                args.backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(args.backend)

    from llicti_amd.codec import MODE_AC, MODE_RANS, HipCodec, container_to_bytestream_list
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI

    B, H, W = args.batch, args.height, args.width
    mode = MODE_AC if args.container == "ac" else MODE_RANS(int(args.container[4:]))
    torch.manual_seed(1337)
    sd = LLICTI(default_config()).state_dict()                # seed-1337 default init, identical on every rank
    codec = HipCodec(dev)
    codec.load_state_dict(sd)

    rgb_h = make_batch(B, H, W, seed0=rank * B)
    rgb = torch.from_numpy(rgb_h).to(dev)
    stride = codec.max_container_bytes(H, W)
    cont = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    seg = torch.zeros((B, 49), dtype=torch.int32, device=dev)
    rec = torch.empty_like(rgb)

    from llicti_amd import shard
    barrier = shard.barrier

    def enc(m=mode, c=None, s_=None):
        return codec.encode(rgb, mode=m, out=cont if c is None else c, seg_len=seg if s_ is None else s_)

    def dec(m=mode, c=None, s_=None):
        return codec.decode(cont if c is None else c, seg if s_ is None else s_, H, W, mode=m, out=rec)

    def step():
        enc()
        dec()

    # correctness outside the timed region: lossless, and rank 0's first image bit-exact to the CPU oracle
    step()
    codec.check()
    assert torch.equal(rec, rgb), "decode(encode(x)) != x"
    seg_h = seg.cpu().numpy()
    total_bytes = int(seg_h.sum())
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # separate encode / decode timings (informational)
    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    t_enc = timed(enc, max(1, min(3, args.steps)))
    t_dec = timed(dec, max(1, min(3, args.steps)))

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    codec.check()

    # dominant-kernel timing with HIP events on the launch stream, in extra (untimed) profiled steps
    codec.set_profiling(True)
    cnn_ms, cnn_launches, call_ms = 0.0, 0, 0.0
    for fn in (enc, dec):
        fn()
        torch.cuda.synchronize()
        ms, n = codec.last_timing()
        cnn_ms += ms[1]
        cnn_launches += n
        call_ms += ms[0]
    codec.set_profiling(False)

    # the other container on the same batch (untimed region; informational): the torchac-compatible AC
    # container is what bit-exactness with the reference's format is claimed on
    other = {}
    extras = (world == 1) and not args.no_extras         # informational legs: N = 1 only (as cpu_baseline)
    if mode != MODE_AC and extras:
        cont2 = torch.empty_like(cont)
        seg2 = torch.zeros_like(seg)
        enc(MODE_AC, cont2, seg2)
        dec(MODE_AC, cont2, seg2)
        codec.check()
        assert torch.equal(rec, rgb)
        ta = timed(lambda: enc(MODE_AC, cont2, seg2), 1)
        tb = timed(lambda: dec(MODE_AC, cont2, seg2), 1)
        ac_bytes = int(seg2.sum().item())
        other = {"ac_container": {"enc_mpix_s": round(B * H * W / ta / 1e6, 3), "dec_mpix_s": round(B * H * W / tb / 1e6, 3),
                                  "encdec_mpix_s": round(B * H * W / (ta + tb) / 1e6, 3),
                                  "bpp": round(8.0 * ac_bytes / (B * H * W), 4),
                                  "bpp_delta_rans_minus_ac": round(8.0 * (total_bytes - ac_bytes) / (B * H * W), 4)}}
        seg_ac_h = seg2.cpu().numpy()
        cont_ac0 = cont2[0].cpu().numpy()
    elif mode == MODE_AC:
        seg_ac_h, cont_ac0 = seg_h, cont[0].cpu().numpy()
    else:
        seg_ac_h = cont_ac0 = None
    tab_roof = nat = None
    if rank == 0 and extras:
        nat = natural_like_leg(torch, dev, B, H, W, mode)
    if rank == 0 and extras:
        cont2 = seg2 = None
        torch.cuda.empty_cache()
        tab_roof = table_kernel_roofline(codec, torch)

    # whole job: time = MAX over ranks, bytes / pixels = SUM over ranks (the only collectives of the run)
    agg = shard.aggregate(elapsed, total_bytes, B * H * W, device=dev if (world == 1 or args.backend == "nccl") else "cpu")
    elapsed = agg["elapsed_s"]

    if rank == 0:
        pix = agg["pixels"]
        value = pix * args.steps / elapsed / 1e6
        flops = 2.0 * MAC_PER_POSITION * positions_per_image(H, W) * B * 2      # rank 0's launches: encode + decode pass
        achieved = flops / (cnn_ms * 1e-3) / 1e12 if cnn_ms > 0 else 0.0
        out = {
            "metric": "MPix/s encode+decode", "value": round(value, 3), "unit": "MPix/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{B}x{W}x{H} uniform-noise RGB per GPU, container {args.container}, seed-1337 weights"
                                   + ("; BASELINE.json configs[2]" if (B == 24 and mode != MODE_AC) else ""),
                       "batch_per_gpu": B, "height": H, "width": W, "container": args.container,
                       "sharding": f"images/{world}gpu"},
            "enc_mpix_s": round(B * H * W / t_enc / 1e6, 3), "dec_mpix_s": round(B * H * W / t_dec / 1e6, 3),
            "bpp": round(agg["bpp"], 4),
            "roofline": {"bound": "mfma", "kernel": "band_params_kernel<0|1|2> (fp32 MFMA 16x16x4)",
                         "achieved": round(achieved, 3), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4), "traffic": None,
                         "launches": cnn_launches, "kernel_ms_per_step": round(cnn_ms, 3),
                         "call_ms_profiled": round(call_ms, 3),
                         "flop_per_step": flops},
        }
        out["roofline"]["traffic"] = pmc_traffic()
        out.update(other)
        if nat is not None:
            out["natural_like"] = nat
        if tab_roof is not None:
            out["roofline_cdf_table"] = tab_roof
        if not args.no_cpu_baseline and world == 1:
            cb, bl = cpu_baseline(H, W)
            if cont_ac0 is not None:
                # the same image through the HIP path (AC container) must give the oracle's bytes
                got = container_to_bytestream_list(cont_ac0, seg_ac_h[0])
                cb["bitexact_vs_hip"] = bool(got == bl)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
