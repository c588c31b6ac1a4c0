"""python -m llicti_amd.cli encode IN.(png|ppm|jpg) OUT.llic [--container ac|auto|rans<M>|wrans<M>|xrans<M>] [--checkpoint model_best.pth.tar]
   python -m llicti_amd.cli decode IN.llic OUT.(png|ppm)
   python -m llicti_amd.cli info   IN.llic

File-level front end of the MI355X hot path (needs a GPU: there is no CPU fallback).  Without a checkpoint the
seed-1337 default init is used, as the reference does when `model_best.pth.tar` is missing (agents/base.py:78-80)."""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def _model(container, checkpoint):
    import torch
    from .config import default_config
    from .graphs.models.LLICTI_nets import LLICTI
    torch.manual_seed(1337)
    model = LLICTI(default_config(container=container)).to("cuda:0").eval()
    if checkpoint:
        from .weights import load_reference_state_dict
        load_reference_state_dict(model, torch.load(checkpoint, map_location="cuda:0")["state_dict"])
    return model, torch


def main(argv=None):
    ap = argparse.ArgumentParser(prog="llicti_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("encode"); e.add_argument("src"); e.add_argument("dst")
    e.add_argument("--container", default="ac"); e.add_argument("--checkpoint", default=None)
    d = sub.add_parser("decode"); d.add_argument("src"); d.add_argument("dst"); d.add_argument("--checkpoint", default=None)
    i = sub.add_parser("info"); i.add_argument("src")
    a = ap.parse_args(argv)
    from . import fileio
    if a.cmd == "info":
        from .codec import header_dims, mode_of_header, name_of_mode
        bl = fileio.read_llic(a.src)
        H, W = header_dims(bl[0][0] + bl[0][1] + bl[0][2])
        n = sum(len(s) for r in bl for s in r)
        mode = mode_of_header(bl)
        print(f"{a.src}: {W}x{H} RGB, container {name_of_mode(mode)}, {n} bytes, {8.0 * n / (H * W):.4f} bpp")
        return 0
    if a.cmd == "encode":
        rgb = fileio.read_image(a.src)
        model, torch = _model(a.container, a.checkpoint)
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0")
        t0 = time.time()
        bl, _ = model.compress(x)
        torch.cuda.synchronize()
        dt = time.time() - t0
        fileio.write_llic(a.dst, bl)
        n = sum(len(s) for r in bl for s in r)
        print(f"{a.src} -> {a.dst}: {rgb.shape[2]}x{rgb.shape[1]}, {n} bytes, {8.0 * n / (rgb.shape[1] * rgb.shape[2]):.4f} bpp, {dt:.3f} s")
        return 0
    bl = fileio.read_llic(a.src)
    model, torch = _model("ac", a.checkpoint)
    t0 = time.time()
    x = model.decompres(bl, torch.device("cuda:0"))
    torch.cuda.synchronize()
    dt = time.time() - t0
    fileio.write_image(a.dst, (x[0] * 255).round().to(torch.uint8).cpu().numpy())
    print(f"{a.src} -> {a.dst}: {x.shape[3]}x{x.shape[2]}, {dt:.3f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
