#!/usr/bin/env python3
"""Large-configuration round trips on one MI355X (not part of the test suite: minutes of GPU time, tens of GB):
BASELINE.json configs[4]'s whole 256-image batch on ONE GPU, a 64-image batch through the AC container, and the
format's largest image (8160 x 8160)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llicti_amd.codec import HipCodec, MODE_AC, MODE_RANS, MODE_RANS_AUTO, auto_modes
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

torch.manual_seed(1337)
codec = HipCodec(torch.device("cuda", 0))
codec.load_state_dict(LLICTI(default_config()).state_dict())
results = []
def run(B, H, W, mode, name):
    g = torch.Generator(device="cuda").manual_seed(B + H)
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
    codec.workspace(B, H, W, mode)                   # the (tens of GB) allocation is not part of the timing
    torch.cuda.synchronize(); t0 = time.time()
    cont, seg = codec.encode(rgb, mode=mode)
    codec.check(); torch.cuda.synchronize(); t1 = time.time()
    if mode & 0x10000:                               # the encoder's "auto" count: the decoder takes it from the headers (one count per call here)
        dm = sorted(set(codec.container_modes(cont)))
        assert len(dm) == 1, dm
        mode = dm[0]
        codec.workspace(B, H, W, mode)
    codec.poison_workspace()
    torch.cuda.synchronize(); t1b = time.time()
    rec = codec.decode(cont, seg, H, W, mode=mode)
    codec.check(); torch.cuda.synchronize(); t2 = time.time()
    ok = bool(torch.equal(rec, rgb))
    mp = B * H * W / 1e6
    print(f"{name}: B={B} {W}x{H} ok={ok} enc {mp/(t1-t0):.1f} MPix/s dec {mp/(t2-t1b):.1f} MPix/s bpp {8.0*float(seg.sum())/(B*H*W):.3f} "
          f"workspace {codec._ws.numel()/2**30:.2f} GiB peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    assert ok
    results.append({"case": name, "B": B, "W": W, "H": H, "lossless": ok, "enc_mpix_s": round(mp / (t1 - t0), 1), "dec_mpix_s": round(mp / (t2 - t1b), 1),
                    "bpp": round(8.0 * float(seg.sum()) / (B * H * W), 4), "workspace_GiB": round(codec._ws.numel() / 2**30, 2)})
    del rgb, cont, seg, rec
    codec._ws = None
    torch.cuda.empty_cache()
run(256, 512, 768, MODE_RANS_AUTO(15), "configs[4] batch on one GPU (xauto15: container auto, the encoder picks 20 streams per image on this noise)")
run(256, 512, 768, MODE_RANS_AUTO(15), "same, warm")
run(32, 512, 768, MODE_RANS_AUTO(15), "configs[4] per-GPU batch (32 images, xauto15)")
run(256, 512, 768, MODE_RANS(8, wide=2), "256 images, 8 xwide streams per image (round 5's per-GPU container of configs[4])")
run(256, 512, 768, MODE_RANS(1, wide=2), "256 images, ONE xwide stream per image")
run(256, 512, 768, MODE_RANS(8), "256 images, narrow streams (rans8)")
run(64, 512, 768, MODE_AC, "AC container")
run(1, 8160, 8160, MODE_RANS(128, wide=2), "largest image, 128 xwide streams (four per segment; the 32-bit plane offsets at their limit)")
run(1, 8160, 8160, MODE_RANS(64, wide=2), "largest image, 64 xwide streams")
run(1, 8160, 8160, MODE_RANS(32, wide=2), "largest image, 32 xwide streams")
run(1, 8160, 8160, MODE_RANS(14, wide=2), "largest image, 14 xwide streams")
run(1, 8160, 8160, MODE_RANS(128), "largest image, 128-stream latency mode")
run(1, 8160, 8160, MODE_RANS(14, wide=1), "largest image, 14 wide streams")
run(3, 2160, 3840, MODE_RANS(64, wide=2), "three 4K images, 64 xwide streams each")
run(24, 512, 768, MODE_RANS_AUTO(15), "bench batch in the timed encoder mode (xauto15)")
run(24, 512, 768, MODE_RANS(10, wide=2), "bench batch, 10 xwide streams (round 5's timed container, v4 layout)")
run(24, 512, 768, MODE_RANS(10, wide=1), "bench batch, wide streams (wrans10)")
run(24, 512, 768, MODE_RANS(10), "bench batch, narrow streams (rans10)")
run(7, 1055, 2049, MODE_RANS(11, wide=2), "odd-size images, 11 xwide streams")
run(5, 1055, 2049, MODE_RANS(7, wide=1), "odd-size images, 7 wide streams")
run(2, 2160, 3840, MODE_AC, "two 4K images, AC container")
def run_mixed(n, mode, name, big=False):
    """round 5: a batch of MIXED sizes at the sizes of the reference's own test set (or, big: large odd sizes incl. the format's widest)"""
    shapes = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eval_shapes.json")))["shapes"]
    sh = [(2160, 3840), (1055, 2049), (33, 8160), (8160, 33), (4097, 1031), (512, 768)] if big else shapes[100:100 + n]
    Hs, Ws = [h for h, _ in sh], [w for _, w in sh]
    g = torch.Generator(device="cuda").manual_seed(n)
    flat = torch.randint(0, 256, (sum(3 * h * w for h, w in sh),), dtype=torch.uint8, device="cuda", generator=g)
    if mode == "auto":
        mode = auto_modes(list(zip(Hs, Ws)))         # container "auto": each image's own encoder mode
    codec.workspace_v(Hs, Ws, mode)
    torch.cuda.synchronize(); t0 = time.time()
    cont, seg = codec.encode_v(flat, Hs, Ws, mode)
    codec.check(); torch.cuda.synchronize(); t1 = time.time()
    if mode == "auto" or (not isinstance(mode, int)) or (mode & 0x10000):
        mode = codec.container_modes(cont)           # per image, from the headers
        codec.workspace_v(Hs, Ws, mode)
    codec.poison_workspace()
    torch.cuda.synchronize(); t1b = time.time()
    rec = codec.decode_v(cont, seg, Hs, Ws, mode)
    codec.check(); torch.cuda.synchronize(); t2 = time.time()
    ok = bool(torch.equal(rec, flat))
    mp = sum(h * w for h, w in sh) / 1e6
    print(f"{name}: {len(sh)} images, {mp:.1f} MPix ok={ok} enc {mp/(t1-t0):.1f} MPix/s dec {mp/(t2-t1b):.1f} MPix/s workspace {codec._ws.numel()/2**30:.2f} GiB", flush=True)
    assert ok
    results.append({"case": name, "images": len(sh), "megapixels": round(mp, 2), "lossless": ok, "enc_mpix_s": round(mp / (t1 - t0), 1), "dec_mpix_s": round(mp / (t2 - t1b), 1),
                    "bpp": round(8.0 * float(seg.sum()) / (mp * 1e6), 4), "workspace_GiB": round(codec._ws.numel() / 2**30, 2)})
    del flat, cont, seg, rec
    codec._ws = None
    torch.cuda.empty_cache()
run_mixed(24, "auto", "24 images at the reference's eval-set sizes, mixed, container auto (a count per image)")
run_mixed(24, MODE_RANS(8, wide=2), "24 images at the reference's eval-set sizes, mixed (xrans8)")
run_mixed(24, MODE_RANS(8, wide=2), "same, warm")
run_mixed(256, MODE_RANS(1, wide=2), "256 images at the reference's eval-set sizes in ONE call (xrans1)")
run_mixed(6, MODE_RANS(14, wide=2), "six large odd sizes incl. 33x8160 and 8160x33 in one call (xrans14)", big=True)
run_mixed(6, MODE_RANS(32), "the same six in 32 narrow streams (rans32)", big=True)
if len(sys.argv) > 1:
    json.dump({"tool": "tools/stress.py", "note": "single runs incl. first-call plan set-up unless marked warm; decode on a poisoned workspace", "runs": results},
              open(sys.argv[1], "w"), indent=1)
