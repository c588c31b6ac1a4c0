"""Stream-count sweep of the xwide v4 container on the MI355X: encode + decode time, decode kernel groups and bytes against the reference-format
container, for the timed batch (24 x 768x512 noise), one 768x512 image, a natural-like batch and one 3840x2160 image.
python tools/sweep_v4.py [out.json]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from llicti_amd.codec import HipCodec, MODE_RANS, MODE_AC
from helpers import make_image

dev = torch.device("cuda:0")
out = {}


def sweep(name, sd, rgb, Ms, reps=5):
    codec = HipCodec(dev); codec.load_state_dict(sd)
    legs = bench.Legs(torch, codec, dev)
    B, _, H, W = rgb.shape
    r_ac = legs.run(rgb, MODE_AC, reps=1)
    rows = {}
    for M in Ms:
        mode = MODE_RANS(M, wide=2)
        r = legs.run(rgb, mode, reps=reps)
        codec.set_profiling(True)
        cont, seg = codec.encode(rgb, mode=mode); torch.cuda.synchronize(); ke, _ = codec.last_timing_detail()
        codec.decode(cont, seg, H, W, mode=mode); torch.cuda.synchronize(); kd, _ = codec.last_timing_detail()
        codec.set_profiling(False)
        r["bpp_delta_vs_ac_container"] = round(8.0 * (r["bytes"] - r_ac["bytes"]) / (B * H * W), 6)
        r["bytes_per_image_over_ac"] = round((r["bytes"] - r_ac["bytes"]) / B, 1)
        r["enc_kernel_ms"] = {k: round(v, 3) for k, v in ke.items() if v}
        r["dec_kernel_ms"] = {k: round(v, 3) for k, v in kd.items() if v}
        rows[f"xrans{M}"] = r
        print(name, f"xrans{M}", r["encdec_mpix_s"], r["bpp_delta_vs_ac_container"], r["dec_kernel_ms"], flush=True)
    out[name] = {"ac": r_ac, "modes": rows}
    legs.free(); codec.close()


sd_r = dict(np.load(os.path.join(ROOT, "tests/golden/weights_rand1337.npz")))
sd_t = dict(np.load(os.path.join(ROOT, "tests/golden/weights_trainedlike.npz")))
noise = torch.from_numpy(bench.make_batch(24, 512, 768, 0)).to(dev)
sweep("noise_24x768x512", sd_r, noise, (8, 10, 12, 14, 16, 18, 20, 21, 24, 28, 32))
sweep("noise_1x768x512", sd_r, noise[:1].contiguous(), (10, 16, 20, 24, 32, 64, 128), reps=10)
sweep("noise_32x768x512", sd_r, torch.from_numpy(bench.make_batch(32, 512, 768, 0)).to(dev), (8, 12, 16, 20, 24))
smooth = torch.from_numpy(np.stack([make_image("smooth", 512, 768, 11 + i) for i in range(24)])).to(dev)
sweep("smooth_24x768x512", sd_t, smooth, (10, 14, 16, 18, 20))
big = torch.from_numpy(bench.make_batch(1, 2160, 3840, 0)).to(dev)
sweep("noise_1x3840x2160", sd_r, big, (32, 64, 128), reps=3)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6_sweep_v4.json", "w"), indent=1)
