#!/usr/bin/env python3
"""N encode + decode steps of 24 x 768x512 images DRAWN FROM A 1.7-BIT MODEL (bench.py's model_drawn leg: one live mixture component of sigma 0.6
grey levels; the reference's trained model spends 1.68 bits per last-stage symbol) in a given container -- the workload to put under
`rocprofv3 --kernel-trace --stats` for the serial tail kernels' durations on cheap content.  Usage: python3 tools/cheap_step.py [container] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from llicti_amd.codec import MODE_AC, HipCodec, mode_of_name
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "xrans10"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, H, W = 24, 512, 768
dev = torch.device("cuda", 0)
sd = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_trainedlike.npz")))
for k in list(sd):
    if k.endswith("layers1toL.2.bias"):
        b = sd[k].copy(); b[0:15] = 0.6 / 255.0; b[30:45] = np.tile(np.array([1.0, 1e-7, 1e-7, 1e-7, 1e-7], np.float32), 3); sd[k] = b
    if k.endswith("layers1toL.2.weight"):
        w = sd[k].copy(); w[0:15] = 0.0; w[30:45] = 0.0; sd[k] = w
codec = HipCodec(dev); codec.load_state_dict(sd)
x0 = torch.from_numpy(bench.make_batch(B, H, W, 0)).to(dev)
cont, seg = codec.encode(x0, mode=MODE_AC); codec.check()
ch, sh = cont.cpu().numpy().copy(), seg.cpu().numpy()
rng = np.random.default_rng(1)
for b in range(B):
    h0, n = int(sh[b, :4].sum()), int(sh[b].sum())
    ch[b, h0:n] = rng.integers(0, 256, n - h0, dtype=np.uint8)
x = codec.decode(torch.from_numpy(ch).to(dev), seg, H, W, mode=MODE_AC).clone()
torch.cuda.synchronize()
mode = mode_of_name(name)
c, s = codec.encode(x, mode=mode); codec.check()
if mode & 0x10000:               # encoder mode "auto": from here on the container it wrote (the decoder needs the container's own mode)
    mode = sorted(set(codec.container_modes(c)))[0]
r = codec.decode(c, s, H, W, mode=mode); codec.check()
assert torch.equal(r, x)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    codec.encode(x, mode=mode, out=c, seg_len=s)
    codec.decode(c, s, H, W, mode=mode, out=r)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{name}: {steps} steps, {dt * 1e3:.3f} ms per encode + decode = {B * H * W / dt / 1e6:.1f} MPix/s, {8.0 * float(s.sum()) / (B * H * W):.4f} bpp")
