#!/usr/bin/env python3
"""Timing of the rANS tail kernel for several builds of the library on the same containers (GPU box): pricing experiments.
  python tools/probe_tail_variants.py gen DIR            -- 24 x 768x512 images drawn from the "single" model of tools/probe_cheap_content.py
                                                             (1.7 bits per last-stage symbol) and a noise batch, encoded in xrans10 / rans10 with the
                                                             default build; containers to DIR
  LLICTI_HIP_SO=... python tools/probe_tail_variants.py run DIR NAME   -- decode them with that build, print the kernel groups
Builds: hipcc -D switches of rans_tail_kernel (-DTAIL_SPEC=0: no speculated window; -DTAIL_AHEAD=n: preparing wavefronts per chain).
The timing-only switches of profiles/r5/tail_speculation.json (coder alone / preparing wavefronts alone / forced hits: wrong pixels by construction,
"lossless": false in their rows) were removed from the kernel again after the measurement: history, commit a3bba44's parent series."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from llicti_amd.codec import MODE_AC, HipCodec, mode_of_name
B, H, W = 24, 512, 768
dev = torch.device("cuda", 0)
MODES = ("xrans10", "rans10")


def weights(kind):
    if kind == "noise":
        return dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_rand1337.npz")))
    src = open(os.path.join(ROOT, "tools", "probe_cheap_content.py")).read().split("out = {")[0]
    ns = {"__file__": os.path.join(ROOT, "tools", "probe_cheap_content.py")}
    exec(compile(src, "probe_cheap_content_head", "exec"), ns)
    return ns["cheap_sd"](kind)


if sys.argv[1] == "gen":
    d = sys.argv[2]; os.makedirs(d, exist_ok=True)
    for kind in ("single", "noise"):
        codec = HipCodec(dev); codec.load_state_dict(weights(kind))
        x = torch.from_numpy(np.stack([np.random.default_rng(i).integers(0, 256, (3, H, W), dtype=np.uint8) for i in range(B)])).to(dev)
        if kind != "noise":
            cont, seg = codec.encode(x, mode=MODE_AC); codec.check()
            ch, sh = cont.cpu().numpy().copy(), seg.cpu().numpy()
            rng = np.random.default_rng(1)
            for b in range(B):
                h0, n = int(sh[b, :4].sum()), int(sh[b].sum())
                ch[b, h0:n] = rng.integers(0, 256, n - h0, dtype=np.uint8)
            x = codec.decode(torch.from_numpy(ch).to(dev), seg, H, W, mode=MODE_AC)
        np.save(f"{d}/{kind}_x.npy", x.cpu().numpy())
        for name in MODES:
            c, s = codec.encode(x, mode=mode_of_name(name)); codec.check()
            n = int(s.sum(dim=1).max().item())
            np.save(f"{d}/{kind}_{name}_c.npy", c[:, :n + 64].cpu().numpy()); np.save(f"{d}/{kind}_{name}_s.npy", s.cpu().numpy())
        codec.close()
else:
    d, vname = sys.argv[2], sys.argv[3]
    for kind in ("single", "noise"):
        codec = HipCodec(dev); codec.load_state_dict(weights(kind))
        x = torch.from_numpy(np.load(f"{d}/{kind}_x.npy")).to(dev)
        for name in MODES:
            mode = mode_of_name(name)
            stride = codec.max_container_bytes(H, W)
            c0 = np.load(f"{d}/{kind}_{name}_c.npy")
            c = torch.zeros((B, stride), dtype=torch.uint8, device=dev); c[:, :c0.shape[1]] = torch.from_numpy(c0).to(dev)
            s = torch.from_numpy(np.load(f"{d}/{kind}_{name}_s.npy")).to(dev)
            r = codec.decode(c, s, H, W, mode=mode)
            try:
                codec.check(); ok = bool(torch.equal(r, x))
            except Exception:
                ok = False
            codec.set_profiling(True)
            ms = []
            for _ in range(3):
                codec.decode(c, s, H, W, mode=mode, out=r); torch.cuda.synchronize()
                cat, _ = codec.last_timing_detail(); ms.append(cat["rans_tail"])
            codec.set_profiling(False)
            print(json.dumps({"build": vname, "content": kind, "container": name, "lossless": ok, "rans_tail_ms": round(sorted(ms)[1], 3), "rans_stage_ms": round(cat["rans_stage"], 3)}), flush=True)
        codec.close()
