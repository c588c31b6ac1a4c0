#!/usr/bin/env python3
"""Where does a mixed-size batch lose against an equal-size one?  (i) the bench batch (24 x 768x512, xrans10) through the equal-size code path and
through the mixed-size one (`force_ragged`: per-image tables, tile lists, the RAGGED CNN instantiation) -- same work, so the difference is the code
path; (ii) 24 images at the reference's eval-set sizes (mixed) against 24 images of their MEAN size -- the difference on top is load imbalance
(a launch waits for its largest image); (iii) the same mixed batches with a stream count per image in proportion to its pixels.  Usage: python tools/ab_ragged.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from llicti_amd.codec import HipCodec, mode_of_name  # noqa: E402
from llicti_amd.config import default_config  # noqa: E402
from llicti_amd.graphs.models.LLICTI_nets import LLICTI  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1337)
codec = HipCodec(dev)
codec.load_state_dict(LLICTI(default_config()).state_dict())
mode = mode_of_name("xrans10")


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run_uniform(B, H, W, ragged):
    codec.set_tuning("force_ragged", int(ragged))
    x = torch.from_numpy(np.stack([np.random.default_rng(i).integers(0, 256, (3, H, W), dtype=np.uint8) for i in range(B)])).to(dev)
    cont, seg = codec.encode(x, mode=mode)
    rec = codec.decode(cont, seg, H, W, mode=mode)
    codec.check()
    assert torch.equal(rec, x)
    te = timed(lambda: codec.encode(x, mode=mode, out=cont, seg_len=seg))
    td = timed(lambda: codec.decode(cont, seg, H, W, mode=mode, out=rec))
    codec.set_profiling(True)
    codec.encode(x, mode=mode, out=cont, seg_len=seg); torch.cuda.synchronize(); ce, _ = codec.last_timing_detail()
    codec.decode(cont, seg, H, W, mode=mode, out=rec); torch.cuda.synchronize(); cd, _ = codec.last_timing_detail()
    codec.set_profiling(False)
    codec.set_tuning("force_ragged", 0)
    mp = B * H * W / 1e6
    return {"enc_ms": round(te, 3), "dec_ms": round(td, 3), "encdec_mpix_s": round(mp / (te + td) * 1e3, 1),
            "enc_kernel_ms": {k: round(v, 3) for k, v in ce.items() if v > 0}, "dec_kernel_ms": {k: round(v, 3) for k, v in cd.items() if v > 0}}


def run_mixed(sh, mode=mode):
    Hs, Ws = [h for h, _ in sh], [w for _, w in sh]
    flat = torch.from_numpy(np.concatenate([np.random.default_rng(i).integers(0, 256, 3 * h * w, dtype=np.uint8) for i, (h, w) in enumerate(sh)])).to(dev)
    cont, seg = codec.encode_v(flat, Hs, Ws, mode)
    dmode = mode
    if (not isinstance(mode, int)) or (mode & 0x10000):
        dmode = codec.container_modes(cont)          # encoder modes "auto": the decoder takes every image's own mode from its header
    rec = codec.decode_v(cont, seg, Hs, Ws, dmode)
    codec.check()
    assert torch.equal(rec, flat)
    te = timed(lambda: codec.encode_v(flat, Hs, Ws, mode, out=cont, seg_len=seg))
    td = timed(lambda: codec.decode_v(cont, seg, Hs, Ws, dmode, out=rec))
    codec.set_profiling(True)
    codec.encode_v(flat, Hs, Ws, mode, out=cont, seg_len=seg); torch.cuda.synchronize(); ce, _ = codec.last_timing_detail()
    codec.decode_v(cont, seg, Hs, Ws, dmode, out=rec); torch.cuda.synchronize(); cd, _ = codec.last_timing_detail()
    codec.set_profiling(False)
    mp = sum(h * w for h, w in sh) / 1e6
    return {"megapixels": round(mp, 2), "enc_ms": round(te, 3), "dec_ms": round(td, 3), "encdec_mpix_s": round(mp / (te + td) * 1e3, 1),
            "enc_kernel_ms": {k: round(v, 3) for k, v in ce.items() if v > 0}, "dec_kernel_ms": {k: round(v, 3) for k, v in cd.items() if v > 0}}


out = {"container": "xrans10"}
out["bench_batch_equal_size_path"] = run_uniform(24, 512, 768, False)
out["bench_batch_mixed_size_path"] = run_uniform(24, 512, 768, True)
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "eval_shapes.json")))["shapes"]
for k0 in (100, 300):
    sh = shapes[k0:k0 + 24]
    out[f"eval_set_images_{k0}_{k0 + 24}_mixed"] = run_mixed(sh)
    out[f"eval_set_images_{k0}_{k0 + 24}_mixed"]["sizes"] = sorted(set(map(tuple, sh)))
    sh_sorted = sorted(sh, key=lambda s: s[0] * s[1])
    out[f"eval_set_images_{k0}_{k0 + 24}_sorted_by_size"] = run_mixed(sh_sorted)
    # (iii) a stream count per image, from its own size (llicti_encode_images_vm; round 5: balanced_modes, in proportion to the pixels)
    from llicti_amd.codec import auto_modes, name_of_mode
    bm = auto_modes([tuple(s) for s in sh])
    out[f"eval_set_images_{k0}_{k0 + 24}_stream_count_per_image"] = run_mixed(sh, bm)
    out[f"eval_set_images_{k0}_{k0 + 24}_stream_count_per_image"]["encoder_modes"] = sorted(set(name_of_mode(m) for m in bm))
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
