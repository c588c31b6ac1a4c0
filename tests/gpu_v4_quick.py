"""Quick GPU parity of the xwide v4 container against the oracle (encode bytes, decode of HIP and oracle bytes, poisoned workspace)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import oracle as orc
from llicti_amd.weights import pack_state_dict
from llicti_amd.codec import HipCodec, MODE_RANS, container_to_bytestream_list
from helpers import make_image, make_sampled_image
gold = 'tests/golden/'
fails = 0
for wname in ("rand1337", "trainedlike"):
    sd = dict(np.load(gold + f"weights_{wname}.npz"))
    W_o = orc.Weights(pack_state_dict(sd))
    codec = HipCodec("cuda:0"); codec.load_state_dict(sd)
    cases = [("noise", 96, 160), ("smooth", 96, 160), ("noise", 67, 93), ("smooth", 256, 384), ("noise", 32, 32), ("smooth", 512, 768), ("noise", 512, 768)]
    for kind, H, Wd in cases:
        img = make_image(kind, H, Wd, 3)
        dev = torch.from_numpy(img[None]).to("cuda:0")
        for M in (1, 2, 3, 10, 16, 20, 32, 64, 128):
            if H * Wd > 100000 and M in (2, 3, 64, 128): continue
            mode = MODE_RANS(M, wide=2)
            cont, seg = codec.encode(dev, mode=mode); codec.check()
            got = container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy())
            ref = orc.encode_image_rans(img, W_o, M, 2)
            same = got == ref
            codec.poison_workspace()
            rec = codec.decode(cont, seg, H, Wd, mode=mode); codec.check()
            ok = bool(torch.equal(rec, dev))
            if not (same and ok):
                fails += 1
                nd = [(i, j, len(a), len(b)) for i, (ra, rb) in enumerate(zip(got, ref)) for j, (a, b) in enumerate(zip(ra, rb)) if a != b][:4]
                print("FAIL", wname, kind, H, Wd, "M", M, "bytes_equal", same, "lossless", ok, nd, flush=True)
            else:
                print("ok", wname, kind, H, Wd, "M", M, sum(len(x) for r in got for x in r), flush=True)
print("FAILS", fails)
sys.exit(1 if fails else 0)
