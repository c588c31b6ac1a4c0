#!/usr/bin/env python3
"""Debug aid (GPU box; test infrastructure -- it calls the CPU oracle, hence under tests/; not collected by pytest): HIP vs oracle
rANS container, stream by stream; cross-decodes (HIP decodes the oracle's bytes, the
oracle decodes HIP's)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from llicti_amd.codec import HipCodec, MODE_RANS, container_to_bytestream_list, bytestream_list_to_container
from llicti_amd.weights import pack_state_dict
from oracle import oracle as orc

H, W, M, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), (sys.argv[4] if len(sys.argv) > 4 else "noise")
wname = sys.argv[5] if len(sys.argv) > 5 else "trainedlike"
wide = len(sys.argv) > 6 and sys.argv[6] == "wide"
sd = {k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", f"weights_{wname}.npz")).items()}
c = HipCodec("cuda:0"); c.load_state_dict(sd)
Wo = orc.Weights(pack_state_dict(sd))
from helpers import make_batch
rgb = make_batch(kind, 2, H, W, seed0=70)
x = torch.from_numpy(rgb).cuda()
cont, seg = c.encode(x, mode=MODE_RANS(M, wide)); c.check()
ch, sh = cont.cpu().numpy(), seg.cpu().numpy()
for b in range(2):
    bl = container_to_bytestream_list(ch[b], sh[b])
    ref = orc.encode_image_rans(rgb[b], Wo, M, wide)
    print("image", b, "header equal", bl[0] == ref[0])
    for m in range(M):
        a, r = bl[1 + m // 9][m % 9], ref[1 + m // 9][m % 9]
        fd = next((i for i in range(min(len(a), len(r))) if a[i] != r[i]), None)
        print(f"  stream {m}: hip {len(a)} B T={a[0] | a[1] << 8}  oracle {len(r)} B T={r[0] | r[1] << 8}  first diff at {fd}")
    try:
        ok = np.array_equal(orc.decode_image_rans(bl, Wo), rgb[b]); print("  oracle decodes HIP bytes:", ok)
    except Exception as e:
        print("  oracle decodes HIP bytes: FAIL", e)
# HIP decodes oracle bytes
stride = cont.shape[1]
cont2 = torch.zeros_like(cont); seg2 = torch.zeros_like(seg)
for b in range(2):
    buf, sl = bytestream_list_to_container(orc.encode_image_rans(rgb[b], Wo, M, wide))
    cont2[b, :len(buf)] = torch.from_numpy(buf).cuda(); seg2[b] = torch.from_numpy(sl).cuda()
rec = c.decode(cont2, seg2, H, W, mode=MODE_RANS(M, wide))
try:
    c.check(); print("HIP decodes oracle bytes: status ok, equal =", bool((rec.cpu().numpy() == rgb).all()))
except Exception as e:
    print("HIP decodes oracle bytes: status", e, "equal =", bool((rec.cpu().numpy() == rgb).all()))
    d = (rec.cpu().numpy() != rgb)
    print("   wrong pixels per image/plane:", d.reshape(2, 3, -1).sum(-1).tolist())
