"""Quick GPU parity of the xwide v4 container against the oracle (encode bytes, decode of HIP bytes on a poisoned workspace), fixed and auto counts."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import oracle as orc
from llicti_amd.weights import pack_state_dict
from llicti_amd.codec import HipCodec, MODE_RANS, MODE_RANS_AUTO, container_to_bytestream_list, image_streams, mode_of_header, name_of_mode
from helpers import make_image, make_sampled_image
from test_oracle_golden import _cheap_case
gold = 'tests/golden/'
fails = 0
full = "--full" in sys.argv
for wname in ("rand1337", "trainedlike"):
    sd = dict(np.load(gold + f"weights_{wname}.npz"))
    W_o = orc.Weights(pack_state_dict(sd))
    codec = HipCodec("cuda:0"); codec.load_state_dict(sd)
    cases = [("noise", 96, 160), ("smooth", 96, 160), ("noise", 67, 93), ("smooth", 256, 384), ("noise", 32, 32), ("smooth", 512, 768), ("noise", 512, 768)]
    for kind, H, Wd in cases:
        img = make_image(kind, H, Wd, 3)
        dev = torch.from_numpy(img[None]).to("cuda:0")
        Ms = [(M, False) for M in ((1, 2, 3, 10, 16, 20, 32, 64, 128) if full else (3, 16))] + [(max(1, image_streams(H, Wd)), True), (5, True)]
        for M, auto in Ms:
            if H * Wd > 100000 and M in (2, 3, 64, 128) and not auto: continue
            mode = MODE_RANS_AUTO(M) if auto else MODE_RANS(M, wide=2)
            cont, seg = codec.encode(dev, mode=mode); codec.check()
            got = container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy())
            ref = orc.encode_image_rans(img, W_o, M, 2, auto=auto)
            same = got == ref
            dmode = codec.container_modes(cont)[0]
            codec.poison_workspace()
            rec = codec.decode(cont, seg, H, Wd, mode=dmode); codec.check()
            ok = bool(torch.equal(rec, dev))
            if not (same and ok):
                fails += 1
                print("FAIL", wname, kind, H, Wd, name_of_mode(mode), "->", name_of_mode(dmode), "bytes_equal", same, "lossless", ok, flush=True)
            else:
                print("ok", wname, kind, H, Wd, name_of_mode(mode), "->", name_of_mode(dmode), sum(len(x) for r in got for x in r), flush=True)
# cheap content: the auto rule halves the count where the last stage cannot fill the payloads
for kind in ("sharp", "single"):
    sd, W_c, img = _cheap_case(kind)
    codec = HipCodec("cuda:0"); codec.load_state_dict(sd)
    H, Wd = img.shape[1:]
    dev = torch.from_numpy(img[None]).to("cuda:0")
    for M in (4, 6, 9):
        mode = MODE_RANS_AUTO(M)
        cont, seg = codec.encode(dev, mode=mode); codec.check()
        got = container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy())
        ref = orc.encode_image_rans(img, W_c, M, 2, auto=True)
        dmode = codec.container_modes(cont)[0]
        codec.poison_workspace()
        rec = codec.decode(cont, seg, H, Wd, mode=dmode); codec.check()
        ok = got == ref and bool(torch.equal(rec, dev))
        fails += 0 if ok else 1
        print("ok" if ok else "FAIL", "cheap", kind, H, Wd, name_of_mode(mode), "->", name_of_mode(dmode), sum(len(x) for r in got for x in r), flush=True)
# a batch of mixed sizes with auto counts per image
sd = dict(np.load(gold + "weights_rand1337.npz")); W_o = orc.Weights(pack_state_dict(sd))
codec = HipCodec("cuda:0"); codec.load_state_dict(sd)
imgs = [make_image("noise", 256, 384, 1), make_image("smooth", 321, 481, 2), make_image("noise", 150, 131, 3), make_image("smooth", 512, 768, 4)]
Hs, Ws = [i.shape[1] for i in imgs], [i.shape[2] for i in imgs]
flat = torch.from_numpy(np.concatenate([i.reshape(-1) for i in imgs])).to("cuda:0")
modes = [MODE_RANS_AUTO(max(1, image_streams(h, w))) for h, w in zip(Hs, Ws)]
cont, seg = codec.encode_v(flat, Hs, Ws, modes); codec.check()
dm = codec.container_modes(cont)
for b, im in enumerate(imgs):
    got = container_to_bytestream_list(cont[b].cpu().numpy(), seg[b].cpu().numpy())
    ref = orc.encode_image_rans(im, W_o, modes[b] & 0xFF, 2, auto=True)
    ok = got == ref
    fails += 0 if ok else 1
    print("ok" if ok else "FAIL", "mixed", b, Hs[b], Ws[b], name_of_mode(modes[b]), "->", name_of_mode(dm[b]), flush=True)
codec.poison_workspace()
rec = codec.decode_v(cont, seg, Hs, Ws, dm); codec.check()
okm = bool(torch.equal(rec, flat)); fails += 0 if okm else 1
print("mixed decode lossless", okm)
print("FAILS", fails)
sys.exit(1 if fails else 0)
