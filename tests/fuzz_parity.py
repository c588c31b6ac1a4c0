#!/usr/bin/env python3
"""Randomised HIP-vs-oracle parity hunt (GPU box): random shapes, image kinds, weight perturbations, containers.
Test infrastructure (it calls the CPU oracle), hence under tests/; not collected by pytest (minutes of GPU time).
usage: tests/fuzz_parity.py [N_CASES] [SEED] [SUMMARY.json] [xwide] [corrupt]   -- stops at the first mismatch with a reproducer line ("xwide": only the
256-lane containers, whose tail -- two seeded chains, radix-A seeds -- has the most cases: narrow value ranges, streams shorter than the seeds);
the summary (cases, per-container / per-kind counts, wall time) is what profiles/<round>/fuzz_summary.json holds."""
import collections, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from llicti_amd.codec import HipCodec, MODE_AC, MODE_RANS, MODE_RANS_AUTO, container_to_bytestream_list
from llicti_amd.weights import pack_state_dict
from oracle import oracle as orc
from helpers import make_image

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
XWIDE_ONLY = len(sys.argv) > 4 and "xwide" in sys.argv[4:]
CORRUPT_ALL = len(sys.argv) > 4 and "corrupt" in sys.argv[4:]      # every rANS case also decodes a corrupted copy (default: a third of them)
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
base = {w: dict(np.load(os.path.join(gold, f"weights_{w}.npz"))) for w in ("rand1337", "trainedlike")}
t0 = time.time()
counts = collections.Counter()
pixels = 0
for case in range(N):
    wname = ("rand1337", "trainedlike")[int(rng.integers(0, 2))]
    sd = {k: np.array(v) for k, v in base[wname].items()}
    scale = float(rng.choice([1.0, 1.0, 0.3, 3.0, 12.0, 60.0]))
    if scale != 1.0:
        for k in sd:
            if "layers1toL.2" in k:
                sd[k] = (sd[k] * scale).astype(np.float32)
    H, W = int(rng.integers(32, 161)), int(rng.integers(32, 201))
    B = int(rng.integers(1, 4))
    kind = ("noise", "smooth", "flat", "binary", "narrow")[int(rng.integers(0, 5))]
    big = kind in ("flat", "binary", "narrow") and rng.integers(0, 4) == 0      # round 5: cheap content, large enough for xwide tails beyond 2,047 / 4,095 symbols
    if big:
        H, W = int(rng.integers(200, 361)), int(rng.integers(250, 421))
    mixed = bool(rng.integers(0, 2))               # round 5: the images of the call differ in size (llicti_encode_images_v; rANS containers)
    sizes = [(H, W)] + [((int(rng.integers(32, 161)), int(rng.integers(32, 201))) if mixed else (H, W)) for _ in range(B - 1)]
    imgs = []
    for b in range(B):
        H, W = sizes[b]
        seed = int(rng.integers(0, 2 ** 31))
        if kind in ("noise", "smooth"):
            imgs.append(make_image(kind, H, W, seed))
        elif kind == "narrow":                     # few pixel values: a small Cg alphabet A, i.e. many seed symbols per chain (xwide tail)
            r = np.random.default_rng(seed)
            imgs.append((r.integers(0, int(r.choice([2, 3, 5, 9, 17])), (3, H, W)) + int(r.integers(0, 200))).astype(np.uint8))
        elif kind == "flat":
            imgs.append(np.broadcast_to(np.random.default_rng(seed).integers(0, 256, (3, 1, 1), dtype=np.uint8), (3, H, W)).copy())
        else:
            imgs.append(np.random.default_rng(seed).choice(np.array([0, 255], np.uint8), size=(3, H, W)))
    H, W = sizes[0]
    M = int(rng.choice([0, 1, 2, 3, 4, 8, 10, 11, 16, 32, 64, 128, -1, -5, -10, -14, -1001, -1003, -1009, -1015, -1021, -1032, -1064, -1128]))     # negative: |M| wide streams, |M| - 1000 xwide streams
    if XWIDE_ONLY or big: M = int(rng.choice([-1001, -1002, -1003, -1005, -1009, -1010, -1014, -1016, -1020, -1021, -1027, -1032, -1064, -1128] if not big else [-1001, -1002, -1003]))
    wide = 0 if M >= 0 else (2 if M <= -1000 else 1)
    M = abs(M) % 1000
    mode = MODE_AC if M == 0 else MODE_RANS(M, wide)
    if M == 0 and len(set(sizes)) > 1:             # the reference-format container codes one size per call
        sizes = [sizes[0]] * B
        imgs = [imgs[0]] + [np.random.default_rng(int(rng.integers(0, 2 ** 31))).integers(0, 256, (3,) + sizes[0], dtype=np.uint8) for _ in range(B - 1)]
    ragged = len(set(sizes)) > 1
    tag = f"case {case}: {wname} x{scale} {kind}{' big' if big else ''} B={B} {'x'.join(f'{w}x{h}' for h, w in sizes) if ragged else f'{W}x{H}'} M={M}{('', ' wide', ' xwide')[wide]}"
    codec = HipCodec("cuda:0")
    codec.load_state_dict(sd)
    W_o = orc.Weights(pack_state_dict(sd))
    Hs, Ws = [h for h, _ in sizes], [w for _, w in sizes]
    Mb = [M] * B                                    # streams of image b (round 5: a count per image in one call, llicti_encode_images_vm)
    if M != 0 and B > 1 and rng.integers(0, 2) == 0:
        pool = [1, 2, 3, 4, 8, 10, 11, 16, 32] if wide == 0 else [1, 2, 3, 5, 9, 10, 14] if wide == 1 else [1, 2, 3, 5, 9, 10, 14, 16, 20, 21, 32]
        Mb = [M] + [int(rng.choice(pool)) for _ in range(B - 1)]
    # round 6: the xwide "auto" ENCODER mode now and then -- the encoder picks each image's count on the device from what its last stage costs; the
    # decoder takes the count from the container's header; the oracle applies the same rule (orc_auto_streams)
    auto = wide == 2 and all(m <= 32 for m in Mb) and rng.integers(0, 3) == 0
    modes = [(MODE_RANS_AUTO(m) if auto else MODE_RANS(m, wide)) for m in Mb] if M != 0 else mode
    if auto:
        mode = modes[0]
        tag += " auto"
    per_image = M != 0 and (len(set(Mb)) > 1 or (auto and B > 1))
    if ragged or per_image or (M != 0 and rng.integers(0, 4) == 0):          # (equal sizes through the _v entry now and then)
        flat = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs])).cuda()
        cont, seg = codec.encode_v(flat, Hs, Ws, modes)
        codec.check()
        if auto:
            modes = codec.container_modes(cont)    # what the encoder picked, from the headers
            mode = modes[0]
        codec.poison_workspace()                   # the decode must not find the encoder's planes in the workspace
        rec = codec.decode_v(cont, seg, Hs, Ws, modes)
        codec.check()
        assert torch.equal(rec, flat), "ROUND TRIP " + tag
    else:
        x = torch.from_numpy(np.stack(imgs)).cuda()
        cont, seg = codec.encode(x, mode=mode)
        codec.check()
        if auto:                                   # (B == 1 here: one mode per call)
            mode = codec.container_modes(cont)[0]
            modes = [mode]
        codec.workspace(B, H, W, mode)
        codec.poison_workspace()
        rec = codec.decode(cont, seg, H, W, mode=mode)
        codec.check()
        assert torch.equal(rec, x), "ROUND TRIP " + tag
    ch, sh = cont.cpu().numpy(), seg.cpu().numpy()
    for b in range(B):
        ref = orc.encode_image(imgs[b], W_o) if M == 0 else orc.encode_image_rans(imgs[b], W_o, Mb[b], wide, auto=auto)
        assert container_to_bytestream_list(ch[b], sh[b]) == ref, "BYTES " + tag + f" image {b}"
    if M != 0 and (CORRUPT_ALL or rng.integers(0, 3) == 0):
        # round 5, corrupted containers: bytes of ONE image's streams flipped -> the HIP decoder and the oracle agree on whether the image is malformed
        # (status LLICTI_EFORMAT <=> the oracle's decoder refuses it) and, where neither notices (a flip the integrity checks cannot see), on every pixel;
        # the other images of the call decode to their originals; nothing hangs or faults.  (The reference-format container has no integrity check.)
        from llicti_amd._lib import LlictiError
        bsel = int(rng.integers(0, B))
        hdr, used = int(sh[bsel][:4].sum()), int(sh[bsel].sum())
        bad = cont.clone()
        for p_ in rng.integers(hdr, used, int(rng.choice([1, 1, 2, 8]))):
            bad[bsel, int(p_)] ^= int(rng.integers(1, 256))
        codec.poison_workspace(0x3C)
        if ragged or per_image:
            recb = codec.decode_v(bad, seg, Hs, Ws, modes)
            offs = HipCodec.flat_offsets(Hs, Ws)[0]
            got = [recb[int(offs[b]):int(offs[b]) + 3 * Hs[b] * Ws[b]].view(3, Hs[b], Ws[b]).cpu().numpy() for b in range(B)]
        else:
            codec.workspace(B, H, W, mode)
            recb = codec.decode(bad, seg, H, W, mode=mode)
            got = [recb[b].cpu().numpy() for b in range(B)]
        try:
            codec.check()
            flagged = False
        except LlictiError:
            flagged = True
        st = codec.image_status(B)
        assert flagged == bool(st.any()), "STATUS " + tag
        try:
            want = orc.decode_image_rans(container_to_bytestream_list(bad[bsel].cpu().numpy(), sh[bsel]), W_o)
        except RuntimeError:
            want = None
        assert (st[bsel] != 0) == (want is None), "CORRUPT VERDICT " + tag + f" image {bsel}: hip status {st[bsel]}, oracle {'refuses' if want is None else 'accepts'}"
        if want is not None:
            assert np.array_equal(got[bsel], want), "CORRUPT PIXELS " + tag + f" image {bsel}"
            counts["corrupt_undetected_same_pixels"] += 1
        else:
            counts["corrupt_detected_by_both"] += 1
        for b in range(B):
            if b != bsel:
                assert st[b] == 0 and np.array_equal(got[b], imgs[b]), "CORRUPT NEIGHBOUR " + tag + f" image {b}"
        counts["corrupted_containers"] += 1
    if ragged: counts["mixed_size_calls"] += 1
    if per_image: counts["stream_count_per_image_calls"] += 1
    if big: counts["big_cheap_images"] += 1
    codec.close()
    counts[f"container:{'ac' if M == 0 else ('rans%d', 'wrans%d', 'xrans%d')[wide] % M}"] += 1
    counts[f"kind:{kind}"] += 1
    counts[f"weights:{wname}x{scale}"] += 1
    pixels += sum(h * w for h, w in sizes)
    if case % 10 == 9:
        print(f"{case + 1} cases ok ({time.time() - t0:.0f} s); last: {tag}", flush=True)
print("fuzz ok:", N, "cases")
if len(sys.argv) > 3:
    json.dump({"tool": "tests/fuzz_parity.py", "cases": N, "seed": int(sys.argv[2]), "mismatches": 0, "pixels": pixels,
               "wall_s": round(time.time() - t0, 1), "containers": "xwide only" if XWIDE_ONLY else "all", "checked": "decode(encode(x)) == x on a poisoned workspace; every image's container "
               "byte-identical to the CPU oracle's; corrupted rANS containers (byte flips in one image's streams): HIP status == the oracle's verdict, equal pixels where neither notices, neighbours intact", "counts": dict(sorted(counts.items()))}, open(sys.argv[3], "w"), indent=1)
