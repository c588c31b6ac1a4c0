"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the
CPU oracle on the same seeded inputs and against the golden fixtures.  Integer / byte / index work and --
because the numerics spec fixes every fp32 rounding -- the CNN outputs, the 16-bit tables and the
bitstreams are required to be BIT-EXACT."""
import numpy as np
import pytest

from conftest import CASES, load_case, load_state_dict
from helpers import make_batch, make_image

pytestmark = pytest.mark.gpu

PARAM_TOL = 1e-5      # vs the reference's PyTorch numbers (north_star); vs the oracle the bar is equality


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def codecs(torch_mod):
    from llicti_amd.codec import HipCodec
    cache = {}

    def get(wname):
        if wname not in cache:
            c = HipCodec("cuda:0")
            c.load_state_dict(load_state_dict(wname))
            cache[wname] = c
        return cache[wname]
    yield get
    for c in cache.values():
        c.close()


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("shape", [(32, 32), (67, 93), (128, 80), (33, 250)])
def test_lift_unlift_exact(torch_mod, codecs, shape):
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs("rand1337")
    H, W = shape
    rgb = make_batch("noise", 3, H, W, seed0=10)
    planes, fplanes, mm = c.lift(_dev(torch, rgb))
    for b in range(3):
        p_ref, mm_ref = orc.lift(rgb[b])
        assert np.array_equal(planes[b].cpu().numpy(), p_ref)
        assert np.array_equal(mm[b].cpu().numpy(), [mm_ref[1], mm_ref[2], mm_ref[4], mm_ref[5]])
        assert np.array_equal(fplanes[b].cpu().numpy(), p_ref.astype(np.float32) / np.float32(255))
    assert np.array_equal(c.unlift(planes).cpu().numpy(), rgb)


@pytest.mark.parametrize("case", CASES)
def test_band_params_bitexact_and_golden(torch_mod, codecs, golden_index, oracle_weights, case):
    """MFMA CNN == oracle's k-ordered fmaf chains bit for bit, and within 1e-5 of the reference's PyTorch."""
    from oracle import oracle as orc
    torch = torch_mod
    info = golden_index[case]
    c = codecs(info["weights"])
    W_o = oracle_weights(info["weights"])
    g = load_case(case)
    rgb = g["rgb"]
    planes, fplanes, mm = c.lift(_dev(torch, rgb[None]))
    p_host = planes[0].cpu().numpy()
    try:
        for rows in (16, 8, 4, -1, 0):             # the three tile forms of the kernel, round 3's choice rule, then the automatic choice
            c.set_tuning("cnn_tile_rows", rows)
            for lvl in range(5):
                for band in range(3):
                    got = np.ascontiguousarray(c.params60(c.band_params(fplanes, lvl, band))[0].cpu().numpy())
                    ref = orc.band_params(p_host, lvl, band, W_o)
                    assert got.shape == ref.shape
                    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (rows, lvl, band, np.abs(got - ref).max())
                    key = f"params_s{lvl}_b{band}"
                    if key in g.files:
                        assert np.abs(got - np.transpose(g[key], (1, 2, 0))).max() < PARAM_TOL
    finally:
        c.set_tuning("cnn_tile_rows", 0)


@pytest.mark.parametrize("case", ["smooth_67x93_tl", "noise_33x64_tl", "noise_32x32_rand"])
def test_cdf_tables_and_pairs_bitexact(torch_mod, codecs, golden_index, oracle_weights, case):
    from oracle import oracle as orc
    torch = torch_mod
    info = golden_index[case]
    c = codecs(info["weights"])
    W_o = oracle_weights(info["weights"])
    rgb = load_case(case)["rgb"]
    H, W = rgb.shape[1:]
    planes, fplanes, mm = c.lift(_dev(torch, rgb[None]))
    p_host, mm_ref = orc.lift(rgb)
    offs = {0: (1, 1), 1: (0, 1), 2: (1, 0)}
    for lvl in range(5):
        for band in range(3):
            params = c.band_params(fplanes, lvl, band)
            par_ref = orc.band_params(p_host, lvl, band, W_o)
            pairs = c.cdf_pairs(planes, params, mm, lvl, band).cpu().numpy().view(np.uint32)
            for clr in range(3):
                clow, chigh, sym = orc.stream_pairs(p_host, mm_ref, lvl, band, clr, par_ref)
                got = pairs[clr, 0]
                assert np.array_equal(got & 0xFFFF, clow), (lvl, band, clr)
                assert np.array_equal(got >> 16, chigh & 0xFFFF), (lvl, band, clr)
                minv = -127 if clr == 0 else int(mm_ref[clr])
                maxv = 128 if clr == 0 else int(mm_ref[3 + clr])
                Lp = maxv - minv + 2
                stride = 264 if clr == 0 else 512
                tab = c.cdf_tables(planes, params, mm, lvl, band, clr, row_stride=stride)[0].cpu().numpy().view(np.uint16)
                *_, hc, wc = __import__("llicti_amd._lib", fromlist=["level_geom"]).level_geom(H, W, lvl, band)
                assert tab.shape == (hc * wc, stride)
                oi, oj = offs[band]
                idx = np.unique(np.linspace(0, hc * wc - 1, 12).astype(int))
                for n in idx:
                    i, j = divmod(int(n), wc)
                    R, Cc = (2 * i + oi) << lvl, (2 * j + oj) << lvl
                    row = orc.cdf_row(par_ref[i, j], clr, np.float32(p_host[0, R, Cc]) / np.float32(255),
                                      np.float32(p_host[1, R, Cc]) / np.float32(255), minv, maxv)
                    assert np.array_equal(tab[n, :Lp], row), (lvl, band, clr, n)
                    assert (tab[n, Lp:] == 0xFFFF).all()


def test_cdf_table_kernel_adversarial_params(torch_mod, codecs):
    """Every row of the full-table kernel against the oracle for hand-made CNN outputs: very narrow and very
    wide components, means far outside / on the edges of the range, saturating weights -- the cases the
    kernel's constant-fill intervals (entries it does not evaluate) must get exactly right."""
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs("rand1337")
    H = W = 32
    rng = np.random.default_rng(7)
    rgb = make_batch("noise", 1, H, W, seed0=3)
    planes, fplanes, mm = c.lift(_dev(torch, rgb))
    p_host, _ = orc.lift(rgb[0])
    h = w = 16
    n = h * w
    par60 = np.zeros((n, 60), np.float32)
    kinds = rng.integers(0, 6, size=n)
    for r in range(n):
        k = kinds[r]
        sig = {0: rng.uniform(1e-5, 1e-3, 15), 1: rng.uniform(0.003, 0.05, 15), 2: rng.uniform(0.2, 3.0, 15),
               3: rng.uniform(-1, 1, 15), 4: 10.0 ** rng.uniform(-6, 3, 15), 5: rng.uniform(0.0004, 0.01, 15)}[k]
        mu = {0: rng.uniform(-1.0, 1.0, 15), 1: rng.uniform(-1.2, 1.2, 15), 2: rng.uniform(-0.5, 0.5, 15),
              3: rng.uniform(-5, 5, 15), 4: rng.choice([-1.0, -0.5, 0.0, 0.5, 1.0, 127.5 / 255, -127.5 / 255], 15),
              5: (rng.integers(-255, 256, 15) + rng.choice([0.0, 0.5, -0.5], 15)) / 255.0}[k]
        wt = {0: rng.uniform(0, 1, 15), 1: rng.uniform(-0.5, 1, 15), 2: 10.0 ** rng.uniform(-8, 2, 15),
              3: rng.uniform(0, 1, 15), 4: rng.uniform(0, 1, 15), 5: rng.uniform(0, 1, 15)}[k]
        par60[r, 0:15], par60[r, 15:30], par60[r, 30:45] = sig, mu, wt
        par60[r, 45:60] = rng.uniform(-1, 1, 15)
    par64 = np.zeros((1, 4, 16, n), np.float32)                    # channel-planar device layout: [B][64 planes][h * w]
    par64[0, :, :15, :] = par60.reshape(n, 4, 15).transpose(1, 2, 0)
    params = _dev(torch, par64.reshape(1, 64, h, w))
    for (mnco, mncg, mxco, mxcg) in ((-255, -255, 255, 255), (-3, -40, 5, 61), (0, -1, 0, 0)):
        mm2 = _dev(torch, np.array([[mnco, mncg, mxco, mxcg]], np.int32))
        for clr in range(3):
            minv = -127 if clr == 0 else (mnco, mncg)[clr - 1]
            maxv = 128 if clr == 0 else (mxco, mxcg)[clr - 1]
            Lp = maxv - minv + 2
            for stride in ((264,) if clr == 0 else (512, max(8, (Lp + 7) // 8 * 8))):
                tab = c.cdf_tables(planes, params, mm2, 0, 0, clr, row_stride=stride)[0].cpu().numpy().view(np.uint16)
                for r in range(n):
                    i, j = divmod(r, w)
                    R, Cc = 2 * i + 1, 2 * j + 1
                    row = orc.cdf_row(par60[r], clr, np.float32(p_host[0, R, Cc]) / np.float32(255),
                                      np.float32(p_host[1, R, Cc]) / np.float32(255), minv, maxv)
                    assert np.array_equal(tab[r, :Lp], row), (clr, r, int(kinds[r]), stride)
                    assert (tab[r, Lp:] == 0xFFFF).all()


def test_ac_seam_matches_oracle(torch_mod, codecs):
    """torchac seam: explicit tables + symbols -> bytes equal to the oracle coder; decode inverts."""
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs("rand1337")
    rng = np.random.default_rng(5)
    for Lp, N, S in [(257, 700, 3), (12, 1000, 2), (512, 300, 4), (2, 50, 1), (3, 64, 2)]:
        stride = (Lp + 7) // 8 * 8
        cdfs = np.full((S, N, stride), 0xFFFF, np.uint16)
        syms = np.zeros((S, N), np.int16)
        for s in range(S):
            # random strictly increasing rows in the reference's format: q + arange, last entry wraps to 0
            pm = rng.random((N, Lp - 1)) ** 4 + 1e-4
            cum = np.concatenate([np.zeros((N, 1)), np.cumsum(pm, 1)], 1)
            cum /= cum[:, -1:]
            q = np.rint(cum * (65536 - (Lp - 1))).astype(np.int64) + np.arange(Lp)
            cdfs[s, :, :Lp] = (q & 0xFFFF).astype(np.uint16)
            syms[s] = rng.integers(0, Lp - 1, N)
        out, ln = c.ac_encode(_dev(torch, cdfs.view(np.int16)), _dev(torch, syms), Lp)
        out_h, ln_h = out.cpu().numpy(), ln.cpu().numpy()
        streams = []
        for s in range(S):
            ref = orc.ac_encode_tables(cdfs[s, :, :Lp].copy(), syms[s])
            assert bytes(out_h[s, :ln_h[s]]) == ref, (Lp, s)
            streams.append(ref)
        # decode through the seam: 4-byte aligned, zero padded streams
        in_stride = (max(len(x) for x in streams) + 3) // 4 * 4 + 16
        buf = np.zeros((S, in_stride), np.uint8)
        for s, x in enumerate(streams):
            buf[s, :len(x)] = np.frombuffer(x, np.uint8)
        dec = c.ac_decode(_dev(torch, cdfs.view(np.int16)), Lp, _dev(torch, buf), _dev(torch, ln_h.astype(np.int32)), N)
        assert np.array_equal(dec.cpu().numpy(), syms)
        for s in range(S):
            assert np.array_equal(orc.ac_decode_tables(cdfs[s, :, :Lp].copy(), streams[s]), syms[s])


def _encode_to_lists(c, torch, rgb_batch):
    from llicti_amd.codec import container_to_bytestream_list
    cont, seg = c.encode(_dev(torch, rgb_batch))
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    return [container_to_bytestream_list(cont_h[b], seg_h[b]) for b in range(rgb_batch.shape[0])], cont, seg


@pytest.mark.parametrize("kind,H,W,wname", [("noise", 32, 32, "rand1337"), ("smooth", 64, 48, "trainedlike"),
                                            ("smooth", 67, 93, "trainedlike"), ("noise", 33, 64, "trainedlike"),
                                            ("smooth", 100, 131, "trainedlike"), ("noise", 95, 40, "rand1337")])
def test_whole_image_bitstream_bitexact(torch_mod, codecs, oracle_weights, kind, H, W, wname):
    """HIP container == oracle container byte for byte; both decoders accept both; lossless."""
    from oracle import oracle as orc
    from llicti_amd.codec import bytestream_list_to_container
    torch = torch_mod
    c = codecs(wname)
    W_o = oracle_weights(wname)
    rgb = make_batch(kind, 2, H, W, seed0=20)
    lists, cont, seg = _encode_to_lists(c, torch, rgb)
    for b in range(2):
        ref = orc.encode_image(rgb[b], W_o)
        assert lists[b] == ref, [(i, j) for i in range(6) for j in range(9) if lists[b][i][j] != ref[i][j]][:5]
        assert np.array_equal(orc.decode_image(lists[b], W_o), rgb[b])      # oracle decodes the GPU stream
    rec = c.decode(cont, seg, H, W)
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)                           # GPU decodes the GPU stream


@pytest.mark.parametrize("case", CASES)
def test_golden_headers_and_pixels(torch_mod, codecs, golden_index, case):
    torch = torch_mod
    info = golden_index[case]
    c = codecs(info["weights"])
    g = load_case(case)
    lists, cont, seg = _encode_to_lists(c, torch, g["rgb"][None])
    bl = lists[0]
    assert bl[0][0] == g["hdr0"].tobytes() and bl[0][1] == g["hdr_minmax"].tobytes()
    assert bl[0][2] == g["hdr_pad"].tobytes() and bl[0][3] == g["hdr_dc"].tobytes()
    H, W = g["rgb"].shape[1:]
    rec = c.decode(cont, seg, H, W).cpu().numpy()[0]
    assert np.array_equal(rec, g["reco_rgb"])
    # rate: total bytes within 0.001 bpp... of the ideal code length of the REFERENCE's own tables
    bits_ref = 0.0
    for s in range(5):
        for b in range(3):
            for cl in range(3):
                lo = g[f"clow_s{s}_b{b}_c{cl}"].astype(np.int64)
                hi = g[f"chigh_s{s}_b{b}_c{cl}"].astype(np.int64)
                bits_ref += -np.log2((hi - lo) / 65536.0).sum()
    coded_bits = 8 * sum(len(x) for row in bl[1:] for x in row)
    # 45 streams x <= 2 bytes of arithmetic-coder termination is the only overhead over the ideal length
    assert coded_bits - bits_ref < 45 * 16 + 0.001 * H * W
    assert coded_bits - bits_ref > -0.001 * H * W - 16


@pytest.mark.parametrize("kind,H,W,wname,M", [("smooth", 67, 93, "trainedlike", 1), ("noise", 64, 48, "rand1337", 4),
                                              ("smooth", 100, 131, "trainedlike", 8), ("noise", 33, 250, "trainedlike", 16),
                                              ("smooth", 256, 256, "trainedlike", 32), ("noise", 256, 384, "rand1337", 64),
                                              ("noise", 128, 192, "rand1337", 10), ("smooth", 150, 131, "trainedlike", 11),
                                              ("smooth", 512, 768, "trainedlike", 128)])
def test_rans_container_bitexact(torch_mod, codecs, oracle_weights, kind, H, W, wname, M):
    """The throughput container (rANS v3): HIP bytes == oracle bytes, both decoders invert it, and a stream that has symbols
    costs about 8 bytes over the ideal length (an empty one 251: tiny images with many streams)."""
    from oracle import oracle as orc
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list
    torch = torch_mod
    c = codecs(wname)
    W_o = oracle_weights(wname)
    rgb = make_batch(kind, 2, H, W, seed0=70)
    cont, seg = c.encode(_dev(torch, rgb), mode=MODE_RANS(M))
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    for b in range(2):
        bl = container_to_bytestream_list(cont_h[b], seg_h[b])
        ref = orc.encode_image_rans(rgb[b], W_o, M)
        assert bl == ref
        assert np.array_equal(orc.decode_image_rans(bl, W_o), rgb[b])
        n_ac = sum(len(x) for row in orc.encode_image(rgb[b], W_o) for x in row)
        n_rans = sum(len(x) for row in bl for x in row)
        assert -64 <= n_rans - n_ac <= 260 * M + 64
        assert sum(1 for row in bl[1:] for x in row if len(x)) == min(M, 32)      # M = 64 / 128: 2 / 4 streams per segment
    rec = c.decode(cont, seg, H, W, mode=MODE_RANS(M))
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)


@pytest.mark.parametrize("kind,H,W,wname,M", [("smooth", 67, 93, "trainedlike", 1), ("noise", 128, 192, "rand1337", 10),
                                              ("smooth", 150, 131, "trainedlike", 7), ("noise", 33, 250, "trainedlike", 14),
                                              ("smooth", 256, 384, "trainedlike", 5)])
def test_rans_wide_container_bitexact(torch_mod, codecs, oracle_weights, kind, H, W, wname, M):
    """Wide streams (128 lanes per stream, two 64-symbol chunks per coder step): HIP bytes == oracle bytes, both decoders invert
    it (the HIP one on a poisoned workspace), and the container is only accepted in the mode its header names."""
    from oracle import oracle as orc
    from llicti_amd._lib import LlictiError
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list, mode_of_header
    torch = torch_mod
    c = codecs(wname)
    W_o = oracle_weights(wname)
    rgb = make_batch(kind, 2, H, W, seed0=90)
    mode = MODE_RANS(M, wide=True)
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    assert mode_of_header(cont_h[0, :17]) == mode
    for b in range(2):
        bl = container_to_bytestream_list(cont_h[b], seg_h[b])
        assert bl == orc.encode_image_rans(rgb[b], W_o, M, wide=True)
        assert np.array_equal(orc.decode_image_rans(bl, W_o), rgb[b])
        assert sum(1 for row in bl[1:] for x in row if len(x)) == M
    rec = _decode_poisoned(c, cont, seg, H, W, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)
    c.decode(cont, seg, H, W, mode=MODE_RANS(M))            # same M, narrow streams: another container
    with pytest.raises(LlictiError):
        c.check()


@pytest.mark.parametrize("kind,H,W,wname,M,B", [("smooth", 67, 93, "trainedlike", 1, 2), ("noise", 128, 192, "rand1337", 9, 2),
                                                ("smooth", 150, 131, "trainedlike", 7, 3), ("noise", 33, 250, "trainedlike", 14, 2),
                                                ("smooth", 256, 384, "trainedlike", 5, 2), ("noise", 250, 131, "rand1337", 32, 1),
                                                ("smooth", 512, 768, "trainedlike", 64, 1), ("noise", 512, 768, "rand1337", 9, 2)])
def test_rans_xwide_container_bitexact(torch_mod, codecs, oracle_weights, kind, H, W, wname, M, B):
    """XWIDE streams (256 lanes per stream, ONE decoder lane per symbol: rans_decode_stage_lane_kernel): HIP bytes == oracle bytes, both
    decoders invert it (the HIP one on a poisoned workspace, whose LDS bit ring is refilled from the stream one step ahead), and the
    container is only accepted in the mode its header names.  Shapes: coded crops narrower than the band grid, stages shorter than one
    chunk, a stage that ends mid-chunk, 64 streams (two per segment), a full-size image (the ring is refilled ~60 times per stream)."""
    from oracle import oracle as orc
    from llicti_amd._lib import LlictiError
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list, mode_of_header
    torch = torch_mod
    c = codecs(wname)
    W_o = oracle_weights(wname)
    rgb = make_batch(kind, B, H, W, seed0=90)
    mode = MODE_RANS(M, wide=2)
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    assert mode_of_header(cont_h[0, :17]) == mode
    for b in range(B):
        bl = container_to_bytestream_list(cont_h[b], seg_h[b])
        assert bl == orc.encode_image_rans(rgb[b], W_o, M, wide=2)
        if b == 0:
            assert np.array_equal(orc.decode_image_rans(bl, W_o), rgb[b])
        assert sum(1 for row in bl[1:] for x in row if len(x)) == (32 if M == 64 else M)
    rec = _decode_poisoned(c, cont, seg, H, W, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)
    if M <= 14:
        c.decode(cont, seg, H, W, mode=MODE_RANS(M, wide=1))    # same M, wide streams: another container
        with pytest.raises(LlictiError):
            c.check()


@pytest.mark.parametrize("M", [1, 3, 14])
def test_rans_xwide_tail_seeds_and_chains_bitexact(torch_mod, codecs, oracle_weights, M):
    """The xwide tail's corner cases on the GPU (two chains from both ends of the payload, seeds of n raw symbols in radix A): one batch of
    images with 3, 2, 1 and 256 pixel values -- A from a handful to 511, n from 31 to 3, streams shorter than their seeds at M = 14, a flat
    image whose coded tail symbols are all free (T at the format's cap) -- HIP bytes == oracle bytes for each, decode on a poisoned workspace."""
    from oracle import oracle as orc
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list
    torch = torch_mod
    H, W = 64, 96
    r = np.random.default_rng(21)
    rgb = np.stack([(r.integers(0, 3, (3, H, W)) + 90).astype(np.uint8), (r.integers(0, 2, (3, H, W)) * 7).astype(np.uint8),
                    np.full((3, H, W), 201, np.uint8), r.integers(0, 256, (3, H, W), dtype=np.uint8)])
    c = codecs("trainedlike")
    W_o = oracle_weights("trainedlike")
    mode = MODE_RANS(M, wide=2)
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    for b in range(len(rgb)):
        assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image_rans(rgb[b], W_o, M, wide=2), b
    rec = _decode_poisoned(c, cont, seg, H, W, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)


@pytest.mark.parametrize("mode_name", ["xrans3", "rans4", "ac"])
def test_encoder_tuning_switches_bitexact(torch_mod, codecs, mode_name):
    """The encoder's schedule switches change WHEN kernels run, never what they write: levels 4..1 on a side stream (`enc_side_levels`: the
    per-band pairs launches instead of the one-launch-per-level form), the mixed-size code path on an equal-size batch (`force_ragged`: per-image
    tables, tile lists), every CNN tile form (`cnn_tile_rows` 16 / 8 / 4 / the round-3 rule), alone and with the tile lists -- containers
    byte-identical to the default schedule's, decode lossless."""
    from llicti_amd.codec import mode_of_name
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = make_batch("smooth", 3, 150, 131, seed0=321)
    mode = mode_of_name(mode_name)
    x = _dev(torch, rgb)
    cont0, seg0 = c.encode(x, mode=mode)
    c.check()
    seg_h = seg0.cpu().numpy()
    lens = seg_h.sum(axis=1)
    ref = [cont0[b, :int(lens[b])].cpu().numpy().copy() for b in range(len(rgb))]
    try:
        for ragged in ((0, 1) if mode_name != "ac" else (0,)):      # (the reference-format container codes equal sizes only: no tile lists there)
            c.set_tuning("force_ragged", ragged)
            for key, val in (("enc_side_levels", 0), ("enc_side_levels", 1), ("cnn_tile_rows", 16), ("cnn_tile_rows", 8), ("cnn_tile_rows", 4), ("cnn_tile_rows", -1)):
                c.set_tuning(key, val)
                cont, seg = c.encode(x, mode=mode)
                c.check()
                assert np.array_equal(seg.cpu().numpy(), seg_h), (ragged, key, val)
                for b in range(len(rgb)):
                    assert np.array_equal(cont[b, :int(lens[b])].cpu().numpy(), ref[b]), (ragged, key, val, b)
                if key == "cnn_tile_rows":
                    rec = _decode_poisoned(c, cont0, seg0, 150, 131, mode)      # the decoder's launches in that form too
                    assert np.array_equal(rec.cpu().numpy(), rgb), (ragged, key, val)
                c.set_tuning(key, 0)
    finally:
        for key in ("enc_side_levels", "force_ragged", "cnn_tile_rows"):
            c.set_tuning(key, 0)
    rec = _decode_poisoned(c, cont0, seg0, 150, 131, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)


def test_rans_known_answer_hip(torch_mod, codecs):
    """The committed known-answer vectors of the rANS containers (tests/golden/rans_vectors.npz, frozen by test_rans_known_answer on the CPU:
    64- and 128-lane streams in the v3 layout, xwide streams in the v4 layout): the HIP encoder reproduces the stored bytes, the HIP decoder
    turns the stored bytes back into the fixture's pixels -- without the oracle in the loop; and the two larger xwide v4 containers, stored as
    hashes (tails that fill their payload and spill; one chain / two chains)."""
    import hashlib
    import os
    from conftest import GOLDEN
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    vec = np.load(os.path.join(GOLDEN, "rans_vectors.npz"))
    for case, wname in [("smooth_67x93_tl", "trainedlike"), ("noise_32x32_rand", "rand1337"), ("noise_33x64_tl", "trainedlike")]:
        c = codecs(wname)
        rgb = load_case(case)["rgb"]
        H, W = rgb.shape[1:]
        for key, mode in (("M1", MODE_RANS(1)), ("M4", MODE_RANS(4)), ("W3", MODE_RANS(3, wide=True)), ("X4", MODE_RANS(3, wide=2))):
            want, lens = vec[f"{case}_{key}_bytes"], vec[f"{case}_{key}_seglen"]
            seg_want = np.concatenate([lens[:4], lens[9:]]).astype(np.int32)        # bytestream_list rows of 9 -> the 49 segments
            cont, seg = c.encode(_dev(torch, rgb[None]), mode=mode)
            c.check()
            n = int(seg.sum().item())
            assert np.array_equal(seg[0].cpu().numpy(), seg_want), (case, key)
            assert n == want.size and np.array_equal(cont[0, :n].cpu().numpy(), want), (case, key)
            cont2 = torch.zeros_like(cont)
            cont2[0, :n] = _dev(torch, want)
            rec = _decode_poisoned(c, cont2, _dev(torch, seg_want[None]), H, W, mode)
            assert np.array_equal(rec[0].cpu().numpy(), rgb), (case, key)
    for key, kind, H, W, seed, wname, M in (("X4big_smooth", "smooth", 256, 384, 11, "trainedlike", 4), ("X4big_noise", "noise", 96, 160, 3, "rand1337", 2)):
        c = codecs(wname)
        rgb = make_image(kind, H, W, seed)
        cont, seg = c.encode(_dev(torch, rgb[None]), mode=MODE_RANS(M, wide=2))
        c.check()
        lens = vec[f"{key}_seglen"]
        assert np.array_equal(seg[0].cpu().numpy(), np.concatenate([lens[:4], lens[9:]]).astype(np.int32)), key
        n = int(seg.sum().item())
        assert hashlib.sha256(cont[0, :n].cpu().numpy().tobytes()).digest() == vec[f"{key}_sha256"].tobytes(), key


def test_rans_kodak_batch_roundtrip(torch_mod, codecs):
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    c = codecs("rand1337")
    rgb = make_batch("noise", 3, 512, 768, seed0=80)
    for M in (8, 16):
        cont, seg = c.encode(_dev(torch, rgb), mode=MODE_RANS(M))
        rec = c.decode(cont, seg, 512, 768, mode=MODE_RANS(M))
        c.check()
        assert np.array_equal(rec.cpu().numpy(), rgb)
    # a container is only accepted in the mode its header names
    from llicti_amd._lib import LlictiError
    c.decode(cont, seg, 512, 768, mode=MODE_RANS(8))
    with pytest.raises(LlictiError):
        c.check()


def test_kodak_shape_properties(torch_mod, codecs):
    """BASELINE.json full size (768x512): round trip, batch independence, idempotence."""
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = np.concatenate([make_batch("smooth", 2, 512, 768, seed0=40), make_batch("noise", 1, 512, 768, seed0=41)])
    lists, cont, seg = _encode_to_lists(c, torch, rgb)
    rec = c.decode(cont, seg, 512, 768)
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)
    # coding an image alone gives the same bytes as coding it inside a batch
    l1, _, _ = _encode_to_lists(c, torch, rgb[1:2])
    assert l1[0] == lists[1]
    # encode(decode(encode(x))) == encode(x)
    l2, _, _ = _encode_to_lists(c, torch, rec.cpu().numpy())
    assert l2 == lists
    # sanity of the probability model (not a reference number): i.i.d. noise costs more than smooth content
    bpp = [8 * sum(len(x) for row in l for x in row) / (512 * 768) for l in lists]
    assert bpp[2] > bpp[0] and bpp[2] > bpp[1]


def test_large_odd_image_roundtrip(torch_mod, codecs):
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = make_batch("smooth", 1, 1080 // 2 + 1, 1920 // 2 + 3, seed0=50)
    lists, cont, seg = _encode_to_lists(c, torch, rgb)
    rec = c.decode(cont, seg, rgb.shape[2], rgb.shape[3])
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)


def _wild_state_dict(scale, seed):
    """Seed-1337 architecture with weights blown up: wide, overlapping, far-off-range mixtures (stress for the
    decoders' search: an approximate hint that is often wrong must still end in the exact symbol)."""
    sd = {k: np.array(v) for k, v in load_state_dict("rand1337").items()}
    rng = np.random.default_rng(seed)
    for k in sd:
        if k.endswith("layers1toL.2.weight") or k.endswith("layers1toL.2.bias"):
            sd[k] = (sd[k] * scale + rng.standard_normal(sd[k].shape) * 0.05 * scale).astype(np.float32)
    return sd


@pytest.mark.parametrize("scale,kind", [(6.0, "noise"), (25.0, "smooth"), (120.0, "noise")])
def test_wild_weights_bitexact_both_containers(torch_mod, scale, kind):
    from llicti_amd.codec import HipCodec, MODE_RANS, container_to_bytestream_list
    from llicti_amd.weights import pack_state_dict
    from oracle import oracle as orc
    torch = torch_mod
    sd = _wild_state_dict(scale, int(scale))
    c = HipCodec("cuda:0")
    c.load_state_dict(sd)
    W_o = orc.Weights(pack_state_dict(sd))
    rgb = make_batch(kind, 2, 48, 80, seed0=int(scale))
    try:
        lists, cont, seg = _encode_to_lists(c, torch, rgb)
        for b in range(2):
            assert lists[b] == orc.encode_image(rgb[b], W_o), b
        rec = c.decode(cont, seg, 48, 80)
        c.check()
        assert np.array_equal(rec.cpu().numpy(), rgb)
        for M, wide in ((4, 0), (3, 1), (3, 2)):               # four, two and ONE decoder lane per symbol: each kernel's own gallop-and-bisect path
            mode = MODE_RANS(M, wide=wide)
            cont, seg = c.encode(_dev(torch, rgb), mode=mode)
            c.check()
            seg_h, cont_h = seg.cpu().numpy(), cont.cpu().numpy()
            for b in range(2):
                assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image_rans(rgb[b], W_o, M, wide), (b, M, wide)
            rec = c.decode(cont, seg, 48, 80, mode=mode)
            c.check()
            assert np.array_equal(rec.cpu().numpy(), rgb), (M, wide)
    finally:
        c.close()


def test_4k_image_roundtrip_rans(torch_mod, codecs):
    """BASELINE.json configs[3] shape: one 3840x2160 image (level 4 is 68x120 with padH = 1)."""
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    c = codecs("trainedlike")
    rng = np.random.default_rng(99)
    small = make_batch("smooth", 1, 270, 480, seed0=70)[0]
    rgb = np.repeat(np.repeat(small, 8, axis=1), 8, axis=2)             # 2160 x 3840, smooth ...
    rgb = (rgb.astype(np.int16) + rng.integers(-3, 4, size=rgb.shape)).clip(0, 255).astype(np.uint8)[None]   # ... plus fine noise
    cont, seg = c.encode(_dev(torch, rgb), mode=MODE_RANS(16))
    c.check()
    rec = c.decode(cont, seg, 2160, 3840, mode=MODE_RANS(16))
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)
    hdr = cont[0, :17].cpu().numpy()
    assert hdr[1] == 68 and hdr[2] == 120 and int(hdr[15]) | (int(hdr[16]) << 8) == 2      # pad flags: level 4 rows only
    bpp = 8.0 * float(seg.sum().item()) / (2160 * 3840)
    assert 0.5 < bpp < 24.0


@pytest.mark.parametrize("H,W", [(8160, 32), (32, 8160), (1055, 2049)])
def test_extreme_shapes_roundtrip(torch_mod, codecs, H, W):
    """The format's maximum dimension (h4, w4 are uint8: H, W <= 8160) at minimum width, and an odd mid-size
    image through the reference (AC) container."""
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = make_batch("smooth", 1, min(H, 1055), min(W, 2049), seed0=71)
    if rgb.shape[2] != H or rgb.shape[3] != W:
        rgb = np.ascontiguousarray(np.resize(rgb, (1, 3, H, W)))
    for mode in ((0, MODE_RANS(2), MODE_RANS(2, wide=2)) if H * W < 3_000_000 else (MODE_RANS(2), MODE_RANS(3, wide=2))):
        cont, seg = c.encode(_dev(torch, rgb), mode=mode)
        c.check()
        rec = c.decode(cont, seg, H, W, mode=mode)
        c.check()
        assert np.array_equal(rec.cpu().numpy(), rgb), (H, W, mode)


def test_malformed_container_rejected(torch_mod, codecs):
    from llicti_amd._lib import LlictiError, EFORMAT
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = make_batch("smooth", 1, 64, 64, seed0=60)
    cont, seg = c.encode(_dev(torch, rgb))
    c.check()
    bad = cont.clone()
    bad[0, 0] = 4                               # wrong number of scales (assert at LLICTI_nets.py:424)
    c.decode(bad, seg, 64, 64)
    with pytest.raises(LlictiError) as e:
        c.check()
    assert e.value.code == EFORMAT
    with pytest.raises(LlictiError):
        c.decode(cont, seg, 64, 96)             # header says 64x64
        c.check()
    rec = c.decode(cont, seg, 64, 64)           # context still usable afterwards
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)


def test_reference_api_roundtrip(torch_mod, capsys):
    """graphs.models.LLICTI_nets.LLICTI: compress / decompres with the reference's container and seeded init; `x_ycocg` (read back from the
    encoder's workspace, no second lift) equals the reference's; decompres(..., xorg=) runs the reference's diagnostic (LLICTI_nets.py:167-171)."""
    torch = torch_mod
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    from oracle import oracle as orc
    from llicti_amd.weights import pack_state_dict
    torch.manual_seed(1337)
    model = LLICTI(default_config()).to("cuda:0").eval()
    rgb = make_image("noise", 67, 93, 1)
    x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0")
    bl, x_ycocg = model.compress(x)
    assert len(bl) == 6 and all(len(r) == 9 for r in bl)
    g = load_case("noise_67x93_rand")           # same image, same seeded weights as the fixture
    assert bl[0][1] == g["hdr_minmax"].tobytes() and bl[0][3] == g["hdr_dc"].tobytes()
    assert np.array_equal(x_ycocg.cpu().numpy()[0], g["x_ycocg_f32"])
    x_reco = model.decompres(bl, torch.device("cuda:0"), xorg=x_ycocg)
    assert "does NOT match" not in capsys.readouterr().out
    bad = x_ycocg.clone()
    bad[0, 1, 5, 7] += 2.0 / 255
    model.decompres(bl, torch.device("cuda:0"), xorg=bad)
    assert "Decoded YCoCg img does NOT match original YCoCg image perfectly! The maximum of absolute error is 2.0000" in capsys.readouterr().out
    assert float(((x - x_reco) * 255).abs().max()) < 0.5          # the reference's own check (llicti_agent.py:151-152)
    assert np.array_equal((x_reco * 255).round().to(torch.uint8).cpu().numpy()[0], rgb)   # and exact as integers
    W_o = orc.Weights(pack_state_dict(model.state_dict()))
    assert orc.encode_image(rgb, W_o) == bl
    # the throughput container through the same API
    m2 = LLICTI(default_config(container="rans4")).to("cuda:0").eval()
    m2.load_state_dict(model.state_dict())
    bl2, _ = m2.compress(x)
    assert bl2 == orc.encode_image_rans(rgb, W_o, 4)
    x2 = model.decompres(bl2, torch.device("cuda:0"))          # any model instance decodes either container
    assert np.array_equal((x2 * 255).round().to(torch.uint8).cpu().numpy()[0], rgb)


def test_decompres_batch_mixed_sizes(torch_mod, oracle_weights):
    """LLICTI.decompres_batch on bytestream_lists of DIFFERENT sizes (rANS containers: decode_batch_async then returns a flat buffer) gives a
    list of [1,3,H,W] tensors, each what decompres() returns for that image; equal sizes still give one [B,3,H,W] tensor; the reference format refuses
    mixed sizes up front."""
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    torch = torch_mod
    torch.manual_seed(1337)
    model = LLICTI(default_config(container="xrans2")).to("cuda:0").eval()
    imgs = [make_image("smooth", 96, 128, 1), make_image("noise", 67, 93, 2), make_image("smooth", 150, 131, 3)]
    lists = model.encode_batch_async(imgs).lists()
    out = model.decompres_batch(lists, torch.device("cuda:0"))
    assert isinstance(out, list) and len(out) == 3
    for o, im, bl in zip(out, imgs, lists):
        assert tuple(o.shape) == (1, 3) + im.shape[1:] and o.dtype == torch.float32
        assert np.array_equal((o * 255).round().to(torch.uint8).cpu().numpy()[0], im)
        assert torch.equal(o, model.decompres(bl, torch.device("cuda:0")))
    same = model.decompres_batch([lists[0], lists[0]], torch.device("cuda:0"))
    assert torch.is_tensor(same) and tuple(same.shape) == (2, 3, 96, 128)
    model.set_container("ac")
    ac = [model.compress(torch.from_numpy(im.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0"))[0] for im in imgs[:2]]
    with pytest.raises(ValueError):
        model.decompres_batch(ac, torch.device("cuda:0"))


def test_agent_eval_model(torch_mod, caplog):
    import logging
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    caplog.set_level(logging.INFO)
    agent = LLICTIAgent(default_config(test_data="synthetic:48x80x2"))
    res = agent.run()
    agent.finalize()
    assert len(res) == 2 and all(r["max_abs_err"] < 1e-3 for r in res)   # float32 uint8/255 on host vs device division
    assert sum("Check: Decoded img matches original" in r.message for r in caplog.records) == 2
    assert all(len(r["rates"]) == 6 and all(len(row) == 9 for row in r["rates"]) for r in res)


def test_agent_batched_eval_equals_unbatched(torch_mod, tmp_path, caplog, oracle_weights):
    """LLICTIAgent.eval_model with config.eval_batch codes `eval_batch` CONSECUTIVE test images per call, WHATEVER THEIR SIZES (the
    reference's loader yields arbitrary sizes one at a time, dataloaders/image_dl.py:40-45), through the batched, software-pipelined path
    (LLICTI.encode_batch_async / decode_batch_async on lists of images -> llicti_encode_images_v / llicti_decode_images_v).  Its
    bytestream_lists equal, image by image, what the one-image loop writes in the same container, and the oracle's for two images; per-image
    log lines (in order), rates and the lossless check are the same.  container "auto" picks, per batch, the container its smallest image
    allows; in the reference-format container a batch closes where the size changes; an in-memory data set works."""
    import logging
    from oracle import oracle as orc
    from llicti_amd import fileio
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.codec import mode_of_header, mode_of_name
    from llicti_amd.config import default_config
    torch = torch_mod
    caplog.set_level(logging.INFO)
    # 10 images of six sizes, interleaved, as files (the reference's test loader reads a directory): batches of 3, 3, 3, 1
    sizes = [(96, 128), (96, 128), (67, 93), (128, 96), (96, 128), (150, 131), (67, 93), (150, 131), (97, 351), (96, 128)]
    imgs = [make_image("noise" if i % 2 else "smooth", h, w, 400 + i) for i, (h, w) in enumerate(sizes)]
    for i, im in enumerate(imgs):
        fileio.write_image(str(tmp_path / f"img_{i:02d}.ppm"), im)
    cname = "xrans2"
    a_b = LLICTIAgent(default_config(test_data=str(tmp_path), eval_batch=3, container=cname, keep_streams=True))
    res_b = a_b.run()
    a_u = LLICTIAgent(default_config(test_data=str(tmp_path), container=cname, eval_batch=1, keep_streams=True))
    res_u = a_u.run()
    import re
    lines = [r.message for r in caplog.records if "Check: Decoded img matches original" in r.message]
    assert len(res_b) == len(res_u) == 10 and len(lines) == 20
    heads = [tuple(int(v) for v in re.match(r"\s*(\d+)\s+(\d+)x\s*(\d+) ", ln).groups()) for ln in lines]
    assert heads[:10] == heads[10:] == [(i, h, w) for i, (h, w) in enumerate(sizes)]      # both runs: one line per image, in the loader's order
    assert [r["batch"] for r in res_b] == [3] * 9 + [1]
    for rb, ru, im in zip(res_b, res_u, imgs):
        assert (rb["idx"], rb["H"], rb["W"]) == (ru["idx"], ru["H"], ru["W"]) == (rb["idx"], im.shape[1], im.shape[2])
        assert rb["bytestream_list"] == ru["bytestream_list"]
        assert rb["rates"] == ru["rates"] and rb["bpsp"] == ru["bpsp"]
        assert rb["max_abs_err"] == 0.0 and ru["max_abs_err"] < 1e-3
        assert mode_of_header(rb["bytestream_list"]) == mode_of_name(cname)
    W_o = oracle_weights("rand1337")                     # the agent's seed-1337 default init (no checkpoint in the test directory)
    for i in (2, 8):
        assert res_b[i]["bytestream_list"] == orc.encode_image_rans(imgs[i], W_o, 2, 2)
    # ... and both runs logged the same mean rate table ("te" rows, loggers/rate.py) -- once each
    tables = [r.message for r in caplog.records if "scl4->" in r.message]
    assert len(tables) == 2 and tables[0].split("(")[0:-1] == tables[1].split("(")[0:-1]      # identical but for the trailing time stamp
    # container "auto": each image's own mode -- a function of the image, not of its batch (these sizes are too small for xwide streams but for
    # 150x131 / 97x351 next to one another: a call shares a lane kind, so a batch with a tiny image goes out in one 64-lane stream per image)
    from llicti_amd.codec import MODE_RANS, auto_modes, image_streams
    a_a = LLICTIAgent(default_config(test_data=imgs, eval_batch=4, container="auto", keep_streams=True))
    res_a = a_a.run()
    assert len(res_a) == 10 and all(r["max_abs_err"] == 0.0 for r in res_a)
    for k0 in (0, 4, 8):
        want = auto_modes(sizes[k0:k0 + 4])
        for r, m in zip(res_a[k0:k0 + 4], want):
            assert not (m & 0x10000) and mode_of_header(r["bytestream_list"]) == m == MODE_RANS(1)
    big = [(224, 301), (256, 384), (160, 352), (321, 481)]                  # one call, several stream counts' worth of sizes
    big_imgs = [make_image("smooth", h, w, 430 + i) for i, (h, w) in enumerate(big)]
    a_g = LLICTIAgent(default_config(test_data=big_imgs, eval_batch=4, container="auto", keep_streams=True))
    res_g = a_g.run()
    got = [mode_of_header(r["bytestream_list"]) for r in res_g]
    assert got == [MODE_RANS(image_streams(h, w), wide=2) for h, w in big] and len(set(got)) > 1 and all(r["max_abs_err"] == 0.0 for r in res_g)
    W_t = W_o
    for i in (1, 3):                                                        # ... and each is the oracle's "auto" encode of that image alone
        assert res_g[i]["bytestream_list"] == orc.encode_image_rans(big_imgs[i], W_t, image_streams(*big[i]), 2, auto=True)
    # in-memory data set, default (reference-format) container, batch of 4: a batch closes where the size changes; equals the oracle
    a_m = LLICTIAgent(default_config(test_data=imgs[:4], eval_batch=4, container="ac", keep_streams=True))
    res_m = a_m.run()
    assert len(res_m) == 4 and all(r["max_abs_err"] == 0.0 for r in res_m)
    assert [r["batch"] for r in res_m] == [2, 2, 1, 1]
    assert res_m[1]["bytestream_list"] == orc.encode_image(imgs[1], W_o)
    assert res_m[2]["bytestream_list"] == orc.encode_image(imgs[2], W_o)


def test_agent_batched_lossless_check_reports_a_difference(torch_mod, caplog):
    """The batched eval's lossless check (llicti_agent.py:151-162) is one comparison pass per batch; the SIZE of an error is worked out only for an
    image that differs.  A decoder that returns one wrong sub-pixel (injected here: the decoded buffer of the second batch is altered before the
    check reads it) must end in the reference's "does NOT match" line for exactly that image, with the right magnitude, and leave the others alone."""
    import logging
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    torch = torch_mod
    caplog.set_level(logging.INFO)
    imgs = [make_image("smooth" if i % 2 else "noise", h, w, 700 + i) for i, (h, w) in enumerate([(96, 128), (67, 93), (96, 128), (128, 96), (96, 128)])]
    a = LLICTIAgent(default_config(test_data=imgs, eval_batch=2, container="xrans1", keep_streams=True))
    real = a.model.decode_batch_async
    calls = {"n": 0}

    def tampered(lists, devc=None, slot=0, flat=False):
        rec, Hs, Ws = real(lists, devc, slot=slot, flat=flat)
        calls["n"] += 1
        if calls["n"] == 2:                                # batch 1 = images 2 and 3: image 3's first sub-pixel is off by 7 grey levels
            o = 3 * Hs[0] * Ws[0]
            rec[o] = (rec[o].to(torch.int16) + 7).clamp(0, 255).to(torch.uint8) if int(rec[o]) <= 248 else rec[o] - 7
        return rec, Hs, Ws
    a.model.decode_batch_async = tampered
    res = a.run()
    assert [r["max_abs_err"] for r in res] == [0.0, 0.0, 0.0, 7.0, 0.0]
    bad = [r.message for r in caplog.records if "does NOT match" in r.message]
    assert len(bad) == 1 and bad[0].lstrip().startswith("3 ") and "7.0000" in bad[0], bad
    assert sum("Check: Decoded img matches original" in r.message for r in caplog.records) == 4


def test_agent_auto_container_is_a_function_of_the_image(torch_mod, caplog):
    """Container "auto" (VERDICT r5 #1 / weak #1): an image's container is a function of THE IMAGE -- its size gives a stream count, the encoder
    adjusts it on the device by what the image's own last stage costs -- not of what was coded before it, next to it, or of eval_batch.  A data
    set that mixes three content classes (flat images, whose last stage cannot fill the payloads; natural-like images; uniform noise) coded
    with eval_batch 1, 2, 3, 6 and in reverse order gives every image the same bytes every time, each equal to the oracle's "auto" encode of that
    image alone; the cheap images get fewer streams than the natural-like ones, the noise images more.  A config WITHOUT container / eval_batch
    keys (the reference's own llicti_A.json) runs exactly this path, and says so."""
    import logging
    from oracle import oracle as orc
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.codec import MODE_RANS, auto_counts, image_streams, mode_of_header
    from llicti_amd.config import default_config
    from llicti_amd.weights import load_reference_state_dict, pack_state_dict
    torch = torch_mod
    caplog.set_level(logging.INFO)
    from conftest import load_state_dict
    sd = load_state_dict("trainedlike")
    W_c = orc.Weights(pack_state_dict(sd))
    H, W = 256, 384
    # three content classes under ONE set of weights: flat images (a last stage that costs next to nothing: it cannot fill the payloads of the size
    # rule's count), natural-like ones, uniform noise (expensive symbols)
    cheap = [np.full((3, H, W), 77, np.uint8), np.full((3, H, W), 180, np.uint8)]      # (grey: Co = Cg = 0, where these weights' mixtures sit)
    imgs = [cheap[0], make_image("smooth", H, W, 21), make_image("noise", H, W, 22), cheap[1], make_image("noise", H, W, 23), make_image("smooth", H, W, 24)]

    def run(data, **cfg):
        a = LLICTIAgent(default_config(test_data=data, keep_streams=True, **cfg))
        load_reference_state_dict(a.model, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        res = a.run()
        assert all(r["max_abs_err"] < 1e-3 for r in res)            # (0.0 in the batched path; the one-image loop compares float32 uint8 / 255 of host and device)
        return [r["bytestream_list"] for r in res]
    base = run(imgs, container="auto", eval_batch=6)
    for eb in (1, 2, 3):
        assert run(imgs, container="auto", eval_batch=eb) == base, eb
    assert run(imgs[::-1], container="auto", eval_batch=4)[::-1] == base
    n_info = len([r for r in caplog.records if "eval_model: container" in r.message])
    assert run(imgs, **{}) == base                                 # no container / eval_batch key: the same path ...
    said = [r.message for r in caplog.records if "eval_model: container" in r.message][n_info:]
    assert len(said) == 1 and '"auto"' in said[0] and "eval_batch 24" in said[0], said      # ... and one line saying so
    M = image_streams(H, W)
    lo, _, mid, hi = auto_counts(M)
    counts = [mode_of_header(b) & 0xFF for b in base]
    assert counts == [lo, mid, hi, lo, hi, mid], (counts, (lo, mid, hi))
    for i, b in enumerate(base):
        assert mode_of_header(b) == MODE_RANS(counts[i], wide=2)
        assert b == orc.encode_image_rans(imgs[i], W_c, M, 2, auto=True), i
    # the reference's byte format stays one key away
    ac = run(imgs[:2], container="ac")
    assert ac[1] == orc.encode_image(imgs[1], W_c)


def test_native_client_without_torch(torch_mod, tmp_path):
    """The drop-in boundary is a C library: tools/native_client.cpp -- no Python, no PyTorch in its process, plain hipMalloc buffers -- is
    compiled against include/llicti_hip.h, linked with the in-tree libllicti_hip.so and run: encode -> decode on an overwritten workspace
    is lossless in the throughput container and in the reference's format, and a corrupted container is reported (LLICTI_EFORMAT, the
    image named, its neighbours intact).  (torch_mod only gates the test on a GPU box.)"""
    import os
    import shutil
    import subprocess
    from llicti_amd import _lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so_dir = os.path.dirname(_lib.SO_PATH)
    exe = str(tmp_path / "native_client")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "native_client.cpp"),
                           "-L", so_dir, "-lllicti_hip", "-Wl,-rpath," + so_dir, "-o", exe])
    for args in ([], ["3", "150", "131", "2"], ["2", "256", "384", "10"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "native client ok" in r.stdout, (args, r.stdout[-800:], r.stderr[-800:])
        assert r.stdout.count("lossless: yes") == 2 and "per-image status names it: yes, other images intact: yes" in r.stdout, r.stdout


def test_cli_file_roundtrip(torch_mod, tmp_path, capsys):
    """image file -> .llic -> image file through the command-line front end, both containers."""
    from llicti_amd import cli, fileio
    rgb = make_image("smooth", 75, 131, 90)
    src = tmp_path / "in.ppm"
    fileio.write_image(src, rgb)
    for container in ("ac", "rans4"):
        mid, dst = tmp_path / f"x_{container}.llic", tmp_path / f"out_{container}.png"
        assert cli.main(["encode", str(src), str(mid), "--container", container]) == 0
        assert cli.main(["info", str(mid)]) == 0
        assert cli.main(["decode", str(mid), str(dst)]) == 0
        assert np.array_equal(fileio.read_image(dst), rgb)
    out = capsys.readouterr().out
    assert "131x75" in out and "container rans4" in out


@pytest.mark.parametrize("case,wname", [("fwd_smooth_64x96_tl", "trainedlike"), ("fwd_noise_32x64_rand", "rand1337")])
def test_forward_selfinfo(torch_mod, codecs, oracle_weights, case, wname):
    """LLICTI.forward on the HIP path -- float lift bit-exact to the reference's, self
    information equal to the oracle's up to log2's last bits and within 1e-3 bits + 1e-4 relative of the
    reference's own output; its sum is the ideal code length the coder's byte count is compared with."""
    import os
    from conftest import GOLDEN
    from oracle import oracle as orc
    torch = torch_mod
    g = np.load(os.path.join(GOLDEN, f"{case}.npz"))
    rgb = g["rgb"]
    c = codecs(wname)
    fp = c.lift_train(_dev(torch, rgb[None]))
    assert np.array_equal(fp.cpu().numpy()[0], orc.lift_train(rgb))
    infos = c.forward_selfinfo(_dev(torch, rgb[None]))
    ref_o = orc.forward(rgb, oracle_weights(wname))
    for s in range(5):
        got = infos[s].cpu().numpy()[0]
        assert np.allclose(got, ref_o[s], rtol=2e-6, atol=2e-6), (s, float(np.abs(got - ref_o[s]).max()))
        assert np.allclose(got, g[f"selfinfo_s{s}"], rtol=1e-4, atol=1e-3), s
    tot = float(sum(t.double().sum() for t in infos))
    assert abs(tot - float(g["total_bits"][0])) < 1e-5 * float(g["total_bits"][0])
    # module-level API + the coder's output against the estimate (different lift rounding, same model: close, not equal)
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    model = LLICTI(default_config()).to("cuda:0").eval()
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in load_state_dict(wname).items()})
    x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0")
    out = model.forward(x)
    assert all(torch.equal(a, b) for a, b in zip(out, infos))
    bl, _ = model.compress(x)
    coded_bits = 8 * sum(len(sx) for row in bl[1:] for sx in row)
    if wname == "trainedlike":      # with sigma-floor random weights the estimate's 1e-9 floor (29.9 bits) overshoots the coder's 16-bit cap
        assert 0.9 * tot < coded_bits < 1.1 * tot + 45 * 40
    else:
        assert coded_bits < tot
    with pytest.raises(ValueError):
        model.forward(x[:, :, :31, :])


def _edge_images():
    rng = np.random.default_rng(123)
    H, W = 40, 72
    imgs = {
        "black": np.zeros((3, H, W), np.uint8),
        "white": np.full((3, H, W), 255, np.uint8),
        "constant": np.broadcast_to(np.array([17, 200, 93], np.uint8)[:, None, None], (3, H, W)).copy(),   # Co, Cg ranges of one value: Lp = 2
        "grey_ramp": np.broadcast_to((np.arange(W) * 3 % 256).astype(np.uint8)[None, None, :], (3, H, W)).copy(),   # Co = Cg = 0 everywhere
        "extremes": rng.choice(np.array([0, 255], np.uint8), size=(3, H, W)),                              # Co, Cg span [-255, 255]: Lp = 512
        "one_pixel": np.zeros((3, H, W), np.uint8),
    }
    imgs["one_pixel"][:, 13, 29] = (255, 0, 128)
    return imgs


@pytest.mark.parametrize("wname", ["rand1337", "trainedlike"])
def test_degenerate_images_bitexact_and_roundtrip(torch_mod, codecs, oracle_weights, wname):
    """Alphabets of one symbol (Lp = 2), full-width alphabets (Lp = 512), saturated pixels: both containers
    against the oracle, byte for byte, and back to the pixels."""
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs(wname)
    W_o = oracle_weights(wname)
    imgs = _edge_images()
    rgb = np.stack(list(imgs.values()))
    lists, cont, seg = _encode_to_lists(c, torch, rgb)
    rec = c.decode(cont, seg, rgb.shape[2], rgb.shape[3])
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)
    cont_r, seg_r = c.encode(_dev(torch, rgb), mode=MODE_RANS(2))
    c.check()
    rec = c.decode(cont_r, seg_r, rgb.shape[2], rgb.shape[3], mode=MODE_RANS(2))
    c.check()
    assert np.array_equal(rec.cpu().numpy(), rgb)
    cont_h, seg_h = cont_r.cpu().numpy(), seg_r.cpu().numpy()
    for b, name in enumerate(imgs):
        assert lists[b] == orc.encode_image(rgb[b], W_o), name
        assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image_rans(rgb[b], W_o, 2), name


def test_agent_validate_mode(torch_mod, oracle_weights, caplog):
    """agents/llicti_agent.py:85-103: estimated rate over the image set = mean of sum(self-information) / numel * 3,
    images replicate-padded to multiples of 32 first (:105-113); checked against the oracle's forward()."""
    import logging
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    from oracle import oracle as orc
    caplog.set_level(logging.INFO)
    agent = LLICTIAgent(default_config(test_data="synthetic:40x70x2", mode="validate"))
    got = agent.run()
    W_o = oracle_weights("rand1337")          # the agent seeds the default init with 1337 as well
    tot = 0.0
    for i in range(2):
        rgb = np.random.default_rng(i).integers(0, 256, size=(3, 40, 70), dtype=np.uint8)
        pad = np.pad(rgb, ((0, 0), (0, 24), (0, 26)), mode="edge")
        tot += sum(float(t.astype(np.float64).sum()) for t in orc.forward(pad, W_o)) / pad.size * 3
    assert abs(got - tot / 2) < 1e-4 * (tot / 2)
    assert any("Valid Epoch" in r.message for r in caplog.records)


# ------------------------------------------------------------------------------------------------ round 2
def _decode_poisoned(c, cont, seg, H, W, mode=0):
    """Decode on a workspace that holds NO trace of the encode: both calls carve planes / fplanes at the same offsets of
    the same cached workspace, so a decoder reading a not-yet-decoded pixel would otherwise find the right value there."""
    if mode & 0x10000:                   # an "auto" ENCODER mode: the containers say which count the encoder picked (one per call here)
        dm = sorted(set(c.container_modes(cont)))
        assert len(dm) == 1, dm
        mode = dm[0]
    c.workspace(cont.shape[0], H, W, mode)
    c.poison_workspace(0xA5)
    out = c.decode(cont, seg, H, W, mode=mode)
    c.check()
    return out


def _oracle_container(rgb, W_o, mode):
    """The oracle's bytestream_list of one image in a codec mode: reference format, a rANS container of a fixed count, or the encoder's "auto"."""
    from oracle import oracle as orc
    if mode == 0:
        return orc.encode_image(rgb, W_o)
    return orc.encode_image_rans(rgb, W_o, mode & 0xFF, ((mode & 0xF00) - 0x100) // 0x200, auto=bool(mode & 0x10000))


@pytest.mark.parametrize("mode_name,B,H,W", [("ac", 3, 128, 192), ("ac", 2, 250, 131), ("ac_anchors", 3, 128, 192),
                                             ("ac_anchors", 2, 250, 131), ("rans4", 3, 128, 192),
                                             ("rans16", 2, 250, 131), ("rans1", 2, 67, 93), ("wrans3", 3, 128, 192), ("xrans3", 3, 128, 192),
                                             ("xrans9", 2, 250, 131)])
def test_decode_on_poisoned_workspace(torch_mod, codecs, mode_name, B, H, W):
    """Losslessness of the PIPELINED decoders (3-stream AC chunk pipeline with ac_chunks() > 1, rANS next-step
    prefetch, CNN halo / odd-edge clamps), with B > 1: the workspace is overwritten with 0xA5 between encode and
    decode, and a second decode runs on a FRESH context that never saw the encoder at all."""
    from llicti_amd.codec import HipCodec, mode_of_name
    torch = torch_mod
    c = codecs("trainedlike")
    mode = 0 if mode_name.startswith("ac") else mode_of_name(mode_name)
    # the AC decoder has two table forms (full rows for few images, anchor rows for many): both are run here
    anchors = mode_name == "ac_anchors"
    rgb = np.concatenate([make_batch("smooth", B - 1, H, W, seed0=300), make_batch("noise", 1, H, W, seed0=301)])
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    c.set_tuning("ac_anchor_min_batch", 1 if anchors else 96)
    try:
        rec = _decode_poisoned(c, cont, seg, H, W, mode)
    finally:
        c.set_tuning("ac_anchor_min_batch", 96)
    assert np.array_equal(rec.cpu().numpy(), rgb)
    c2 = HipCodec("cuda:0")
    try:
        c2.load_state_dict(load_state_dict("trainedlike"))
        c2.set_tuning("ac_anchor_min_batch", 1 if anchors else 96)
        c2.workspace(B, H, W, mode)
        c2.poison_workspace(0x5A)
        rec2 = c2.decode(cont.clone(), seg.clone(), H, W, mode=mode)
        c2.check()
        assert np.array_equal(rec2.cpu().numpy(), rgb)
    finally:
        c2.close()


def test_malformed_segment_lengths_rejected(torch_mod, codecs):
    """a negative or oversized EARLIER seg_len entry must end in LLICTI_EFORMAT, not in a read outside the
    container buffer; a too-small in_stride is refused on the host."""
    from llicti_amd._lib import LlictiError, EFORMAT, EINVAL
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = make_batch("smooth", 2, 64, 96, seed0=61)
    for mode in (0, MODE_RANS(4)):
        cont, seg = c.encode(_dev(torch, rgb), mode=mode)
        c.check()
        for (k, v) in ((5, -100000), (4, 1 << 30), (6, -1), (3, -(1 << 31)), (10, (1 << 31) - 1), (0, 7)):
            bad = seg.clone()
            bad[1, k] = v
            c.decode(cont, bad, 64, 96, mode=mode)
            with pytest.raises(LlictiError) as e:
                c.check()
            assert e.value.code == EFORMAT, (mode, k, v)
        with pytest.raises(LlictiError) as e:
            c.decode(cont[:, :16].contiguous(), seg, 64, 96, mode=mode)
        assert e.value.code == EINVAL
        rec = _decode_poisoned(c, cont, seg, 64, 96, mode)        # the context is still usable
        assert np.array_equal(rec.cpu().numpy(), rgb)


def test_ac_decode_seam_ignores_bytes_past_len(torch_mod, codecs):
    """llicti_ac_decode_u16cdf honours d_len: garbage behind a stream must read as torchac's zero bits."""
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs("rand1337")
    rng = np.random.default_rng(11)
    Lp, N, S = 257, 500, 3
    stride = 264
    from test_ref_ac import random_rows
    cdfs = np.full((S, N, stride), 0xFFFF, np.uint16)
    syms = rng.integers(0, Lp - 1, (S, N)).astype(np.int16)
    streams = []
    for s in range(S):
        cdfs[s, :, :Lp] = random_rows(rng, N, Lp, 3.0)
        streams.append(orc.ac_encode_tables(cdfs[s, :, :Lp].copy(), syms[s]))
    in_stride = (max(len(x) for x in streams) + 3) // 4 * 4 + 32
    buf = rng.integers(0, 256, (S, in_stride)).astype(np.uint8)            # garbage everywhere ...
    for s, x in enumerate(streams):
        buf[s, :len(x)] = np.frombuffer(x, np.uint8)                        # ... except the stream itself
    lens = np.array([len(x) for x in streams], np.int32)
    dec = c.ac_decode(_dev(torch, cdfs.view(np.int16)), Lp, _dev(torch, buf), _dev(torch, lens), N)
    assert np.array_equal(dec.cpu().numpy(), syms)


@pytest.mark.parametrize("N", [1, 2, 3, 5])
def test_ac_decode_seam_short_streams(torch_mod, codecs, N):
    """Streams of a few bytes (1 .. 3 symbols give 1 .. 8 bytes): the decoder's first three words must read as zero bits past
    d_len[s] too, whatever the buffer holds there (torchac's get() semantics)."""
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs("rand1337")
    rng = np.random.default_rng(100 + N)
    Lp, S, stride = 257, 16, 264
    from test_ref_ac import random_rows
    cdfs = np.full((S, N, stride), 0xFFFF, np.uint16)
    syms = rng.integers(0, Lp - 1, (S, N)).astype(np.int16)
    streams = []
    for s in range(S):
        cdfs[s, :, :Lp] = random_rows(rng, N, Lp, 3.0)
        streams.append(orc.ac_encode_tables(cdfs[s, :, :Lp].copy(), syms[s]))
        assert np.array_equal(orc.ac_decode_tables(cdfs[s, :, :Lp].copy(), streams[-1], N), syms[s])
    assert min(len(x) for x in streams) <= 8
    in_stride = 64
    buf = np.full((S, in_stride), 0xFF, np.uint8)                           # all-ones garbage: the worst case for a zero-bit tail
    for s, x in enumerate(streams):
        buf[s, :len(x)] = np.frombuffer(x, np.uint8)
    lens = np.array([len(x) for x in streams], np.int32)
    dec = c.ac_decode(_dev(torch, cdfs.view(np.int16)), Lp, _dev(torch, buf), _dev(torch, lens), N)
    assert np.array_equal(dec.cpu().numpy(), syms)


def test_malformed_header_is_deterministic_and_per_image(torch_mod, codecs):
    """One image of a batch with a header that does not match the call: the call reports EFORMAT, llicti_image_status names
    the image, the OTHER images decode correctly, and the bad image's pixels do not depend on what the workspace held."""
    from llicti_amd._lib import LlictiError, EFORMAT
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    c = codecs("trainedlike")
    rgb = make_batch("smooth", 3, 64, 96, seed0=900)
    for mode in (0, MODE_RANS(2)):
        cont, seg = c.encode(_dev(torch, rgb), mode=mode)
        c.check()
        bad = cont.clone()
        bad[1, 1] ^= 0x01                                                   # h4 of image 1
        outs = []
        for poison in (0xA5, 0x3C):
            c.poison_workspace(poison)
            rec = c.decode(bad, seg, 64, 96, mode=mode)
            with pytest.raises(LlictiError) as e:
                c.check()
            assert e.value.code == EFORMAT
            st = c.image_status(3)
            assert st[0] == 0 and st[2] == 0 and st[1] == EFORMAT, st
            outs.append(rec.cpu().numpy())
        assert np.array_equal(outs[0][0], rgb[0]) and np.array_equal(outs[0][2], rgb[2])
        assert np.array_equal(outs[0][1], outs[1][1])                       # garbage, but the same garbage
        rec = _decode_poisoned(c, cont, seg, 64, 96, mode)
        assert np.array_equal(rec.cpu().numpy(), rgb) and not c.image_status(3).any()


def _reference_shaped_checkpoint(path, seed):
    """What agents/base.py:83-100 saves: epoch / iteration / best_valid_loss / state_dict / optimizer / scheduler / logger
    states; the state_dict carries compressai's extra buffers on every conditional_prob_model and NON-seed weights."""
    import torch
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    torch.manual_seed(seed)
    m = LLICTI(default_config())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed + 1)
    for k in sd:                                             # move the weights away from any default init
        if k.endswith(".weight") or k.endswith(".bias"):
            sd[k] = sd[k] * 1.3 + 0.01 * torch.randn(sd[k].shape, generator=g)
    for b in range(3):
        p = f"entropymodel.entmdls_scale_band.0.{b}.conditional_prob_model."
        sd[p + "_offset"] = torch.zeros(64, dtype=torch.int32)
        sd[p + "_quantized_cdf"] = torch.zeros((64, 40), dtype=torch.int32)
        sd[p + "_cdf_length"] = torch.zeros(64, dtype=torch.int32)
        sd[p + "scale_table"] = torch.linspace(0.11, 256, 64)
    ckpt = {"epoch": 123, "iteration": 45678, "best_valid_loss": 2.7, "state_dict": sd,
            "optimizer": {"state": {}, "param_groups": [{"lr": 1e-4}]}, "scheduler": {"best": 2.7},
            "train_logger": {}, "valid_logger": {}, "test_logger": {}}
    torch.save(ckpt, path)
    return sd


def test_agent_loads_reference_shaped_checkpoint(torch_mod, tmp_path):
    """a model_best.pth.tar as the reference writes it -> LLICTIAgent(mode='eval_model') -> the bytes the
    oracle produces for THOSE weights; a checkpoint whose keys do not match raises instead of silently keeping the init."""
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    from llicti_amd.weights import pack_state_dict
    from oracle import oracle as orc
    torch = torch_mod
    sd = _reference_shaped_checkpoint(tmp_path / "model_best.pth.tar", seed=77)
    cfg = default_config(test_data="synthetic:48x80x2", checkpoint_dir=str(tmp_path))
    agent = LLICTIAgent(cfg)
    own = agent.model.state_dict()
    for k in own:
        assert torch.equal(own[k].cpu(), sd[k]), k            # the checkpoint's values, not the seeded init
    W_o = orc.Weights(pack_state_dict({k: v for k, v in sd.items() if k in own}))
    res = agent.run()
    assert len(res) == 2 and all(r["max_abs_err"] < 1e-3 for r in res)
    for i in range(2):
        rgb = np.random.default_rng(i).integers(0, 256, size=(3, 48, 80), dtype=np.uint8)
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0")
        bl, _ = agent.model.compress(x)
        assert bl == orc.encode_image(rgb, W_o), i
    # cli --checkpoint takes the same file
    from llicti_amd import cli, fileio
    src, mid = tmp_path / "in.ppm", tmp_path / "x.llic"
    fileio.write_image(src, rgb)
    assert cli.main(["encode", str(src), str(mid), "--checkpoint", str(tmp_path / "model_best.pth.tar")]) == 0
    assert fileio.read_llic(mid) == orc.encode_image(rgb, W_o)
    # wrong keys: strict, like agents/base.py:60
    bad = torch.load(tmp_path / "model_best.pth.tar")
    bad["state_dict"] = {"module." + k: v for k, v in bad["state_dict"].items()}
    torch.save(bad, tmp_path / "model_best.pth.tar")
    with pytest.raises((KeyError, RuntimeError)):
        LLICTIAgent(cfg)


def _bench_container_mode(H=512, W=768):
    """The ENCODER mode bench.py times for images of H x W (bench.default_container: container "auto", a function of the image), as a codec mode."""
    import importlib.util
    import os
    from conftest import ROOT
    from llicti_amd.codec import mode_of_name
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    name = bench.default_container(H, W)
    return name, mode_of_name(name)


def test_full_size_oracle_parity(torch_mod, codecs, oracle_weights):
    """BASELINE.json full sizes inside the suite: the 24 x 768x512 batch (configs[2]) in the container bench.py TIMES
    (bench.default_container(24)) with 3 of the 24 images compared byte for byte with the oracle (it costs seconds per image)
    and the whole batch within the north star's 0.001 bpp of the reference-format container; 2 x 768x512 in the AC container
    (configs[1]'s shape, reference format) against the oracle, and configs[0]'s 256x256 random-RGB image in the AC
    container; every decode on a poisoned workspace."""
    from oracle import oracle as orc
    from llicti_amd.codec import container_to_bytestream_list
    torch = torch_mod
    c = codecs("rand1337")
    W_o = oracle_weights("rand1337")
    H, W = 512, 768
    name, mode = _bench_container_mode()
    rgb = np.stack([np.random.default_rng(i).integers(0, 256, size=(3, H, W), dtype=np.uint8) for i in range(24)])   # bench.py's batch
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    for b in (0, 11, 23):
        ref = _oracle_container(rgb[b], W_o, mode)
        assert container_to_bytestream_list(cont_h[b], seg_h[b]) == ref, (name, b)
    rec = _decode_poisoned(c, cont, seg, H, W, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)
    n_timed = int(seg_h.sum())
    cont_a, seg_a = c.encode(_dev(torch, rgb))
    c.check()
    dbpp = 8.0 * (n_timed - int(seg_a.sum().item())) / (24 * H * W)
    assert abs(dbpp) <= 0.001, f"timed container {name}: {dbpp:+.5f} bpp against the reference-format container"
    del cont_a, seg_a
    cont, seg = c.encode(_dev(torch, rgb[:2]))
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    for b in range(2):
        assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image(rgb[b], W_o), b
    rec = _decode_poisoned(c, cont, seg, H, W, 0)
    assert np.array_equal(rec.cpu().numpy(), rgb[:2])
    small = np.random.default_rng(0).integers(0, 256, size=(1, 3, 256, 256), dtype=np.uint8)                         # configs[0]
    cont, seg = c.encode(_dev(torch, small))
    c.check()
    bl = container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy())
    assert bl == orc.encode_image(small[0], W_o)
    assert np.array_equal(orc.decode_image(bl, W_o), small[0])
    rec = _decode_poisoned(c, cont, seg, 256, 256, 0)
    assert np.array_equal(rec.cpu().numpy(), small)


def test_configs4_per_gpu_batch_oracle_parity(torch_mod, codecs, oracle_weights):
    """BASELINE.json configs[4]: 256 images over 8 GPUs = 32 x 768x512 per GPU.  Rank 7's batch (bench.py seeds rank * B ...)
    in the timed container: one image against the oracle byte for byte, the batch lossless on a poisoned workspace."""
    from oracle import oracle as orc
    from llicti_amd.codec import container_to_bytestream_list
    torch = torch_mod
    c = codecs("rand1337")
    W_o = oracle_weights("rand1337")
    H, W, B, rank = 512, 768, 32, 7
    name, mode = _bench_container_mode()
    rgb = np.stack([np.random.default_rng(rank * B + i).integers(0, 256, size=(3, H, W), dtype=np.uint8) for i in range(B)])
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    b = 29
    ref = _oracle_container(rgb[b], W_o, mode)
    assert container_to_bytestream_list(cont[b].cpu().numpy(), seg[b].cpu().numpy()) == ref, name
    rec = _decode_poisoned(c, cont, seg, H, W, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)


def test_4k_image_oracle_parity(torch_mod, codecs, oracle_weights):
    """BASELINE.json configs[3] against the oracle: one 3840x2160 uniform-noise image (bench.py's image_4k leg) in the container that
    leg's HEADLINE uses (bench.IMAGE_4K_HEADLINE: the fastest mode within 0.001 bpp of the reference format -- 64 xwide streams), HIP bytes
    == oracle bytes (about a minute of CPU: the oracle evaluates 24.9 M symbols), lossless on a poisoned workspace, and inside the budget
    against the reference-format container of the same image (whose bytes are the oracle's too)."""
    import bench
    from oracle import oracle as orc
    from llicti_amd.codec import container_to_bytestream_list, mode_of_name
    torch = torch_mod
    c = codecs("rand1337")
    W_o = oracle_weights("rand1337")
    rgb = np.random.default_rng(0).integers(0, 256, size=(1, 3, 2160, 3840), dtype=np.uint8)
    mode = mode_of_name(bench.IMAGE_4K_HEADLINE)
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    got = container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy())
    ref = _oracle_container(rgb[0], W_o, mode)
    assert got == ref
    rec = _decode_poisoned(c, cont, seg, 2160, 3840, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)
    n_timed = int(seg.sum().item())
    cont_a, seg_a = c.encode(_dev(torch, rgb))
    c.check()
    dbpp = 8.0 * (n_timed - int(seg_a.sum().item())) / (2160 * 3840)
    assert abs(dbpp) <= 0.001, f"{bench.IMAGE_4K_HEADLINE}: {dbpp:+.6f} bpp against the reference-format container"


@pytest.mark.parametrize("wname,kind", [("rand1337", "noise"), ("trainedlike", "smooth")])
def test_ac_anchor_decoder_many_images_and_edges(torch_mod, codecs, oracle_weights, wname, kind):
    """The anchor form of the AC decoder where it is the default (B >= 96), on small images, plus the degenerate
    alphabets (Lp = 2, Lp = 512) with the form forced: pixels back exactly, and the streams are the oracle's."""
    from oracle import oracle as orc
    from llicti_amd.codec import container_to_bytestream_list
    torch = torch_mod
    c = codecs(wname)
    rgb = make_batch(kind, 100, 40, 56, seed0=500)
    cont, seg = c.encode(_dev(torch, rgb))
    c.check()
    rec = _decode_poisoned(c, cont, seg, 40, 56, 0)
    assert np.array_equal(rec.cpu().numpy(), rgb)
    W_o = oracle_weights(wname)
    for b in (0, 57, 99):
        assert container_to_bytestream_list(cont[b].cpu().numpy(), seg[b].cpu().numpy()) == orc.encode_image(rgb[b], W_o)
    edge = np.stack(list(_edge_images().values()))
    cont, seg = c.encode(_dev(torch, edge))
    c.check()
    c.set_tuning("ac_anchor_min_batch", 1)
    try:
        rec = _decode_poisoned(c, cont, seg, edge.shape[2], edge.shape[3], 0)
    finally:
        c.set_tuning("ac_anchor_min_batch", 96)
    assert np.array_equal(rec.cpu().numpy(), edge)


@pytest.mark.parametrize("wide", [0, 1, 2])
def test_rans_v3_integrity_check_detects_corruption(torch_mod, codecs, oracle_weights, wide):
    """rANS v3 ends with three known quantities: the main bit region is read to its last bit, and the tail coder (whose
    stream is what the 64 lane states are left with) returns to its start state 2^31 with no bit left.  A corrupted stream
    bit, final state or tail count ends, with overwhelming probability, in one of them being wrong -- both decoders (HIP,
    oracle) report it instead of returning wrong pixels silently; nothing crashes or hangs.  A v2 container (header byte 0
    without the format bit) is rejected deterministically."""
    from oracle import oracle as orc
    from llicti_amd._lib import LlictiError, EFORMAT
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list, header_dims, mode_of_header
    torch = torch_mod
    c = codecs("trainedlike")
    W_o = oracle_weights("trainedlike")
    rgb = make_batch("smooth", 2, 96, 128, seed0=700)
    mode = MODE_RANS(2, wide=wide)
    cont, seg = c.encode(_dev(torch, rgb), mode=mode)
    c.check()
    seg_h = seg.cpu().numpy()
    hdr = int(seg_h[1, :4].sum())
    s0 = int(seg_h[1, 4])
    pay = 248 << int(wide)                         # bytes of the final states: 64 / 128 / 256 lanes x 31 bits
    # a bit of the bit region | a final state | the tail count | the last byte of the bit region
    for where, val in ((hdr + 2 + 400, 0x10), (hdr + s0 - 100, 0x04), (hdr, 0x01), (hdr + s0 - pay - 1, 0xFF)):
        bad = cont.clone()
        bad[1, where] ^= val
        rec = c.decode(bad, seg, 96, 128, mode=mode)
        with pytest.raises(LlictiError) as e:
            c.check()
        assert e.value.code == EFORMAT, where
        assert np.array_equal(rec[0].cpu().numpy(), rgb[0])                           # the other image of the batch is untouched
        assert list(c.image_status(2)) == [0, EFORMAT]                               # ... and the status names the bad one
        bl = container_to_bytestream_list(bad[1].cpu().numpy(), seg_h[1])
        with pytest.raises(Exception):
            orc.decode_image_rans(bl, W_o)
    # the retired v2 tag (0x80 | lg2(M) << 4 | 5): the format bit clear
    v2 = cont.clone()
    v2[:, 0] = 0x95
    c.decode(v2, seg, 96, 128, mode=mode)
    with pytest.raises(LlictiError) as e:
        c.check()
    assert e.value.code == EFORMAT
    with pytest.raises(LlictiError):
        header_dims(bytes(v2[0, :17].cpu().numpy()))
    with pytest.raises(ValueError):
        mode_of_header(int(v2[0, 0]))
    if wide == 2:
        # the xwide tags of the v3 layout (rounds 4-5; 0xE9 named two xwide streams): retired -- refused by the library, the header helper and the oracle
        v3 = cont.clone()
        v3[:, 0] = 0xE9
        v3[:, 16] = v3[:, 16] & 0x03                       # (... whose pad field had no count in it)
        c.decode(v3, seg, 96, 128, mode=mode)
        with pytest.raises(LlictiError) as e:
            c.check()
        assert e.value.code == EFORMAT
        with pytest.raises(LlictiError):
            header_dims(bytes(v3[0, :17].cpu().numpy()))
        with pytest.raises(ValueError):
            mode_of_header(bytes(v3[0, :17].cpu().numpy()))
        with pytest.raises(Exception):
            orc.decode_image_rans(container_to_bytestream_list(v3[0].cpu().numpy(), seg_h[0]), W_o)
    rec = _decode_poisoned(c, cont, seg, 96, 128, mode)
    assert np.array_equal(rec.cpu().numpy(), rgb)


def test_two_contexts_concurrently_bitexact(torch_mod):
    """Two contexts on two HIP streams, encode of batch k next to decode of batch k - 1, with a third stream of unrelated GEMMs: the
    wavefronts of a coder workgroup drift apart under a neighbour's kernels, which is what exposes hand-offs between them that rely on
    lockstep (round 4: the encoder's LDS bit ring was one round too small and wrapped onto words still being flushed -- never in a test
    that had the chip to itself).  Every container must equal the one a quiet single-stream encode wrote, every decode its input, in the
    timed container and in a narrow and a wide one."""
    from llicti_amd.codec import HipCodec, mode_of_name
    torch = torch_mod
    dev = torch.device("cuda", 0)
    sd = load_state_dict("rand1337")
    B, H, W = 24, 512, 768
    rgb = [_dev(torch, make_batch("noise" if i % 2 == 0 else "smooth", B, H, W, seed0=900 + 40 * i)) for i in range(2)]
    ce, cd = HipCodec(dev), HipCodec(dev)
    try:
        ce.load_state_dict(sd)
        cd.load_state_dict(sd)
        s_e, s_d, s_x = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        a = torch.randn((2048, 2048), device=dev)
        for name in ("xrans16", "rans10", "wrans10"):
            mode = mode_of_name(name)
            ref = []
            for x in rgb:                                       # quiet reference: one stream, nothing else running
                c0, s0 = ce.encode(x, mode=mode)
                ce.check()
                ref.append((c0.clone(), s0.clone()))
            torch.cuda.synchronize()
            stride = ref[0][0].shape[1]
            cont = [torch.empty((B, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
            seg = [torch.zeros((B, 49), dtype=torch.int32, device=dev) for _ in range(2)]
            rec = [torch.empty_like(rgb[0]) for _ in range(2)]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            for k in range(8):
                with torch.cuda.stream(s_x):
                    for _ in range(6):
                        a = (a @ a).clamp_(-1, 1)               # neighbours on every compute unit
                with torch.cuda.stream(s_e):
                    ce.encode(rgb[k & 1], mode=mode, out=cont[k & 1], seg_len=seg[k & 1])
                    done[k & 1].record(s_e)
                if k > 0:
                    with torch.cuda.stream(s_d):
                        s_d.wait_event(done[(k - 1) & 1])
                        cd.decode(cont[(k - 1) & 1], seg[(k - 1) & 1], H, W, mode=mode, out=rec[(k - 1) & 1])
                torch.cuda.synchronize()                        # per iteration: the buffers are checked, then reused
                j = k & 1
                assert torch.equal(seg[j], ref[j][1]), (name, k)
                used = torch.arange(stride, device=dev)[None, :] < ref[j][1].sum(dim=1, keepdim=True)       # an image's own bytes; the rest of its stride is not written
                assert torch.equal(cont[j] * used, ref[j][0] * used), (name, k)
                if k > 0:
                    assert torch.equal(rec[(k - 1) & 1], rgb[(k - 1) & 1]), (name, k)
            with torch.cuda.stream(s_e):
                ce.check()
            with torch.cuda.stream(s_d):
                cd.check()
    finally:
        ce.close()
        cd.close()


def test_bench_line_contract(torch_mod):
    """bench.py's one JSON line carries every field the driver reads (small shape, no informational legs)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "3", "--height", "96", "--width", "128", "--steps", "2",
                        "--warmup", "1", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "rccl_ranks", "value_pcie_inclusive", "value_pcie_serial", "bpp",
              "cpu_baseline", "bpp_delta_vs_reference", "meets_north_star", "north_star_check"):
        assert k in d, k
    assert d["meets_north_star"] is False                       # not the north star's 768x512 shape: never claimed on another one
    assert d["coder"]["symbols_per_step"] > 0 and d["coder"]["decode_gsym_s"] > 0 and d["coder"]["encode_gsym_s"] > 0      # symbols/s of the coder
    assert abs(d["bpp_delta_vs_reference"]["timed_container_minus_reference_format_bpp"]) < 4.0      # 3 tiny images, 9 xwide streams each: mostly the 992-byte state blocks (12 kpixel images)
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["bitexact_vs_hip"] is True and cb["timed_container_bitexact_vs_hip"] is True
    tc = cb["torch_cpu"]                                         # the same path on plain PyTorch CPU ops, one image
    assert tc["value"] > 0 and tc["unit"] == "MPix/s" and tc["cores"] >= 1 and abs(tc["stream_bytes_minus_oracle"]) <= 64
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "MPix/s" and d["value"] > 0 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0 < r["whole_path_frac"] <= r["frac"] and set(r["kernel_ms"]) == {"encode", "decode"}
    assert {"cnn", "rans_stage", "rans_tail", "misc"} <= set(r["kernel_ms"]["decode"]) and {"cnn", "cdf_pairs", "rans_encode"} <= set(r["kernel_ms"]["encode"])
    assert len(r["cnn_tflops_per_level"]) == 5
    assert abs(d["ms_per_step"] * d["value"] - 3 * 96 * 128 / 1e3) < 0.02 * 3 * 96 * 128 / 1e3      # value = pixels / time


def _run_bench(args, timeout=900):
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_two_ranks_aggregate(torch_mod):
    """The N > 1 path of bench.py on the GPU: `--gpus 2` from a bare shell (the parent launches the ranks as children).  With two
    GPUs visible: RCCL, one rank per GPU.  On a one-GPU box: gloo with --allow-shared-gpu (a rehearsal, marked as such in the
    line; without the flag the run must refuse).  Either way `value` is SUM pixels / MAX time over the ranks and every rank
    leaves through destroy_process_group (a clean exit code)."""
    torch = torch_mod
    two = torch.cuda.device_count() >= 2
    common = ["--gpus", "2", "--batch", "2", "--height", "96", "--width", "128", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    if not two:
        p, d = _run_bench(common + ["--backend", "gloo"])
        assert p.returncode != 0 and d is None and "allow-shared-gpu" in p.stderr
    p, d = _run_bench(common + (["--backend", "nccl"] if two else ["--backend", "gloo", "--allow-shared-gpu"]))
    assert p.returncode == 0, p.stderr[-2000:]
    assert "destroy_process_group() was not called" not in p.stderr
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert d["backend"] == ("nccl" if two else "gloo") and d["shared_gpu"] is (not two) and d["distinct_devices"] == (2 if two else 1)
    assert abs(d["ms_per_step"] * d["value"] - 2 * 2 * 96 * 128 / 1e3) < 0.02 * 2 * 2 * 96 * 128 / 1e3      # both ranks' pixels over the slower rank's time
    assert d["meets_north_star"] is False and d["config"]["sharding"] == "images/2gpu"


def test_agent_two_ranks_on_gpu_equal_one_rank(torch_mod, tmp_path):
    """SURVEY section 8(e) through the API on the GPU: `LLICTIAgent.eval_model` started by torch.distributed.run as two ranks (RCCL with one rank
    per GPU when two are visible; on a one-GPU box gloo with both ranks on the one GPU -- a rehearsal of the plumbing, not scaling) against the same
    script started alone: rank 0's log -- the per-image lines of all nine images in index order and the rate table -- and the gathered records
    (sizes, bpsp, lossless check) are those of the one-rank run, one image at a time and batched."""
    import json
    import os
    import re
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    torch = torch_mod
    two = torch.cuda.device_count() >= 2
    script = os.path.join(ROOT, "tools", "agent_ranks_demo.py")

    def strip(t):
        t = re.sub(r"Enc/Dec-Times:[0-9.]+/[0-9.]+", "Enc/Dec-Times:T/T", t)
        return re.sub(r"\(\d\d:\d\d:\d\d\)", "(clock)", t)

    # ("auto", round 6: an image's container is a function of the image, so the sharded log equals the one-rank log on flat, natural-like and noise
    #  images alike -- round 5's content-following rule made an image's bytes depend on what its rank had coded before)
    for eb, cont in ((1, "xrans1"), (4, "xrans1"), (3, "auto")):
        one = tmp_path / f"one{eb}{cont}.json"
        p = subprocess.run([sys.executable, script, str(one), str(eb), cont], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        if not two:
            env["LLICTI_DIST_BACKEND"] = "gloo"
        out2 = tmp_path / f"two{eb}{cont}.json"
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), script, str(out2), str(eb), cont], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        a, b = json.load(open(one)), json.load(open(out2))
        assert a["world"] == 1 and b["world"] == 2 and b["own"] == [0, 2, 4, 6, 8]
        assert [r[:3] for r in b["records"]] == [r[:3] for r in a["records"]] and all(r[4] < 1e-3 for r in b["records"])     # (float32 uint8/255 on host vs device division)
        assert np.allclose([r[3] for r in b["records"]], [r[3] for r in a["records"]])
        pick = lambda t: [ln for ln in strip(t).splitlines() if ln.startswith("Agent|") and "bpsp=" in ln] + [strip(t)[strip(t).index("Rate Loss|"):]]      # noqa: E731
        assert pick(b["log"]) == pick(a["log"]), (eb, cont)


def test_plan_cache_eviction(torch_mod, codecs):
    """More distinct (B, H, W, mode) shapes through one context than it has seen before at once: every shape round-trips, including one seen
    before (round 5: the cache holds 32 plans, least recently used out first, on pooled table blocks -- no device synchronisation;
    test_many_sizes_no_device_sync_and_plan_reuse goes past 32)."""
    from llicti_amd.codec import MODE_RANS
    torch = torch_mod
    c = codecs("trainedlike")
    shapes = [(1 + (i % 2), 32 + 8 * i, 40 + 4 * i, MODE_RANS(2) if i % 3 else 0) for i in range(20)] + [(1, 32, 40, 0)]
    for (B, H, W, mode) in shapes:
        rgb = make_batch("smooth", B, H, W, seed0=900 + H)
        cont, seg = c.encode(_dev(torch, rgb), mode=mode)
        c.check()
        rec = _decode_poisoned(c, cont, seg, H, W, mode)
        assert np.array_equal(rec.cpu().numpy(), rgb), (B, H, W, mode)


def test_integration_md_binding_runs(torch_mod, tmp_path):
    """INTEGRATION.md section 1 shows the ctypes binding a maintainer of the reference would add.  The code block is
    executed as written (only the library path is substituted) against this repo's LLICTI module -- whose attribute tree
    is the reference's -- and must produce the same bytestream_list as the packaged path, and decode it."""
    import os
    import re
    from conftest import ROOT
    from llicti_amd import _lib
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    torch = torch_mod
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n# graphs/models/llicti_hip_binding.py.*?\n(.*?)```", md, re.S)
    assert m, "binding code block not found"
    code = m.group(1).replace('C.CDLL("libllicti_hip.so")', f'C.CDLL({_lib.SO_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#binding", "exec"), ns)
    torch.manual_seed(1337)
    model = LLICTI(default_config()).to("cuda:0").eval()
    hip = ns["HipPath"](model, 0)
    rgb = make_image("smooth", 72, 104, 5)
    x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0")
    bl = hip.compress(x)
    ref, _ = model.compress(x)
    assert bl == ref
    x_reco = hip.decompres(bl, torch.device("cuda:0"))
    assert np.array_equal((x_reco * 255).round().to(torch.uint8).cpu().numpy()[0], rgb)
    # section 2 (the torchac seam) and section 3 (LLICTI.forward) in the same namespace, as a maintainer would paste them
    blocks = re.findall(r"```python\n(.*?)```", md, re.S)
    seam = next(b for b in blocks if "def encode_int16_normalized_cdf_hip" in b)
    fwd = next(b for b in blocks if "def forward_hip" in b)
    exec(compile(seam, "INTEGRATION.md#seam", "exec"), ns)
    exec(compile(fwd, "INTEGRATION.md#forward", "exec"), ns)
    import ref_ac
    g = load_case("smooth_64x48_tl")
    rows, idx = g["cdfrows_s1_b0_c0"], g["cdfidx_s1_b0_c0"]
    sym = g["sym_s1_b0_c0"].ravel()[idx].astype(np.int16)
    cdf_t = torch.from_numpy(rows.view(np.int16).copy()).to("cuda:0").reshape(1, 1, 1, len(sym), rows.shape[1])
    sym_t = torch.from_numpy(sym).to("cuda:0").reshape(1, 1, 1, len(sym))
    got = ns["encode_int16_normalized_cdf_hip"](hip.ctx, cdf_t, sym_t)
    assert got == ref_ac.encode(rows.tolist(), sym.tolist())
    pad = make_image("smooth", 64, 96, 6)
    xp = torch.from_numpy(pad.astype(np.float32) / np.float32(255)).unsqueeze(0).to("cuda:0")
    out = ns["forward_hip"](hip.ctx, xp)
    mine = model.forward(xp)
    assert len(out) == 5 and all(torch.equal(a, b) for a, b in zip(out, mine))
    # section 4: a batch of mixed sizes in one call (llicti_encode_images_v / llicti_decode_images_v); each image's bytes == the packaged path's
    mixed = next(b for b in blocks if "def compress_mixed_hip" in b)
    exec(compile(mixed, "INTEGRATION.md#mixed", "exec"), ns)
    imgs = [make_image("smooth", h, w, 30 + i) for i, (h, w) in enumerate([(72, 104), (97, 130), (64, 200)])]
    cont, seg = ns["compress_mixed_hip"](hip.ctx, [torch.from_numpy(a).to("cuda:0") for a in imgs], 0x500 | 2)
    m2 = LLICTI(default_config(container="xrans2")).to("cuda:0").eval()
    m2.load_state_dict(model.state_dict())
    from llicti_amd.codec import container_to_bytestream_list
    for b, a in enumerate(imgs):
        want, _ = m2.compress(torch.from_numpy(a[None]).to("cuda:0"))
        assert container_to_bytestream_list(cont[b].cpu().numpy(), seg[b].cpu().numpy()) == want, b


# ------------------------------------------------------------------------------------------------ batches of mixed sizes
def _flat(rgbs):
    return np.concatenate([r.reshape(-1) for r in rgbs])


def _split(flat, Hs, Ws):
    out, pos = [], 0
    for h, w in zip(Hs, Ws):
        out.append(flat[pos:pos + 3 * h * w].reshape(3, h, w))
        pos += 3 * h * w
    return out


MIXED_SHAPES = [(150, 131), (96, 160), (67, 93), (150, 131), (33, 64), (224, 96), (32, 32), (97, 351)]      # odd sizes, both pad flags, a repeat, portrait and landscape


@pytest.mark.parametrize("mode_name", ["xrans3", "xrans10", "wrans4", "rans8", "rans64"])
@pytest.mark.parametrize("wname", ["trainedlike", "rand1337"])
def test_mixed_size_batch_equals_single_image_and_oracle(torch_mod, codecs, oracle_weights, mode_name, wname):
    """llicti_encode_images_v / llicti_decode_images_v (the reference's test loader yields a different H x W at almost every step,
    dataloaders/image_dl.py:40-45; agents/llicti_agent.py:122-164 codes them one at a time): every image of a mixed-size batch has exactly the
    bytes of its own single-image encode, three of them are compared with the CPU oracle, and the batch decodes losslessly on a poisoned
    workspace -- in every kind of rANS stream (64 / 128 / 256 lanes, grouped segments)."""
    from llicti_amd.codec import container_to_bytestream_list, mode_of_name, _mode_wide
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs(wname)
    mode = mode_of_name(mode_name)
    kinds = ["smooth", "noise"]
    rgbs = [make_image(kinds[i % 2], h, w, 700 + i) for i, (h, w) in enumerate(MIXED_SHAPES)]
    Hs, Ws = [h for h, _ in MIXED_SHAPES], [w for _, w in MIXED_SHAPES]
    flat = _dev(torch, _flat(rgbs))
    cont, seg = c.encode_v(flat, Hs, Ws, mode)
    c.check()
    cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
    W_o = oracle_weights(wname)
    for b, rgb in enumerate(rgbs):
        c1, s1 = c.encode(_dev(torch, rgb[None]), mode=mode)
        c.check()
        n = int(seg_h[b].sum())
        assert np.array_equal(seg_h[b], s1[0].cpu().numpy()), b
        assert np.array_equal(cont_h[b, :n], c1[0, :n].cpu().numpy()), b
        if b in (0, 2, 7):
            assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image_rans(rgb, W_o, mode & 0xFF, wide=_mode_wide(mode)), b
    c.poison_workspace()
    rec = c.decode_v(cont, seg, Hs, Ws, mode)
    c.check()
    assert not c.image_status(len(rgbs)).any()
    for b, (r, rgb) in enumerate(zip(_split(rec.cpu().numpy(), Hs, Ws), rgbs)):
        assert np.array_equal(r, rgb), b
    # ... and a container written by a mixed-size call decodes in a call of its own (and the other way round)
    one = c.decode(cont[5:6].contiguous(), seg[5:6].contiguous(), Hs[5], Ws[5], mode=mode)
    c.check()
    assert np.array_equal(one.cpu().numpy()[0], rgbs[5])


def test_mixed_size_batch_eval_set_shapes(torch_mod, codecs, oracle_weights):
    """A batch at the sizes the reference's own test set has (tests/golden/eval_shapes.json: the 500 sizes its eval log prints, 119 distinct,
    interleaved): 12 consecutive images of it in xrans8 -- bytes of image 0, 5 and 11 equal the oracle's, the rest equal their single-image
    encodes, lossless on a poisoned workspace; then the next 12 through the same context (a new plan: no stale tables)."""
    import json
    import os
    from conftest import GOLDEN
    from llicti_amd.codec import container_to_bytestream_list, mode_of_name
    from oracle import oracle as orc
    torch = torch_mod
    shapes = json.load(open(os.path.join(GOLDEN, "eval_shapes.json")))["shapes"]
    c = codecs("trainedlike")
    W_o = oracle_weights("trainedlike")
    mode = mode_of_name("xrans8")
    for k0 in (404, 416):
        sh = shapes[k0:k0 + 12]
        Hs, Ws = [h for h, _ in sh], [w for _, w in sh]
        rgbs = [make_image("smooth" if i % 3 else "noise", h, w, 900 + k0 + i) for i, (h, w) in enumerate(sh)]
        cont, seg = c.encode_v(_dev(torch, _flat(rgbs)), Hs, Ws, mode)
        c.check()
        cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
        for b, rgb in enumerate(rgbs):
            if b in (0, 5, 11) and k0 == 404:
                assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image_rans(rgb, W_o, 8, wide=2), b
            else:
                c1, s1 = c.encode(_dev(torch, rgb[None]), mode=mode)
                n = int(seg_h[b].sum())
                assert np.array_equal(seg_h[b], s1[0].cpu().numpy()) and np.array_equal(cont_h[b, :n], c1[0, :n].cpu().numpy()), b
        c.poison_workspace()
        rec = c.decode_v(cont, seg, Hs, Ws, mode)
        c.check()
        for b, (r, rgb) in enumerate(zip(_split(rec.cpu().numpy(), Hs, Ws), rgbs)):
            assert np.array_equal(r, rgb), (k0, b)


def test_mixed_size_batch_rejects_reference_format_and_bad_sizes(torch_mod, codecs):
    """The reference-format container codes equal sizes per call (its decoder's chunk pipeline is sized per stage): a mixed-size call in it
    is LLICTI_EINVAL, not a wrong result; so is a size outside 32 .. 8160; a decode whose sizes do not match the headers flags those images."""
    from llicti_amd import _lib
    from llicti_amd.codec import MODE_AC, mode_of_name
    torch = torch_mod
    c = codecs("trainedlike")
    rgbs = [make_image("smooth", 64, 96, 1), make_image("smooth", 96, 64, 2)]
    flat = _dev(torch, _flat(rgbs))
    with pytest.raises(_lib.LlictiError) as e:
        c.encode_v(flat, [64, 96], [96, 64], MODE_AC)
    assert e.value.code == _lib.EINVAL
    with pytest.raises(_lib.LlictiError):
        c.encode_v(flat, [64, 16], [96, 64], mode_of_name("xrans2"))
    # equal sizes through the _v entry are fine in the reference format (same bytes as encode())
    two = [make_image("smooth", 64, 96, 1), make_image("noise", 64, 96, 2)]
    cv, sv = c.encode_v(_dev(torch, _flat(two)), [64, 64], [96, 96], MODE_AC)
    cu, su = c.encode(_dev(torch, np.stack(two)), mode=MODE_AC)
    c.check()
    assert torch.equal(sv, su)
    for b in range(2):
        n = int(su[b].sum())
        assert torch.equal(cv[b, :n], cu[b, :n])
    # swapped sizes at decode: both images flagged, no fault
    mode = mode_of_name("xrans2")
    cont, seg = c.encode_v(flat, [64, 96], [96, 64], mode)
    c.check()
    c.decode_v(cont, seg, [96, 64], [64, 96], mode)
    with pytest.raises(_lib.LlictiError) as e:
        c.check()
    assert e.value.code == _lib.EFORMAT
    assert (c.image_status(2) == _lib.EFORMAT).all()


def test_many_sizes_no_device_sync_and_plan_reuse(torch_mod, codecs):
    """80 different (batch composition, size) plans through one context -- more than the plan cache holds (32) -- round-trip losslessly, a
    size that left the cache comes back with the same bytes, and the calls never block on the device once the table blocks are pooled: the
    host returns from an encode enqueued behind 50 ms of other work on the stream long before that work ends."""
    import time
    from llicti_amd.codec import mode_of_name
    torch = torch_mod
    c = codecs("trainedlike")
    mode = mode_of_name("xrans2")
    first = None
    for k in range(80):
        h, w = 32 + 3 * k, 40 + 2 * (k % 17)
        rgb = make_image("smooth", h, w, k)[None]
        cont, seg = c.encode(_dev(torch, rgb), mode=mode)
        rec = c.decode(cont, seg, h, w, mode=mode)
        assert np.array_equal(rec.cpu().numpy(), rgb), k
        if k == 0:
            first = (cont[0, :int(seg[0].sum())].cpu().numpy().copy(), rgb)
    c.check()
    cont, seg = c.encode(_dev(torch, first[1]), mode=mode)
    assert np.array_equal(cont[0, :int(seg[0].sum())].cpu().numpy(), first[0])
    # the library's own account: from here on ever new sizes cost plan builds and nothing else -- no device synchronisation, no allocation
    before = {k: c.counter(k) for k in ("device_syncs", "device_allocs", "plan_builds", "block_waits")}
    for k in range(60):
        sh = [(40 + 2 * k + 3 * b, 50 + k + 5 * b) for b in range(3)]
        rgbs = [make_image("noise", h, w, 10 * k + b) for b, (h, w) in enumerate(sh)]
        cont, seg = c.encode_v(_dev(torch, _flat(rgbs)), [h for h, _ in sh], [w for _, w in sh], mode)
        rec = c.decode_v(cont, seg, [h for h, _ in sh], [w for _, w in sh], mode)
    c.check()
    assert np.array_equal(rec.cpu().numpy(), _flat(rgbs))
    after = {k: c.counter(k) for k in before}
    assert after["plan_builds"] == before["plan_builds"] + 60 and c.counter("plans_cached") <= 32
    assert after["device_syncs"] == before["device_syncs"] and after["device_allocs"] == before["device_allocs"], (before, after)
    # a new size enqueued behind a long-running kernel: the call must return while that kernel is still running
    a = torch.randn(8192, 8192, device="cuda:0")
    x = _dev(torch, make_image("smooth", 333, 77, 5)[None])
    torch.cuda.synchronize()
    for _ in range(6):
        a = a @ a * 1e-4
    t1 = time.perf_counter()
    cont, seg = c.encode(x, mode=mode)
    rec = c.decode(cont, seg, 333, 77, mode=mode)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    assert (t3 - t2) > 4 * (t2 - t1), f"the calls blocked on the device: enqueue {t2 - t1:.4f} s, remaining GPU work {t3 - t2:.4f} s"
    assert np.array_equal(rec.cpu().numpy()[0], make_image("smooth", 333, 77, 5))


@pytest.mark.parametrize("name", ["smooth11_trainedlike", "noise0_rand1337"])
def test_full_size_params_and_tables_vs_reference(torch_mod, codecs, name):
    """levels 1 and 0 of FULL-SIZE 768x512 images on the HIP kernels against the reference's own numbers
    (tests/golden/fullsize_samples.npz: LLICTIEntropyModel4.get_params outputs and int16 table entries at ~160 positions per (level, band)
    -- image corners and borders, both sides of the MFMA kernel's tile seams, interior -- generated from the reference-owned code by
    tests/golden/make_fixture_fullsize_samples.py): interpolator outputs within 1e-5 (north star), table entries within +-1 (trained-like
    weights) / +-40 (sigma-floor seed-1337 weights) counts, symbols exact."""
    import json
    import os
    from conftest import GOLDEN
    torch = torch_mod
    z = np.load(os.path.join(GOLDEN, "fullsize_samples.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    m = meta[name]
    c = codecs(m["weights"])
    rgb = make_image(m["kind"], m["H"], m["W"], m["seed"])
    planes, fplanes, mm = c.lift(_dev(torch, rgb[None]))
    mm_h = mm[0].cpu().numpy()
    planes_h = planes[0].cpu().numpy()
    ent_tol = 1 if m["weights"] == "trainedlike" else 40
    worst = 0.0
    for lvl in (1, 0):
        for band in range(3):
            tag = f"{name}_l{lvl}_b{band}"
            pos = z[tag + "_pos"].astype(np.int64)
            p64 = c.band_params(fplanes, lvl, band)
            got = c.params60(p64)[0].cpu().numpy()[pos[:, 0], pos[:, 1]]
            err = float(np.abs(got - z[tag + "_params"]).max())
            worst = max(worst, err)
            assert err < PARAM_TOL, (tag, err)
            oi, oj = [(1, 1), (0, 1), (1, 0)][band]
            w_band = got.shape and p64.shape[3]
            for clr in range(3):
                tab = c.cdf_tables(planes, p64, mm, lvl, band, clr, row_stride=512)[0].cpu().numpy().view(np.uint16)     # [hc * wc, 512]
                shift = 127 if clr == 0 else -int(mm_h[clr - 1])
                Lp = meta[f"{tag}_c{clr}_Lp"]
                sym = planes_h[clr][((2 * pos[:, 0] + oi) << lvl), ((2 * pos[:, 1] + oj) << lvl)].astype(np.int64) + shift
                assert np.array_equal(sym, z[f"{tag}_c{clr}_sym"].astype(np.int64)), (tag, clr)
                rows = tab[pos[:, 0] * w_band + pos[:, 1]]                      # 768x512: the coded crop is the band grid
                idx = z[f"{tag}_c{clr}_idx"].astype(np.int64)
                val = z[f"{tag}_c{clr}_val"].astype(np.int64)
                keep = idx < Lp - 1                                            # (the last entry wraps to 0 and is ignored by the coder)
                d = np.abs(np.take_along_axis(rows.astype(np.int64), idx, axis=1) - val)[keep]
                assert d.max() <= ent_tol, (tag, clr, int(d.max()))
    print(f"{name}: max |params - reference| = {worst:.2e}")


def test_full_size_ragged_vs_reference(torch_mod, codecs, oracle_weights):
    """the ragged path at full size against THE REFERENCE.  tests/golden/fullsize_samples_ragged.npz holds the reference's own get_params
    outputs, coded symbols and int16 table entries at ~160 positions per (level <= 1, band) of a 577x768 image of its eval set's odd kind (every
    level's height odd: lazyDWT's bottom-row pad at every level, LLICTI_nets.py:226-240; bands x11 / x10 code one row less than the band grid,
    :396-397; the decoder re-pads, :511-530) -- the padded last row and the corners among them.  (1) The kernel-level entry points on that image:
    parameters within 1e-5, table entries within +-1, symbols exact.  (2) The image inside a batch of MIXED sizes through llicti_encode_images_v /
    llicti_decode_images_v -- tile lists, per-image geometry, the RAGGED band-CNN instantiation and its odd-edge staging: its container is the
    oracle's byte for byte (so every coded symbol and the two table entries behind it are the oracle's, which test_full_size_ragged_samples_vs_reference
    holds to the same fixture), and the CNN outputs of the encoder's and the decoder's last launch (level 0, band x10), read out of the workspace, are
    within 1e-5 of the reference's at the fixture's positions and BIT-EQUAL to the equal-size kernel's."""
    from oracle import oracle as orc
    from test_oracle_golden import _ragged_samples, check_samples_against
    from llicti_amd.codec import MODE_RANS, container_to_bytestream_list
    torch = torch_mod
    z, meta = _ragged_samples()
    name = "smooth13_trainedlike_577x768"
    m = meta[name]
    H, W = m["H"], m["W"]
    c = codecs(m["weights"])
    W_o = oracle_weights(m["weights"])
    rgb = make_image(m["kind"], H, W, m["seed"])
    planes, fplanes, mm = c.lift(_dev(torch, rgb[None]))
    mm_h = mm[0].cpu().numpy()
    mm6 = [0, int(mm_h[0]), int(mm_h[1]), 255, int(mm_h[2]), int(mm_h[3])]
    planes_h = planes[0].cpu().numpy()
    p64, tabs, wcs = {}, {}, {}

    def params_of(lvl, band):
        if (lvl, band) not in p64:
            p64[(lvl, band)] = c.band_params(fplanes, lvl, band)
        return c.params60(p64[(lvl, band)])[0].cpu().numpy()
    pcache = {}

    def params_cached(lvl, band):
        if (lvl, band) not in pcache:
            pcache[(lvl, band)] = params_of(lvl, band)
        return pcache[(lvl, band)]

    def row_of(lvl, band, clr, i, j):
        if (lvl, band, clr) not in tabs:
            params_cached(lvl, band)
            tabs[(lvl, band, clr)] = c.cdf_tables(planes, p64[(lvl, band)], mm, lvl, band, clr, row_stride=512)[0].cpu().numpy().view(np.uint16)
            wcs[(lvl, band, clr)] = meta[f"{name}_l{lvl}_b{band}_c{clr}_crop"][1]
        return tabs[(lvl, band, clr)][i * wcs[(lvl, band, clr)] + j]
    check_samples_against(z, meta, name, tuple((l, b) for l in (1, 0) for b in range(3)), planes_h, mm6, params_cached, row_of, ent_tol=1)
    # (2) the same image in a batch of mixed sizes
    rgbs = [make_image("smooth", 150, 131, 1), rgb, make_image("noise", 321, 481, 2)]
    Hs, Ws = [r.shape[1] for r in rgbs], [r.shape[2] for r in rgbs]
    modes = [MODE_RANS(2, wide=2), MODE_RANS(16, wide=2), MODE_RANS(6, wide=2)]
    cont, seg = c.encode_v(_dev(torch, _flat(rgbs)), Hs, Ws, modes)
    c.check()
    assert container_to_bytestream_list(cont[1].cpu().numpy(), seg[1].cpu().numpy()) == orc.encode_image_rans(rgb, W_o, 16, 2)
    tag = f"{name}_l0_b2"
    pos = z[tag + "_pos"].astype(np.int64)
    h0, w0 = params_cached(0, 2).shape[:2]
    want_bits = p64[(0, 2)][0].reshape(64, h0 * w0)                           # the equal-size kernel's planes of this image
    for what in ("encode", "decode"):
        if what == "decode":
            c.poison_workspace()
            rec = c.decode_v(cont, seg, Hs, Ws, modes)
            c.check()
            for r, im in zip(_split(rec.cpu().numpy(), Hs, Ws), rgbs):
                assert np.array_equal(r, im)
        got = c.last_params_v(Hs, Ws, modes, 1)                                # [64, h * w] of the mixed-size launch
        used = [p for p in range(64) if p % 16 != 15]                          # (plane 15 of a head does not exist: never written)
        assert torch.equal(got[used], want_bits[used]), what
        p60 = c.params60(got.view(1, 64, h0, w0))[0].cpu().numpy()[pos[:, 0], pos[:, 1]]
        assert np.abs(p60 - z[tag + "_params"]).max() < PARAM_TOL, (what, np.abs(p60 - z[tag + "_params"]).max())


@pytest.mark.parametrize("kind", ["sharp", "single", "flat"])
def test_rans_xwide_long_tail_bitexact(torch_mod, kind):
    """xwide tails longer than 2,047 symbols (cheap sources: v4's tails run to 8,160 symbols in multiples of 32, one zero-start chain; see
    tests/test_oracle_golden.py::test_rans_xwide_long_tail) -- HIP bytes == oracle bytes, lossless on a poisoned workspace, alone, as a
    batch, and inside a batch of mixed sizes next to ordinary images; a flat image (symbols that cost nothing: tails at the cap) too."""
    from helpers import xwide_stream_header
    from test_oracle_golden import _cheap_case
    from llicti_amd.codec import HipCodec, container_to_bytestream_list, mode_of_name
    from oracle import oracle as orc
    torch = torch_mod
    sd, W_o, img = _cheap_case("sharp" if kind == "flat" else kind)
    if kind == "flat":
        img = np.full((3, 256, 384), 77, np.uint8)
    c = HipCodec("cuda:0")
    try:
        c.load_state_dict(sd)
        for name, M in (("xrans1", 1), ("xrans2", 2), ("xrans5", 5)):
            mode = mode_of_name(name)
            x = _dev(torch, np.stack([img, img[:, ::-1].copy()]))
            cont, seg = c.encode(x, mode=mode)
            c.check()
            want = orc.encode_image_rans(img, W_o, M, 2)
            got = container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy())
            assert got == want, (kind, name)
            Tf = [32 * xwide_stream_header(s)[0] for s in got[1][:M]]     # the header field: T / 32 rounded up
            if name == "xrans1":
                assert Tf[0] >= 2048, Tf
                if kind != "sharp":
                    assert Tf[0] >= 4096, Tf                           # (v3 needed its escape here; flat: the cap of 8,160)
            rec = _decode_poisoned(c, cont, seg, img.shape[1], img.shape[2], mode)
            assert np.array_equal(rec.cpu().numpy(), x.cpu().numpy())
        mode = mode_of_name("xrans2")
        rgbs = [make_image("smooth", 150, 131, 1), img, make_image("noise", 67, 93, 2)]
        Hs, Ws = [r.shape[1] for r in rgbs], [r.shape[2] for r in rgbs]
        cont, seg = c.encode_v(_dev(torch, _flat(rgbs)), Hs, Ws, mode)
        c.check()
        assert container_to_bytestream_list(cont[1].cpu().numpy(), seg[1].cpu().numpy()) == orc.encode_image_rans(img, W_o, 2, 2)
        c.poison_workspace()
        rec = c.decode_v(cont, seg, Hs, Ws, mode)
        c.check()
        for r, rgb in zip(_split(rec.cpu().numpy(), Hs, Ws), rgbs):
            assert np.array_equal(r, rgb)
    finally:
        c.close()


@pytest.mark.parametrize("kind", ["sharp", "single"])
def test_tail_speculated_window_all_stream_kinds(torch_mod, kind):
    """Round 5: the tail decoder looks at a window its preparing wavefronts speculated on (around the mixture's median) before it searches.  On
    cheap, model-drawn content the window is offered for nearly every symbol and most slots fall inside it ("single": all but the rare tail
    symbol; "sharp": a few per cent miss and take the bucket search behind an offered window) -- every stream kind (64 / 128 / 256 lanes, one
    tail chain or two, the pooled wavefronts of a one-chain xwide stream) decodes to the original on a poisoned workspace, and the bytes it
    decodes are the oracle's."""
    from test_oracle_golden import _cheap_case
    from llicti_amd.codec import HipCodec, container_to_bytestream_list, mode_of_name
    from oracle import oracle as orc
    torch = torch_mod
    sd, W_o, img = _cheap_case(kind)
    c = HipCodec("cuda:0")
    try:
        c.load_state_dict(sd)
        x = _dev(torch, np.stack([img, img[:, :, ::-1].copy(), img[:, ::-1].copy()]))
        for name, M, wide in (("rans1", 1, 0), ("rans4", 4, 0), ("wrans1", 1, 1), ("wrans3", 3, 1), ("xrans1", 1, 2), ("xrans3", 3, 2)):
            mode = mode_of_name(name)
            cont, seg = c.encode(x, mode=mode)
            c.check()
            assert container_to_bytestream_list(cont[0].cpu().numpy(), seg[0].cpu().numpy()) == orc.encode_image_rans(img, W_o, M, wide), (kind, name)
            rec = _decode_poisoned(c, cont, seg, img.shape[1], img.shape[2], mode)
            assert np.array_equal(rec.cpu().numpy(), x.cpu().numpy()), (kind, name)
    finally:
        c.close()


def test_mixed_size_batch_caller_placed_rgb(torch_mod, codecs):
    """llicti_encode_images_v / llicti_decode_images_v with caller-chosen byte offsets of the images in the RGB buffer (`rgb_off`: gaps between the
    images, unaligned starts, a different order in memory than in the call): the containers are those of the tightly packed call, the decoder
    writes every image to its own place and nothing between them; equal sizes at caller-chosen offsets take the table path too."""
    import ctypes as C
    from llicti_amd import _lib
    from llicti_amd.codec import NSEG, _ptr, _stream_ptr, mode_of_name
    torch = torch_mod
    c = codecs("trainedlike")
    mode = mode_of_name("xrans3")
    for shapes in ([(150, 131), (67, 93), (96, 160), (33, 65)], [(96, 160)] * 3):
        rgbs = [make_image("smooth" if i % 2 else "noise", h, w, 40 + i) for i, (h, w) in enumerate(shapes)]
        Hs, Ws = np.array([h for h, _ in shapes], np.int32), np.array([w for _, w in shapes], np.int32)
        B = len(rgbs)
        cont0, seg0 = c.encode_v(_dev(torch, _flat(rgbs)), Hs, Ws, mode)
        c.check()
        # placement: image b at a caller-chosen offset -- reversed order in memory, odd gaps (not multiples of 4: the lift falls back to byte accesses)
        sizes = [r.size for r in rgbs]
        offs, pos = [0] * B, 13
        for b in reversed(range(B)):
            offs[b] = pos
            pos += sizes[b] + 7 + 3 * b
        total = pos + 5
        host = np.full(total, 0xEE, np.uint8)
        for b, r in enumerate(rgbs):
            host[offs[b]:offs[b] + sizes[b]] = r.reshape(-1)
        buf = _dev(torch, host)
        off_arr = np.array(offs, dtype=np.uint64)
        ws = c.workspace_v(Hs, Ws, mode)
        stride = cont0.shape[1]
        cont = torch.empty((B, stride), dtype=torch.uint8, device="cuda:0")
        seg = torch.zeros((B, NSEG), dtype=torch.int32, device="cuda:0")
        _lib.check(c.L.llicti_encode_images_v(c.ctx, _ptr(buf), _ptr(off_arr), B, _ptr(Hs), _ptr(Ws), mode, _ptr(ws), ws.numel(), _ptr(cont), stride,
                                              _ptr(seg), _stream_ptr(c.device)))
        c.check()
        assert torch.equal(seg, seg0)
        for b in range(B):
            n = int(seg0[b].sum())
            assert torch.equal(cont[b, :n], cont0[b, :n]), b
        out = torch.full((total,), 0x55, dtype=torch.uint8, device="cuda:0")
        c.poison_workspace()
        _lib.check(c.L.llicti_decode_images_v(c.ctx, _ptr(cont), stride, _ptr(seg), B, _ptr(Hs), _ptr(Ws), mode, _ptr(ws), ws.numel(), _ptr(out),
                                              _ptr(off_arr), _stream_ptr(c.device)))
        c.check()
        got = out.cpu().numpy()
        want = np.full(total, 0x55, np.uint8)
        for b, r in enumerate(rgbs):
            want[offs[b]:offs[b] + sizes[b]] = r.reshape(-1)
        assert np.array_equal(got, want)                      # every image at its place, the gaps untouched


def test_mixed_size_entry_points_reject_misuse(torch_mod, codecs):
    """llicti_encode_images_v / llicti_decode_images_v: a workspace or an output stride smaller than the batch needs is LLICTI_ENOSPACE (never a
    write past the buffer), null size arrays / B = 0 / an unknown mode are LLICTI_EINVAL, a container whose header names another size than the
    call flags that image only -- and the context keeps working afterwards."""
    from llicti_amd import _lib
    from llicti_amd.codec import NSEG, _ptr, _stream_ptr, mode_of_name
    torch = torch_mod
    c = codecs("trainedlike")
    mode = mode_of_name("xrans2")
    shapes = [(96, 160), (150, 131), (67, 93)]
    rgbs = [make_image("smooth", h, w, 60 + i) for i, (h, w) in enumerate(shapes)]
    Hs, Ws = np.array([h for h, _ in shapes], np.int32), np.array([w for _, w in shapes], np.int32)
    flat = _dev(torch, _flat(rgbs))
    ws = c.workspace_v(Hs, Ws, mode)
    need = int(c.L.llicti_workspace_bytes_v(3, _ptr(Hs), _ptr(Ws), mode))
    stride = max(c.max_container_bytes(h, w) for h, w in shapes)
    cont = torch.empty((3, stride), dtype=torch.uint8, device="cuda:0")
    seg = torch.zeros((3, NSEG), dtype=torch.int32, device="cuda:0")
    st = _stream_ptr(c.device)

    def enc(B=3, hs=Hs, wsz=Ws, m=mode, wbytes=None, ostride=stride):
        return c.L.llicti_encode_images_v(c.ctx, _ptr(flat), None, B, _ptr(hs) if hs is not None else None, _ptr(wsz) if wsz is not None else None, m,
                                          _ptr(ws), need if wbytes is None else wbytes, _ptr(cont), ostride, _ptr(seg), st)
    assert enc(wbytes=need - 256) == _lib.ENOSPACE
    assert enc(ostride=4096) == _lib.ENOSPACE                                   # smaller than any image's container
    assert enc(B=0) == _lib.EINVAL and enc(hs=None) == _lib.EINVAL and enc(m=0x700 | 3) == _lib.EINVAL
    bad = Hs.copy()
    bad[1] = 9000
    assert enc(hs=bad) == _lib.EINVAL
    assert enc() == 0
    c.check()
    # decode with image 1 declared as another size: that image is flagged, the others decode
    H2, W2 = Hs.copy(), Ws.copy()
    H2[1], W2[1] = 131, 150
    offs, total = c.flat_offsets(H2, W2)
    out = torch.empty((total,), dtype=torch.uint8, device="cuda:0")
    wsb = c.workspace_v(H2, W2, mode)
    _lib.check(c.L.llicti_decode_images_v(c.ctx, _ptr(cont), stride, _ptr(seg), 3, _ptr(H2), _ptr(W2), mode, _ptr(wsb), wsb.numel(), _ptr(out), None, st))
    with pytest.raises(_lib.LlictiError):
        c.check()
    stat = c.image_status(3)
    assert stat[1] == _lib.EFORMAT and stat[0] == 0 and stat[2] == 0
    got = _split(out.cpu().numpy(), H2, W2)
    assert np.array_equal(got[0], rgbs[0]) and np.array_equal(got[2], rgbs[2])
    rec = c.decode_v(cont, seg, Hs, Ws, mode)
    c.check()
    assert np.array_equal(rec.cpu().numpy(), _flat(rgbs))


def test_mixed_size_batch_stream_count_per_image(torch_mod, codecs, oracle_weights):
    """llicti_encode_images_vm / llicti_decode_images_vm: one container mode PER IMAGE -- rANS streams of one lane kind whose COUNT differs from image
    to image (every header carries its own): image b's bytes are those of its own single-image encode in modes[b] and the oracle's; lossless on a
    poisoned workspace; container "auto" gives each image a count from its own size; lane kinds must not be mixed, nor the reference format with rANS."""
    from llicti_amd import _lib
    from llicti_amd.codec import MODE_AC, MODE_RANS, container_to_bytestream_list
    from oracle import oracle as orc
    torch = torch_mod
    c = codecs("trainedlike")
    W_o = oracle_weights("trainedlike")
    shapes = [(150, 131), (96, 160), (224, 301), (67, 93), (160, 352), (150, 131)]
    rgbs = [make_image("smooth" if i % 2 else "noise", h, w, 800 + i) for i, (h, w) in enumerate(shapes)]
    Hs, Ws = [h for h, _ in shapes], [w for _, w in shapes]
    flat = _dev(torch, _flat(rgbs))
    for wide, Ms in ((2, [2, 1, 5, 1, 4, 3]), (1, [3, 1, 2, 2, 1, 3]), (0, [4, 8, 1, 2, 32, 3])):
        modes = [MODE_RANS(m, wide=wide) for m in Ms]
        cont, seg = c.encode_v(flat, Hs, Ws, modes)
        c.check()
        cont_h, seg_h = cont.cpu().numpy(), seg.cpu().numpy()
        for b, rgb in enumerate(rgbs):
            c1, s1 = c.encode(_dev(torch, rgb[None]), mode=modes[b])
            n = int(seg_h[b].sum())
            assert np.array_equal(seg_h[b], s1[0].cpu().numpy()) and np.array_equal(cont_h[b, :n], c1[0, :n].cpu().numpy()), (wide, b)
            if b in (0, 2, 4):
                assert container_to_bytestream_list(cont_h[b], seg_h[b]) == orc.encode_image_rans(rgb, W_o, Ms[b], wide), (wide, b)
        c.poison_workspace()
        rec = c.decode_v(cont, seg, Hs, Ws, modes)
        c.check()
        assert np.array_equal(rec.cpu().numpy(), _flat(rgbs)), wide
    with pytest.raises(_lib.LlictiError):
        c.encode_v(flat, Hs, Ws, [MODE_RANS(2, wide=2)] * 5 + [MODE_RANS(2, wide=1)])
    with pytest.raises(_lib.LlictiError):
        c.encode_v(flat, Hs, Ws, [MODE_RANS(2)] * 5 + [MODE_AC])
    # container "auto" on a batch at the reference's eval-set sizes: each image's count from its own size, streams of about equal length, and with
    # the "auto" ENCODER modes mixed with fixed ones refused (all of a call or none)
    import json
    import os
    from conftest import GOLDEN
    from llicti_amd.codec import MODE_RANS_AUTO, auto_modes, image_streams
    sh = [tuple(s) for s in json.load(open(os.path.join(GOLDEN, "eval_shapes.json")))["shapes"][100:124]]
    bm = auto_modes(sh)
    Mb = [m & 0xFF for m in bm]
    assert all(m == MODE_RANS_AUTO(image_streams(h, w)) for m, (h, w) in zip(bm, sh))
    per = [h * w / m for m, (h, w) in zip(Mb, sh)]
    assert max(per) / min(per) < 1.35                                  # about equally long streams (one count for all: 2.2)
    with pytest.raises(_lib.LlictiError):
        c.encode_v(flat, Hs, Ws, [MODE_RANS_AUTO(2)] * 5 + [MODE_RANS(2, wide=2)])
    with pytest.raises(_lib.LlictiError):
        c.decode_v(cont, seg, Hs, Ws, [MODE_RANS_AUTO(2)] * 6)         # an encoder's mode: a decoder takes the container's own
    c.check()
