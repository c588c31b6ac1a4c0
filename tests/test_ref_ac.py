"""The torchac seam (LLICTI_nets.py:406-407, :492-493), read twice: tests/ref_ac.py (pure Python, bit at a time,
written from SURVEY.md Appendix A) against oracle/llicti_oracle.c's coder (CPU tests) and against the HIP coder
through the C-ABI (-m gpu), on the reference's own recorded tables, random tables and edge rows -- plus three
known-answer vectors worked out by hand below.  torchac itself is absent from this image: still "parity unpinned",
but two independent readings and the kernels agree byte for byte."""
import numpy as np
import pytest

import ref_ac
from conftest import load_case
from oracle import oracle as orc


# ---------------------------------------------------------------------------------------------- by hand
# Notation: span = high - low + 1; high' = low - 1 + (span * c_high >> 16); low' = low + (span * c_low >> 16);
# E1: high < 2^31 -> emit 0 (+ pending ones); E2: low >= 2^31 -> emit 1 (+ pending zeros); both shift low, high
# (high gets a 1); E3: 2^30 <= low and high < 3 * 2^30 -> pending += 1, low = (low << 1) & 0x7FFFFFFF,
# high = (high << 1) | 0x80000001.  End: pending += 1; emit (low < 2^30 ? 0 : 1) + pending complements; pad with 0s.
HAND = [
    # 1. Lp = 3, one symbol s = 0 of row [0, 0x8000, (0)]:
    #    span = 2^32; high' = -1 + (2^32 * 0x8000 >> 16) = 0x7FFFFFFF; low' = 0.
    #    E1: emit 0; low = 0, high = 0xFFFFFFFF; no rule applies.
    #    End: pending = 1, low < 2^30 -> emit 0 then one 1.        bits 0 0 1 -> 0010 0000
    ([[0, 0x8000, 0]], [0], bytes([0x20])),
    # 2. same row, s = 1 = max_symbol, so c_high is the constant 0x10000 (the row's last word has wrapped to 0):
    #    high' = -1 + (2^32 * 0x10000 >> 16) = 0xFFFFFFFF; low' = 2^32 * 0x8000 >> 16 = 0x80000000.
    #    E2: emit 1; low = 0, high = 0xFFFFFFFF.
    #    End: pending = 1, emit 0 then one 1.                      bits 1 0 1 -> 1010 0000
    ([[0, 0x8000, 0]], [1], bytes([0xA0])),
    # 3. Lp = 4, row [0, 0x4000, 0xC000, (0)], symbols 1 then 0 (an underflow step whose pending bit is resolved by the
    #    next symbol):
    #    s = 1: high' = -1 + 0xC0000000 = 0xBFFFFFFF, low' = 0x40000000.  Neither E1 nor E2; E3 holds: pending = 1,
    #           low = 0x80000000 & 0x7FFFFFFF = 0, high = 0x7FFFFFFE | 0x80000001 = 0xFFFFFFFF; no rule applies.
    #    s = 0: span = 2^32; high' = -1 + 0x40000000 = 0x3FFFFFFF, low' = 0.
    #           E1: emit 0 and the pending 1; low = 0, high = 0x7FFFFFFF.  E1 again: emit 0; high = 0xFFFFFFFF.
    #    End: pending = 1, low < 2^30: emit 0 then 1.              bits 0 1 0 0 1 -> 0100 1000
    ([[0, 0x4000, 0xC000, 0], [0, 0x4000, 0xC000, 0]], [1, 0], bytes([0x48])),
]


@pytest.mark.parametrize("k", range(len(HAND)))
def test_hand_worked_vectors(k):
    rows, syms, want = HAND[k]
    assert ref_ac.encode(rows, syms) == want
    a = np.array(rows, dtype=np.uint16)
    assert orc.ac_encode_tables(a, np.array(syms, np.int16)) == want
    assert ref_ac.decode(rows, want) == syms
    assert list(orc.ac_decode_tables(a, want)) == syms


def _hand3_decode_by_hand():
    # decode of vector 3, first symbol: value = 0x48000000, low = 0, span = 2^32:
    # count = ((0x48000000 + 1) * 0x10000 - 1) >> 32 = 0x4800; binary search in [0, 0x4000, 0xC000]: m = 1: 0x4000 < 0x4800
    # -> left = 1; m = 2: 0xC000 > 0x4800 -> right = 2 -> symbol 1.
    return 0x4800, 1


def test_hand_decode_count():
    count, sym = _hand3_decode_by_hand()
    assert ((0x48000000 + 1) * 0x10000 - 1) // (1 << 32) == count
    assert ref_ac._binsearch([0, 0x4000, 0xC000, 0], count, 2) == sym


# ---------------------------------------------------------------------------------------------- generators
def random_rows(rng, N, Lp, peaky=4.0):
    """Rows in the reference's format (LLICTI_nets.py:955-983): q + arange(Lp), last entry wraps to 0."""
    pm = rng.random((N, Lp - 1)) ** peaky + 1e-4
    cum = np.concatenate([np.zeros((N, 1)), np.cumsum(pm, 1)], 1)
    cum /= cum[:, -1:]
    q = np.rint(cum * (65536 - (Lp - 1))).astype(np.int64) + np.arange(Lp)
    return (q & 0xFFFF).astype(np.uint16)


def edge_cases(rng):
    """(name, rows uint16 [N, Lp], symbols int16 [N])"""
    out = []
    # alphabet of one symbol (Lp = 2): every symbol is the top symbol with probability 1 -> no information
    out.append(("Lp2", np.tile(np.array([[0, 0]], np.uint16), (37, 1)), np.zeros(37, np.int16)))
    # always the top symbol, at minimum probability 1/65536 (c_low = 0xFFFF, c_high = the 0x10000 constant)
    Lp = 9
    rows = np.tile(np.array([list(range(0, Lp - 2)) + [0xFFFF, 0]], np.uint16), (50, 1))
    out.append(("top_symbol_min_prob", rows, np.full(50, Lp - 2, np.int16)))
    # always symbol 0 at minimum probability
    rows = np.tile(np.array([[0, 1, 0x8000, 0]], np.uint16), (50, 1))
    out.append(("symbol0_min_prob", rows, np.zeros(50, np.int16)))
    # long pending runs: a narrow interval straddling the middle keeps the coder in the underflow rule
    rows = np.tile(np.array([[0, 0x7FFF, 0x8001, 0]], np.uint16), (200, 1))
    sym = np.ones(200, np.int16)
    out.append(("pending_run_then_low", rows, np.concatenate([sym[:-1], [0]]).astype(np.int16)))
    out.append(("pending_run_then_high", rows, np.concatenate([sym[:-1], [2]]).astype(np.int16)))
    out.append(("pending_run_open_end", rows, sym))
    # full-width alphabets with uniform and with wildly peaked rows
    for Lp, peaky in ((512, 1.0), (512, 12.0), (257, 6.0), (3, 1.0)):
        rows = random_rows(rng, 300, Lp, peaky)
        out.append((f"random_Lp{Lp}_p{peaky}", rows, rng.integers(0, Lp - 1, 300).astype(np.int16)))
    # symbols drawn FROM the rows' distribution (what a real stream looks like) and the least likely symbol of every row
    rows = random_rows(rng, 400, 257, 8.0)
    pm = np.diff(np.concatenate([rows[:, :-1].astype(np.int64), np.full((400, 1), 65536)], 1), axis=1)
    out.append(("likely_symbols", rows, np.array([rng.choice(256, p=p / p.sum()) for p in pm], np.int16)))
    out.append(("least_likely_symbols", rows, pm.argmin(1).astype(np.int16)))
    return out


def reference_recorded_tables():
    """The (cdf rows, symbols) the reference itself handed to torchac (tests/golden/make_fixtures.py's recorder)."""
    out = []
    for case in ("smooth_64x48_tl", "noise_33x64_tl", "noise_32x32_rand", "smooth_67x93_tl"):
        c = load_case(case)
        for tag in ("s4_b1_c0", "s3_b0_c1", "s2_b2_c2", "s1_b0_c0", "s0_b1_c1", "s0_b2_c2"):
            rows, idx = c["cdfrows_" + tag], c["cdfidx_" + tag]
            out.append((f"{case}/{tag}", np.ascontiguousarray(rows), c["sym_" + tag].ravel()[idx].astype(np.int16)))
    return out


def all_cases():
    rng = np.random.default_rng(2024)
    return edge_cases(rng) + reference_recorded_tables()


# ---------------------------------------------------------------------------------------------- CPU: two readings
def test_ref_ac_equals_oracle_coder():
    for name, rows, sym in all_cases():
        want = ref_ac.encode(rows.tolist(), sym.tolist())
        got = orc.ac_encode_tables(rows, sym)
        assert got == want, name
        assert ref_ac.decode(rows.tolist(), want) == sym.tolist(), name
        assert np.array_equal(orc.ac_decode_tables(rows, want), sym), name
        # any zero padding behind the stream reads as the zeros torchac's get() supplies
        assert ref_ac.decode(rows.tolist(), want + b"\0\0\0") == sym.tolist(), name


def test_ref_ac_whole_stream_of_an_image(oracle_weights):
    """One complete stream of an image (every symbol, not a sample): the oracle's stream bytes == ref_ac's on the
    oracle's own full tables."""
    c = load_case("smooth_64x48_tl")
    W = oracle_weights("trainedlike")
    rgb = c["rgb"]
    bl = orc.encode_image(rgb, W)
    planes, mm = orc.lift(rgb)
    for (lvl, band, clr) in ((2, 0, 0), (1, 1, 1), (3, 2, 2)):
        par = orc.band_params(planes, lvl, band, W)
        clow, chigh, sym = orc.stream_pairs(planes, mm, lvl, band, clr, par)
        minv = -127 if clr == 0 else int(mm[clr])
        maxv = 128 if clr == 0 else int(mm[3 + clr])
        oi, oj = {0: (1, 1), 1: (0, 1), 2: (1, 0)}[band]
        H, Wd = rgb.shape[1:]
        Hl, Wl, h, w, padH, padW = orc.level_geom(H, Wd, lvl)
        hc = h - padH if band in (0, 2) else h
        wc = w - padW if band in (0, 1) else w
        rows = []
        for n in range(hc * wc):
            i, j = divmod(n, wc)
            R, Cc = (2 * i + oi) << lvl, (2 * j + oj) << lvl
            rows.append(orc.cdf_row(par[i, j], clr, np.float32(planes[0, R, Cc]) / np.float32(255),
                                    np.float32(planes[1, R, Cc]) / np.float32(255), minv, maxv).tolist())
        want = bl[1 + (4 - lvl)][3 * band + clr]
        assert ref_ac.encode(rows, sym.tolist()) == want, (lvl, band, clr)
        assert ref_ac.decode(rows, want) == sym.tolist()


# ---------------------------------------------------------------------------------------------- GPU: the kernels
@pytest.mark.gpu
def test_hip_coder_equals_ref_ac():
    """llicti_ac_encode_u16cdf / llicti_ac_decode_u16cdf (the torchac seam of the C-ABI) against the pure-Python
    reading, on the reference's recorded tables and on the edge rows."""
    import torch
    from llicti_amd.codec import HipCodec
    c = HipCodec("cuda:0")
    try:
        for name, rows, sym in all_cases():
            N, Lp = rows.shape
            stride = max(8, (Lp + 7) // 8 * 8)
            cdf = np.full((1, N, stride), 0xFFFF, np.uint16)
            cdf[0, :, :Lp] = rows
            want = ref_ac.encode(rows.tolist(), sym.tolist())
            out, ln = c.ac_encode(torch.from_numpy(cdf.view(np.int16)).cuda(), torch.from_numpy(sym[None].copy()).cuda(), Lp)
            got = bytes(out[0, :int(ln[0])].cpu().numpy())
            assert got == want, name
            in_stride = (len(want) + 3) // 4 * 4 + 16
            buf = np.zeros((1, in_stride), np.uint8)
            buf[0, :len(want)] = np.frombuffer(want, np.uint8)
            dec = c.ac_decode(torch.from_numpy(cdf.view(np.int16)).cuda(), Lp, torch.from_numpy(buf).cuda(),
                              torch.tensor([len(want)], dtype=torch.int32).cuda(), N)
            assert np.array_equal(dec[0].cpu().numpy(), sym), name
    finally:
        c.close()
