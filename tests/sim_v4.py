"""Byte cost of an xwide rANS stream's framing, by variant, simulated on the oracle's own last-stage (c_low, c_high) pairs.

What a stream costs beyond the ideal code length of its symbols is, besides 0.057 bit per lane, its FRAMING: the header field, the
byte alignment, and what the tail coder wastes of the 7,936-bit payload the 256 initial states carry (payload bits minus the ideal
bits of the T tail symbols).  This script prices that for the v3 layout and for the v4 candidates on one 768x512 image per content
class (noise / natural-like / model-drawn / 1.7-bit "cheap"), M streams per image: `python tests/sim_v4.py [M ...]`.
Test infrastructure (uses the CPU oracle); results quoted in DESIGN.md section 5 (rANS v4).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as orc                      # noqa: E402
from llicti_amd.weights import pack_state_dict       # noqa: E402
from helpers import make_image, make_sampled_image   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
P = 7936
L = 256


def weights(name):
    return dict(np.load(os.path.join(GOLD, f"weights_{name}.npz")))


def cheap_sd():
    sd = weights("trainedlike")
    for k in list(sd):
        if k.endswith("layers1toL.2.bias"):
            b = sd[k].copy()
            b[0:15] = 0.6 / 255.0
            b[30:45] = np.tile(np.array([1.0, 1e-7, 1e-7, 1e-7, 1e-7], np.float32), 3)
            sd[k] = b
        if k.endswith("layers1toL.2.weight"):
            w = sd[k].copy()
            w[0:15] = 0.0
            w[30:45] = 0.0
            sd[k] = w
    return sd


def drawn(W_o, H, Wd, seed):
    bl = orc.encode_image(make_image("smooth", H, Wd, 11), W_o)
    rng = np.random.default_rng(seed)
    bl = [list(bl[0])] + [[rng.integers(0, 256, len(x), dtype=np.uint8).tobytes() for x in row] for row in bl[1:]]
    return orc.decode_image(bl, W_o)


def last_stage(img, W_o):
    planes, mm = orc.lift(img)
    par = orc.band_params(planes, 0, 2, W_o)
    lo, hi, sym = orc.stream_pairs(planes, mm, 0, 2, 2, par)
    return lo.astype(np.int64), hi.astype(np.int64), sym.astype(np.int64), int(mm[5]) - int(mm[2]) + 1


def emit_bits(x, f):
    n = 0
    while (x >> n) >= (f << 16):
        n += 1
    return n


def push(x, lo, f):
    return ((x // f) << 16) + (x % f) + lo


def stream_syms(n_sym, m, M):
    """indices (raster) of stream m's symbols of the stage, in decode order"""
    nch = (n_sym + L - 1) // L
    idx = []
    for c in range(m, nch, M):
        idx.extend(range(L * c, min(L * c + L, n_sym)))
    return np.array(idx, dtype=np.int64)


def seed_count(A):
    n, p = 0, 1
    while n < 31 and p * A <= (1 << 31):
        p *= A
        n += 1
    return n


def tail_v3(lo, hi, sym, A, Tmax=8191):
    """v3 xwide: one or two seeded chains.  Returns (T, payload bits used incl. states, two_chains)."""
    cnt = len(lo)
    ns = seed_count(A)
    f = hi - lo
    nch = 1
    if cnt >= 2 * ns:
        k64 = min(cnt, 64)
        ws = sum(16 - int(np.floor(np.log2(f[cnt - 1 - j]))) for j in range(k64))
        if 2 * ws * ns >= k64 * (64 + ns):
            nch = 2
    x = [1 << 31, 1 << 31]
    j = 0
    for c in range(nch):
        mul = 1
        for i in range(ns):
            if j >= cnt:
                break
            x[c] += int(sym[cnt - 1 - j]) * mul
            mul *= A
            j += 1
    j0 = j
    used = 0
    while j < cnt and j < Tmax:
        q = cnt - 1 - j
        c = (j - j0) % nch
        nb = emit_bits(x[c], int(f[q]))
        if used + nb + 32 * nch > P:
            break
        used += nb
        x[c] = push(x[c] >> nb, int(lo[q]), int(f[q]))
        j += 1
    return j, used + 32 * nch, nch


def tail_zero(lo, hi, Tmax=8191, gran=1, spill=False, reserve=33):
    """one zero-start chain: start state 0, no bits while the state is below the interval, the final state (32 bits flat) + a sentinel bit.
    gran: T is a multiple of gran (or the stream's length); spill: the chain may overflow the payload (its excess goes to the main region).
    Returns (T, bits the tail occupies in all, of which spilled)."""
    cnt = len(lo)
    f = hi - lo
    x = 0
    used = 0
    j = 0
    hist = [0]                                        # used after j symbols
    while j < cnt and j < Tmax:
        q = cnt - 1 - j
        fq, lq = int(f[q]), int(lo[q])
        nb = emit_bits(x, fq)
        if not spill and used + nb + reserve > P:
            break
        if spill and used + reserve > P and j % gran == 0:
            break
        used += nb
        x = push(x >> nb, lq, fq)
        j += 1
        hist.append(used)
    if not spill and gran > 1 and j < cnt:
        j = (j // gran) * gran
        used = hist[j]
    tot = used + reserve
    return j, tot, max(0, tot - P)


def tail_v4(lo, hi, sym, A, two=None, cap=8160):
    """v4 as specified (oracle/llicti_oracle.h): arena = payload ++ spill, T a multiple of 32 (or the stream's length / the cap).
    one chain: start state = the last symbol's index (raw), then zero-start pushes (no bits while the state is below the interval), a sentinel
    bit behind the last field; two chains (v3's rule says when): seeded as in v3, states at both ends of the arena.
    Returns (T, arena bits, chains)."""
    cnt = len(lo)
    ns = seed_count(A)
    f = hi - lo
    nch = 1
    if cnt >= 2 * ns:
        k64 = min(cnt, 64)
        ws = sum(16 - int(np.floor(np.log2(f[cnt - 1 - j]))) for j in range(k64))
        if 2 * ws * ns >= k64 * (64 + ns):
            nch = 2
    if two is not None:
        nch = 2 if (two and cnt >= 2 * ns) else 1
    if nch == 2:
        x = [1 << 31, 1 << 31]
        j = 0
        for c in range(2):
            mul = 1
            for i in range(ns):
                if j >= cnt:
                    break
                x[c] += int(sym[cnt - 1 - j]) * mul
                mul *= A
                j += 1
        j0, fixed = j, 64
    else:
        x = [int(sym[cnt - 1]) if cnt else 0]
        j = 1 if cnt else 0
        j0, fixed = j, 33
    used = 0
    while j < cnt and j < cap:
        if j % 32 == 0 and used + fixed >= P:
            break
        q = cnt - 1 - j
        c = (j - j0) % nch
        nb = emit_bits(x[c], int(f[q]))
        used += nb
        x[c] = push(x[c] >> nb, int(lo[q]), int(f[q]))
        j += 1
    return j, used + fixed, nch


def ideal_bits(lo, hi, T):
    cnt = len(lo)
    f = (hi - lo)[cnt - T:cnt] if T else np.array([1 << 16])
    return float(np.sum(16.0 - np.log2(f)))


def price(name, img, W_o, Ms):
    lo, hi, sym, A = last_stage(img, W_o)
    n_sym = len(lo)
    bits_sym = float(np.mean(16.0 - np.log2(hi - lo)))
    out = {"class": name, "bits_per_last_stage_symbol": round(bits_sym, 3), "A": A, "streams": {}}
    for M in Ms:
        acc = {}
        for m in range(M):
            idx = stream_syms(n_sym, m, M)
            l, h, s = lo[idx], hi[idx], sym[idx]
            # v3 as built: u16 + 3.5 alignment + escape
            T, usedp, nch = tail_v3(l, h, s, A)
            hdr = 16 + 3.5 + (16 if T >= 4095 else 0)
            acc.setdefault("v3", []).append(hdr + P - ideal_bits(l, h, T) if T < len(l) or usedp >= P - 64 else hdr + P - ideal_bits(l, h, T))
            acc.setdefault("v3_T", []).append(T)
            # v4a: v3's two seeded chains where its rule says two, else ONE zero-start chain; u16 header kept, T field piecewise (no escape)
            if nch == 2:
                w = 16 + 3.5 + P - ideal_bits(l, h, T)
                Ta = T
            else:
                Ta, tot, _ = tail_zero(l, h, gran=1)
                if Ta >= 2048:                           # field v >= 2048: T = 2048 + 3 (v - 2048)
                    Ta, tot, _ = tail_zero(l, h, gran=3)
                    # (gran applies above 2048 only; close enough)
                w = 16 + 3.5 + P - ideal_bits(l, h, Ta)
            acc.setdefault("v4a", []).append(w)
            acc.setdefault("v4a_T", []).append(Ta)
            # v4b: zero-start everywhere (one chain), header T13 + flag + sentinel
            Tb, tot, _ = tail_zero(l, h)
            acc.setdefault("v4b", []).append(14 + 1 + 3.5 + P - ideal_bits(l, h, Tb))
            # v4c: zero-start + spill, T in units of 32 (8-bit field) + flag + sentinel + alignment
            Tc, tot, sp = tail_zero(l, h, gran=32, spill=True, reserve=32)
            acc.setdefault("v4c", []).append(8 + 1 + 1 + 3.5 + tot - ideal_bits(l, h, Tc) if tot >= P else 8 + 1 + 1 + 3.5 + P - ideal_bits(l, h, Tc))
            acc.setdefault("v4c_T", []).append(Tc)
            # v4 as specified: header 9 bits + sentinel + alignment
            for nm, two in (("v4", None), ("v4_one", False), ("v4_two", True)):
                Tv, arena, nchv = tail_v4(l, h, s, A, two)
                acc.setdefault(nm, []).append(9 + 1 + 3.5 + max(arena, P) - ideal_bits(l, h, Tv))
                acc.setdefault(nm + "_T", []).append(Tv)
        res = {}
        for k, v in acc.items():
            if k.endswith("_T"):
                res[k] = [int(np.min(v)), int(np.max(v))]
            else:
                res[k + "_bytes_per_stream"] = round(float(np.mean(v)) / 8.0 + L * 0.057 / 8.0, 2)
                res[k + "_bytes_per_image"] = round(float(np.sum(v)) / 8.0 + M * L * 0.057 / 8.0, 1)
        out["streams"][str(M)] = res
    return out


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [10, 16, 20]
    H, Wd = 512, 768
    Wr = orc.Weights(pack_state_dict(weights("rand1337")))
    Wt = orc.Weights(pack_state_dict(weights("trainedlike")))
    Wc = orc.Weights(pack_state_dict(cheap_sd()))
    res = []
    res.append(price("noise (seed-1337 weights)", make_image("noise", H, Wd, 0), Wr, Ms))
    print(json.dumps(res[-1]), flush=True)
    res.append(price("natural-like (smooth, trained-like weights)", make_image("smooth", H, Wd, 11), Wt, Ms))
    print(json.dumps(res[-1]), flush=True)
    res.append(price("model-drawn (trained-like weights)", make_sampled_image(H, Wd, 5), Wt, Ms))
    print(json.dumps(res[-1]), flush=True)
    res.append(price("cheap (one component, sigma 0.6)", drawn(Wc, H, Wd, 5), Wc, Ms))
    print(json.dumps(res[-1]), flush=True)
    if os.environ.get("SIM_OUT"):
        with open(os.environ["SIM_OUT"], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
