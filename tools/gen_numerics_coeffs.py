#!/usr/bin/env python3
"""Derive the polynomial coefficients of the LLICTI-MI355X fp32 numerics spec.

erfc_spec(x), x >= 0 :   r = 1/(x+2);  t = (x-2)*r;  erfc = P(t) * exp_spec(-x*x) * r
   where P(t) ~ erfcx(x)*(x+2) on x in [0, XMAX], fitted here (least squares at Chebyshev nodes,
   then printed as fp32 hex literals).
exp_spec(y), y in [-87, 0]: j = rint(y*log2e); f = y - j*ln2 (two-step fma); exp = 2^j * Q(f)
   Q(f) ~ exp(f) on |f| <= ln2/2.

The emitted numbers are pasted into oracle/llicti_oracle.c and llicti_amd/csrc/numerics.hpp (two
independent restatements of the same spec); this script is kept so they can be re-derived.
"""
import numpy as np
from scipy import special

XMAX = 7.0   # erfc_spec(x) := 0 for x >= 7 (erfc(7) = 4.2e-23: below anything the 16-bit tables can see,
             # and keeps every intermediate in the fp32 normal range)


def fit_erfc(deg):
    tmin, tmax = -1.0, (XMAX - 2) / (XMAX + 2)
    n = 400
    k = np.arange(n)
    u = np.cos(np.pi * (k + 0.5) / n)
    t = 0.5 * (tmax - tmin) * u + 0.5 * (tmax + tmin)
    x = 2 * (1 + t) / (1 - t)
    f = special.erfcx(x) * (x + 2)
    V = np.vander(t, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V, f, rcond=None)
    return c


def fit_exp(deg):
    h = np.log(2) / 2 * 1.0001
    n = 200
    u = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    f = u * h
    # exp(f) = 1 + f + f^2 * R(f)
    R = (np.exp(f) - 1 - f) / (f * f)
    V = np.vander(f, deg - 1, increasing=True)
    c, *_ = np.linalg.lstsq(V, R, rcond=None)
    return c


f32 = np.float32


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def exp_spec(y, ce):
    y = y.astype(f32)
    j = np.rint(y * f32(1.4426950408889634)).astype(f32)
    f = fma32(j, np.full_like(y, f32(-0.693145751953125)), y)
    f = fma32(j, np.full_like(y, f32(-1.428606765330187e-06)), f)
    q = np.full_like(y, f32(ce[-1]))
    for c in ce[-2::-1]:
        q = fma32(q, f, np.full_like(y, f32(c)))
    f2 = (f * f).astype(f32)
    q = fma32(q, f2, f)
    q = (q + f32(1.0)).astype(f32)
    ji = j.astype(np.int32)
    return (q.view(np.int32) + (ji << 23)).view(f32)


def erfc_spec(x, cp, ce):
    x = x.astype(f32)
    r = (f32(1.0) / (x + f32(2.0))).astype(f32)
    t = ((x - f32(2.0)) * r).astype(f32)
    p = np.full_like(x, f32(cp[-1]))
    for c in cp[-2::-1]:
        p = fma32(p, t, np.full_like(x, f32(c)))
    s = (x * x).astype(f32)
    e = fma32(x, x, -s)
    ex = exp_spec(-s, ce)
    ex = fma32(-e, ex, ex)
    return ((p * ex).astype(f32) * r).astype(f32)


if __name__ == "__main__":
    for deg in (9, 10, 11):
        cp = fit_erfc(deg)
        for dexp in (5, 6):
            ce = fit_exp(dexp)
            x = np.linspace(0, XMAX, 2_000_001).astype(f32)
            got = erfc_spec(x, cp, ce).astype(np.float64)
            ref = special.erfc(x.astype(np.float64))
            ulp = np.spacing(ref.astype(f32)).astype(np.float64)
            err = np.abs(got - ref) / ulp
            print(f"deg {deg} exp {dexp}: max ulp err {err.max():.3f} at x={x[err.argmax()]:.4f}  mean {err.mean():.3f}")
    cp = fit_erfc(10)
    ce = fit_exp(6)
    print("P coefficients (t^0..):")
    for c in cp:
        print(f"  {float(f32(c)).hex()}f,  /* {f32(c)!r} */")
    print("Q coefficients (R(f), f^0..):")
    for c in ce:
        print(f"  {float(f32(c)).hex()}f,  /* {f32(c)!r} */")
    print("ln2_hi", float(f32(0.693145751953125)).hex(), "ln2_lo", float(f32(1.428606765330187e-06)).hex(),
          "log2e", float(f32(1.4426950408889634)).hex())
