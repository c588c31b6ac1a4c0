"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (llicti_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False):
    src = [os.path.join(_HERE, f) for f in ("llicti_oracle.c", "llicti_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class _BandW(C.Structure):
    _fields_ = [("K0", C.c_int), ("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p),
                ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p)]


class _Weights(C.Structure):
    _fields_ = [("band", _BandW * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_encode_image.restype = C.c_long
        L.orc_encode_image_rans.restype = C.c_long
        L.orc_ac_encode_tables.restype = C.c_long
        L.orc_ac_encode_pairs.restype = C.c_long
        L.orc_stream_pairs.restype = C.c_long
        L.orc_cdf_float.restype = C.c_float
        L.orc_erfc.restype = C.c_float
        L.orc_erfc.argtypes = [C.c_float]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Weights:
    """Holds the canonical packed arrays (llicti_amd.weights.pack_state_dict output) alive for C."""

    def __init__(self, packed: dict):
        self.packed = packed
        self.c = _Weights()
        for b in range(3):
            d = packed[b]
            bw = self.c.band[b]
            bw.K0 = int(d["K0"])
            for k in ("w0", "b0", "w1", "b1", "w2", "b2"):
                a = d[k]
                assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
                setattr(bw, k, a.ctypes.data)

    def band_ptr(self, b):
        return C.byref(self.c.band[b])


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def erfc(x):
    L = lib()
    x = np.asarray(x, dtype=np.float32)
    return np.array([L.orc_erfc(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def lift(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    planes = np.empty((3, H, W), np.int16)
    mm = np.empty(6, np.int16)
    lib().orc_lift(_p(rgb), H, W, _p(planes), _p(mm))
    return planes, mm


def unlift(planes):
    planes = np.ascontiguousarray(planes, dtype=np.int16)
    _, H, W = planes.shape
    rgb = np.empty((3, H, W), np.uint8)
    lib().orc_unlift(_p(planes), H, W, _p(rgb))
    return rgb


def level_geom(H, W, lvl):
    v = [C.c_int() for _ in range(6)]
    lib().orc_level_geom(H, W, lvl, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)  # Hl, Wl, h, w, padH, padW


def band_params(planes, lvl, band, weights: Weights):
    planes = np.ascontiguousarray(planes, dtype=np.int16)
    _, H, W = planes.shape
    _, _, h, w, _, _ = level_geom(H, W, lvl)
    out = np.empty((h, w, 60), np.float32)
    lib().orc_band_params(_p(planes), H, W, lvl, band, weights.band_ptr(band), _p(out))
    return out


def lift_train(rgb):
    """uint8 [3,H,W] -> float32 [3,H,W]: (Y - 127/255, Co, Cg) of the training path's float lift."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    out = np.empty((3, H, W), np.float32)
    lib().orc_lift_train(_p(rgb), H, W, _p(out))
    return out


def band_params_f(fplanes, lvl, band, weights: "Weights"):
    fplanes = np.ascontiguousarray(fplanes, dtype=np.float32)
    _, H, W = fplanes.shape
    _, _, h, w, _, _ = level_geom(H, W, lvl)
    out = np.empty((h, w, 60), np.float32)
    lib().orc_band_params_f(_p(fplanes), H, W, lvl, band, weights.band_ptr(band), _p(out))
    return out


def selfinfo(fplanes, lvl, band, params):
    """-> float32 [3, h, w]: -log2 pmf of the band's Y, Co, Cg targets (reference get_self_infos)."""
    fplanes = np.ascontiguousarray(fplanes, dtype=np.float32)
    params = np.ascontiguousarray(params, dtype=np.float32)
    _, H, W = fplanes.shape
    _, _, h, w, _, _ = level_geom(H, W, lvl)
    out = np.empty((3, h, w), np.float32)
    lib().orc_selfinfo(_p(fplanes), H, W, lvl, band, _p(params), _p(out))
    return out


def forward(rgb, weights: "Weights"):
    """LLICTI.forward on one uint8 image (H, W multiples of 32): list of 5 arrays [9, h, w], scale 0 first."""
    fp = lift_train(rgb)
    res = []
    for lvl in range(5):
        res.append(np.concatenate([selfinfo(fp, lvl, b, band_params_f(fp, lvl, b, weights)) for b in range(3)]))
    return res


def cdf_row(par, clr, yv, cov, minv, maxv):
    par = np.ascontiguousarray(par, dtype=np.float32)
    row = np.empty(maxv - minv + 2, np.uint16)
    lib().orc_cdf_row(_p(par), clr, C.c_float(yv), C.c_float(cov), minv, maxv, _p(row))
    return row


def cdf_float(par, clr, yv, cov, pt):
    par = np.ascontiguousarray(par, dtype=np.float32)
    return float(lib().orc_cdf_float(_p(par), clr, C.c_float(yv), C.c_float(cov), C.c_float(pt)))


def stream_pairs(planes, minmax, lvl, band, clr, params):
    planes = np.ascontiguousarray(planes, dtype=np.int16)
    minmax = np.ascontiguousarray(minmax, dtype=np.int16)
    params = np.ascontiguousarray(params, dtype=np.float32)
    _, H, W = planes.shape
    _, _, h, w, _, _ = level_geom(H, W, lvl)
    clow = np.empty(h * w, np.uint32)
    chigh = np.empty(h * w, np.uint32)
    sym = np.empty(h * w, np.int16)
    n = lib().orc_stream_pairs(_p(planes), H, W, _p(minmax), lvl, band, clr, _p(params), _p(clow), _p(chigh), _p(sym))
    return clow[:n], chigh[:n], sym[:n]


def ac_encode_tables(cdf, sym):
    cdf = np.ascontiguousarray(cdf).view(np.uint16)
    sym = np.ascontiguousarray(sym, dtype=np.int16).ravel()
    Lp = cdf.shape[-1]
    N = sym.size
    assert cdf.size == N * Lp
    out = np.empty(2 * N + 16, np.uint8)
    n = lib().orc_ac_encode_tables(_p(cdf), Lp, _p(sym), C.c_long(N), _p(out), C.c_long(out.size))
    assert n >= 0
    return out[:n].tobytes()


def ac_encode_pairs(clow, chigh):
    clow = np.ascontiguousarray(clow, dtype=np.uint32)
    chigh = np.ascontiguousarray(chigh, dtype=np.uint32)
    N = clow.size
    out = np.empty(2 * N + 16, np.uint8)
    n = lib().orc_ac_encode_pairs(_p(clow), _p(chigh), C.c_long(N), _p(out), C.c_long(out.size))
    assert n >= 0
    return out[:n].tobytes()


def ac_decode_tables(cdf, stream: bytes, N=None):
    cdf = np.ascontiguousarray(cdf).view(np.uint16)
    Lp = cdf.shape[-1]
    if N is None:
        N = cdf.size // Lp
    buf = np.frombuffer(stream, dtype=np.uint8)
    sym = np.empty(N, np.int16)
    lib().orc_ac_decode_tables(_p(cdf), Lp, _p(buf) if buf.size else None, C.c_long(buf.size), C.c_long(N), _p(sym))
    return sym


def _segments_to_list(buf, seg_len):
    """Flat container -> the reference's bytestream_list (LLICTI_nets.py:352-354, :411)."""
    segs, pos = [], 0
    for n in seg_len:
        segs.append(bytes(buf[pos:pos + n]))
        pos += n
    em = b""
    bl = [[segs[0], segs[1], segs[2], segs[3], em, em, em, em, em]]
    for s in range(5):
        bl.append(segs[4 + 9 * s: 4 + 9 * (s + 1)])
    return bl


def _list_to_segments(bl):
    segs = list(bl[0][:4])
    for s in range(1, 6):
        segs += list(bl[s])
    seg_len = np.array([len(s) for s in segs], dtype=np.int32)
    return np.frombuffer(b"".join(segs), dtype=np.uint8).copy(), seg_len


def encode_image_rans(rgb, weights: "Weights", M=8, wide=False, auto=False):
    """uint8 [3,H,W] -> bytestream_list holding the rANS container (M streams; wide = 1 / True: 128 lanes per stream instead of 64; wide = 2: 256 lanes,
    "xwide", v4 layout; auto (xwide only): M is the size rule's count and the encoder picks the image's own from its last stage, orc_auto_streams)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    cap = 8 * H * W + 4096 + 1024 * (2 * M if auto else M)
    out = np.empty(cap, np.uint8)
    seg = np.zeros(49, np.int32)
    n = lib().orc_encode_image_rans(_p(rgb), H, W, C.byref(weights.c), int(M) | (int(wide) << 8) | (0x1000 if auto else 0), _p(out), C.c_long(cap), _p(seg))
    if n < 0:
        raise RuntimeError(f"orc_encode_image_rans failed: {n}")
    return _segments_to_list(out[:n], seg)


def decode_image_rans(bl, weights: "Weights"):
    buf, seg = _list_to_segments(bl)
    H, W = C.c_int(), C.c_int()
    lib().orc_header_dims(_p(buf), _p(seg), C.byref(H), C.byref(W))
    rgb = np.empty((3, H.value, W.value), np.uint8)
    rc = lib().orc_decode_image_rans(_p(buf), _p(seg), C.byref(weights.c), _p(rgb), C.c_long(rgb.size), C.byref(H), C.byref(W))
    if rc != 0:
        raise RuntimeError(f"orc_decode_image_rans failed: {rc}")
    return rgb


def encode_image(rgb, weights: Weights, full_tables=False):
    """uint8 [3,H,W] -> bytestream_list (reference container)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    cap = 8 * H * W + 4096
    out = np.empty(cap, np.uint8)
    seg = np.zeros(49, np.int32)
    n = lib().orc_encode_image(_p(rgb), H, W, C.byref(weights.c), int(full_tables), _p(out), C.c_long(cap), _p(seg))
    if n < 0:
        raise RuntimeError(f"orc_encode_image failed: {n}")
    return _segments_to_list(out[:n], seg)


def decode_image(bl, weights: Weights, full_tables=False):
    buf, seg = _list_to_segments(bl)
    H, W = C.c_int(), C.c_int()
    lib().orc_header_dims(_p(buf), _p(seg), C.byref(H), C.byref(W))
    rgb = np.empty((3, H.value, W.value), np.uint8)
    rc = lib().orc_decode_image(_p(buf), _p(seg), C.byref(weights.c), int(full_tables), _p(rgb),
                                C.c_long(rgb.size), C.byref(H), C.byref(W))
    if rc != 0:
        raise RuntimeError(f"orc_decode_image failed: {rc}")
    return rgb
