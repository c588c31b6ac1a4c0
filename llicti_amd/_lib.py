"""ctypes binding of libllicti_hip.so (the C-ABI in include/llicti_hip.h).

The library is built in-tree (`python -m llicti_amd._lib` or `__graft_entry__.build()`); there is no
fallback of any kind: if it is missing or no gfx950 device is present, loading / creating a context
raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SO_PATH = os.environ.get("LLICTI_HIP_SO") or os.path.join(_HERE, "libllicti_hip.so")   # env override: A/B builds
_CSRC = os.path.join(_HERE, "csrc")
SOURCES = [os.path.join(_CSRC, "llicti_hip.hip")] + sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".hpp")) + \
          [os.path.join(ROOT, "include", "llicti_hip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-value"]

OK, EINVAL, EHIP, ENOWEIGHTS, ENOSPACE, EFORMAT, ENODEVICE = 0, -1, -2, -3, -4, -5, -6


class LlictiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"llicti_hip error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    stale = force or not os.path.exists(SO_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in SOURCES)
    if stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + HIPCC_FLAGS + ["-o", SO_PATH, SOURCES[0]]
        subprocess.check_call(cmd)
    return SO_PATH


_lib = None
_vp, _i, _l, _sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t
_SIGS = {
    "llicti_last_error": (C.c_char_p, []),
    "llicti_version": (C.c_char_p, []),
    "llicti_create": (_i, [C.POINTER(_vp), _i]),
    "llicti_destroy": (_i, [_vp]),
    "llicti_set_band_weights": (_i, [_vp, _i, _i] + [_vp] * 6),
    "llicti_level_geom": (_i, [_i, _i, _i, _i] + [C.POINTER(_i)] * 8),
    "llicti_lift_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "llicti_unlift_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "llicti_band_params_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "llicti_lift_train_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "llicti_selfinfo_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "llicti_cdf_u16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "llicti_cdf_pairs_u32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "llicti_ac_encode_u16cdf": (_i, [_vp, _vp, _i, _i, _vp, _i, _l, _vp, _l, _vp, _vp]),
    "llicti_ac_decode_u16cdf": (_i, [_vp, _vp, _i, _i, _vp, _l, _vp, _i, _l, _vp, _vp]),
    "llicti_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "llicti_workspace_bytes_v": (_sz, [_i, _vp, _vp, _i]),
    "llicti_workspace_bytes_vm": (_sz, [_i, _vp, _vp, _vp]),
    "llicti_encode_images_vm": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "llicti_decode_images_vm": (_i, [_vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "llicti_encode_images_v": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp, _vp]),
    "llicti_decode_images_v": (_i, [_vp, _vp, _sz, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "llicti_max_container_bytes": (_sz, [_i, _i]),
    "llicti_encode_images": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _sz, _vp, _vp]),
    "llicti_decode_images": (_i, [_vp, _vp, _sz, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "llicti_check_status": (_i, [_vp, _vp]),
    "llicti_header_dims": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "llicti_header_mode": (_i, [_vp, C.POINTER(_i)]),
    "llicti_image_status": (_i, [_vp, _vp, _i, _vp]),
    "llicti_selftest": (_i, []),
    "llicti_last_timing": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(_i)]),
    "llicti_last_timing_detail": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "llicti_last_cnn_level_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "llicti_set_profiling": (_i, [_vp, _i]),
    "llicti_get_counter": (_i, [_vp, C.c_char_p, C.POINTER(_l)]),
    "llicti_workspace_planes": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "llicti_workspace_params_v": (_i, [_vp, _i, _vp, _vp, _vp, _i, C.POINTER(_sz), C.POINTER(_l)]),
    "llicti_set_tuning": (_i, [_vp, C.c_char_p, _i]),
}
EXPORTS = sorted(_SIGS)


def lib():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64; import it first so that this library binds to the HIP
        # runtime already in the process (same SONAME) instead of pulling a second copy from /opt/rocm --
        # two runtimes in one process do not share devices, streams or allocations.
        import torch  # noqa: F401
        if not os.path.exists(SO_PATH):
            raise LlictiError(ENODEVICE, f"{SO_PATH} is missing: run __graft_entry__.build() (hipcc); "
                                         "there is no CPU fallback for the LLICTI hot path")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)      # AttributeError here = the library does not export the header's symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise LlictiError(rc, lib().llicti_last_error().decode(errors="replace"))


def level_geom(H, W, lvl, band=0):
    v = [_i() for _ in range(8)]
    check(lib().llicti_level_geom(H, W, lvl, band, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)   # Hl, Wl, h, w, padH, padW, hc, wc


if __name__ == "__main__":
    print(build(force=True))
