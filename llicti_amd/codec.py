"""Host side of the HIP codec: device memory, streams and container plumbing on torch-ROCm tensors.

`HipCodec` owns one C-ABI context (one per GPU / process) and exposes
  - the kernel-level entry points on torch tensors (used by the parity tests), and
  - `encode(rgb_u8[B,3,H,W]) -> (containers, seg_len)` / `decode(...)`, which keep the containers in HBM,
  - helpers turning a device container into the reference's `bytestream_list` and back.
PyTorch is plumbing here (allocation, streams, host copies); every computation happens in the HIP
library.  Nothing in this module imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .weights import pack_state_dict

NSEG = 49
MODE_AC = 0                      # the reference's container: 45 torchac-algorithm streams per image


# ---- container "auto": a PURE FUNCTION OF THE IMAGE'S SIZE (round 6).  Round 5's rule also looked at the batch size, the compute-unit count and -- through a
# running mean the model kept -- at what had been coded before, so an image's bytes depended on eval_batch, on the coding order and on the rank count
# (ADVICE r5): a sharded run's log differed from the one-rank run's on exactly the content the reference exists for.  Now an image
# of H x W always gets the same container whatever it is coded with, next to or after.
STREAM_BYTES_BUDGETED = 4.7      # what an xwide v4 stream is budgeted at: measured with the oracle on natural-like and model-drawn images of 321x481 ... 768x512, a container
                                 # of M streams is -16 ... -24 + (4.0 ... 4.9) M bytes larger than the reference-format one (2.3-2.8 per stream on noise, 5.1 on a 1.5-bit
                                 # source; tests/sim_v4.py; 1.8 of them are the 256 lanes' 0.057 bit each)
AC_TERMINATION_BYTES = 22.5      # ... the offset: what the reference format spends on its 45 range-coder terminations, less the 1.5 bytes the tables differ by
XWIDE_MIN_SHARE = 6144           # last-stage symbols per stream: the tail fills the 7,936-bit payload the initial states carry from 1.3 bits per symbol up (the reference's
                                 # trained model on natural images: 1.68, exp_debug.log.1:2682; a source cheaper than that leaves payload unused: ~1 byte per missing 8 bits)
NARROW_MIN_TAIL = 512            # ... and a 64-lane stream its 1,984 bits
LENGTH_TABLE_BYTES = 4.0         # a stream of a 64- / 128-stream container pays a u32 in its segment's length table


def last_stage_bits(seg_len, H, W):
    """Bits per symbol the LAST stage's Cg stream (segment 48: level 0, band x10) of an image in the REFERENCE-FORMAT container took
    (bench.py reports it for its content classes).  seg_len: the image's 49 segment lengths."""
    return 8.0 * float(seg_len[NSEG - 1]) / max(1, (H // 2) * (W // 2))


def image_streams(H, W):
    """xwide v4 streams of an H x W image in container "auto" -- a function of the size alone.  Two limits.  BYTES: M streams cost about
    STREAM_BYTES_BUDGETED M over the ideal code length, the reference format's terminations AC_TERMINATION_BYTES, and the north star's 0.001 bpp
    are H W / 8000 bytes (tests/test_oracle_golden.py::test_auto_container_budget_by_size holds the oracle's sizes against it).  PAYLOAD: a
    stream's 256 initial states carry 992 bytes that only its own share of the LAST stage's symbols can fill.  768x512: 15; 0: no xwide stream
    fits (auto_container falls back to one 64-lane stream or the reference format).  More than 32 streams come as 64 or 128 (two / four per
    container segment behind a table of their lengths)."""
    nc_last = (H // 2) * (W // 2)                 # coded positions of level 0, band x10
    budget = H * W / 8000.0 + AC_TERMINATION_BYTES
    m = min(int(budget / STREAM_BYTES_BUDGETED), nc_last // XWIDE_MIN_SHARE)
    if m > 32:
        for big in (128, 64):
            if big * (STREAM_BYTES_BUDGETED + LENGTH_TABLE_BYTES) <= budget and nc_last // XWIDE_MIN_SHARE >= big:
                return big
        m = 32
    return max(0, m)


def image_mode(H, W, mixed=False):
    """ENCODER mode of ONE image in container "auto": MODE_RANS_AUTO(image_streams(H, W)) -- xwide v4 streams, their count picked by the encoder from
    the image itself: what its size gives, a third more where the last stage's symbols are expensive (an xwide stream costs ~2.5 bytes there instead
    of ~4.5), half where the last stage is too cheap to fill the streams' payloads; an image too small for one xwide stream gets a 64-lane stream
    (from ~45x45 pixels) or -- only where the call holds images of one size (`mixed` False) -- the reference format."""
    m = image_streams(H, W)
    if m > 32:
        return MODE_RANS(m, wide=2)             # (64 / 128 streams: very large images; the count is the size rule's)
    if m >= 1:
        return MODE_RANS_AUTO(m)
    if (H // 2) * (W // 2) >= NARROW_MIN_TAIL or mixed:
        return MODE_RANS(1)
    return MODE_AC


def auto_container(H, W):
    """Name of the "auto" ENCODER mode of an H x W image (image_mode): "xauto15" for 768x512 -- the container that comes out is xrans8, xrans15 or
    xrans20, by the image's content."""
    return name_of_mode(image_mode(H, W))


def auto_modes(sizes):
    """Container modes of the images of ONE call ([(H, W), ...]) in container "auto": each image's own (image_mode) -- xwide streams, their number per image
    a function of its size -- unless one of them is too small for an xwide stream: the images of a call share a lane kind, so then every image of the
    call gets one 64-lane stream (the reference format, where all are that small and of one size).  What an image of a call of EQUAL sizes gets depends
    on its size alone; in a call of mixed sizes a tiny neighbour (below ~90x90 pixels) can push an image to the 64-lane kind -- the reference's test set
    has no such image (tests/golden/eval_shapes.json: 321x481 is its smallest)."""
    mixed = len(set(sizes)) > 1
    modes = [image_mode(h, w, mixed) for h, w in sizes]
    if all(_mode_wide(m) == 2 for m in modes if m != MODE_AC) and MODE_AC not in modes:
        return modes
    if all(m == MODE_AC for m in modes):
        return modes
    return [MODE_RANS(1)] * len(sizes)


def MODE_RANS(M=8, wide=False):
    """rANS container: M independent interleaved rANS streams per image.  wide = 0 / False: 64 lanes per stream, M in 1 .. 32, 64, 128 (v3);
    wide = 1 / True: 128 lanes (two 64-symbol chunks per coder step), M in 1 .. 14 (v3); wide = 2 ("xwide"): 256 lanes, one decoder lane per
    symbol, the v4 stream layout, M in 1 .. 32, 64, 128 (include/llicti_hip.h)."""
    return (0x100 + 0x200 * int(wide)) | int(M)


def MODE_RANS_AUTO(M):
    """ENCODE ONLY: xwide v4 streams whose count the encoder picks per image, on the device, from the image itself (include/llicti_hip.h,
    LLICTI_MODE_RANS_X_AUTO): M = the count the image's size gives (image_streams); expensive last-stage symbols -> M + ceil(M / 3), a last stage
    too cheap to fill M payloads -> ceil(M / 2).  The container is an ordinary MODE_RANS(count, wide=2) one: its header says which (mode_of_header)."""
    return 0x10500 | int(M)


def _mode_auto(mode: int) -> bool:
    return bool(mode & 0x10000)


def auto_counts(M):
    """The stream counts an "auto" encode of size-rule count M may pick (llicti_amd/csrc/host_types.hpp: rans_auto_pick): (a last stage that cannot
    fill the payloads, cheap symbols, default, expensive symbols) = (ceil(M / 2), ceil(2 M / 3), M, min(32, M + ceil(M / 3)))."""
    return (M + 1) // 2, (2 * M + 2) // 3, M, min(32, M + (M + 2) // 3)


def _mode_wide(mode: int) -> int:
    return ((mode & 0xF00) - 0x100) // 0x200


def rans_tag(M, wide=False):
    """Header byte 0 of a rANS container with M streams per image (xwide v4: 0xE8 whatever M -- the count is in the pad field, rans_pad_hi)."""
    wide = int(wide)
    ext = 1 if (M > 32 or wide) else 0
    if wide == 2:
        v = 16
    elif wide == 1:
        v = M + 1
    else:
        v = {64: 0, 128: 1}[M] if ext else M - 1
    return 0x88 | (ext << 6) | (((v >> 3) & 3) << 4) | (v & 7)


def rans_pad_hi(M, wide=False):
    """Bits 10 .. 15 of the header's int16 pad field: the stream count of an xwide v4 container (1 .. 32 as they are, 33 / 34 = 64 / 128 streams),
    zero in every other container."""
    return (M if M <= 32 else {64: 33, 128: 34}[M]) if int(wide) == 2 else 0


def mode_of_header(hdr, pad=None) -> int:
    """Container mode of a header: `hdr` = its 17 bytes (bytes 0 .. 2, the 12 min / max bytes, the int16 pad field), or a bytestream_list, or
    byte 0 alone with the pad field in `pad` (only an xwide v4 container needs it: its stream count lives in the field's bits 10 .. 15)."""
    if isinstance(hdr, (list, tuple)):
        hdr = bytes(hdr[0][0]) + bytes(hdr[0][1]) + bytes(hdr[0][2])
    if isinstance(hdr, (bytes, bytearray, np.ndarray)):
        byte0 = int(hdr[0])
        if len(hdr) >= 17:
            pad = int(hdr[15]) | (int(hdr[16]) << 8)
    else:
        byte0 = int(hdr)
    u = 0 if pad is None else (int(pad) >> 10) & 0x3F
    if byte0 == 5:
        return MODE_AC
    if (byte0 & 0x88) == 0x88:          # rANS: bits 5,4,2,1,0 = v; bit 6 clear: M = v + 1; set: v = 0, 1 -> 64, 128 streams, 2 .. 15 -> v - 1 wide streams,
        v = (((byte0 >> 4) & 3) << 3) | (byte0 & 7)       # 16 -> xwide streams in the v4 layout, their count in the pad field's high bits
        if (byte0 >> 6) & 1:
            if v >= 16:
                if v != 16 or not (1 <= u <= 34):
                    raise ValueError(f"container tag 0x{byte0:02x} with pad field high bits {u}: an xwide container of the retired v3 layout (rounds 4-5), "
                                     "or an xwide v4 tag without its pad field -- this build reads and writes xwide v4")
                return MODE_RANS(u if u <= 32 else {33: 64, 34: 128}[u], wide=2)
            if u:
                raise ValueError("pad field high bits set in a container that is not xwide v4")
            if v <= 1:
                return MODE_RANS(64 << v)
            return MODE_RANS(v - 1, wide=1)
        if u:
            raise ValueError("pad field high bits set in a container that is not xwide v4")
        return MODE_RANS(v + 1)
    if (byte0 & 0x88) == 0x80:
        raise ValueError(f"container tag 0x{byte0:02x} is the retired LLICTI-rANS v2 format; this build reads and writes v3 / v4 only")
    raise ValueError(f"unknown container tag 0x{byte0:02x}")


def mode_of_name(name: str) -> int:
    """"ac" | "rans<M>" | "wrans<M>" (wide streams: 128 lanes) | "xrans<M>" (xwide streams: 256 lanes) -> mode."""
    name = str(name).lower()
    if name == "ac":
        return MODE_AC
    if name.startswith("xauto"):
        return MODE_RANS_AUTO(int(name[5:]))
    if name.startswith("xrans"):
        return MODE_RANS(int(name[5:]), wide=2)
    if name.startswith("wrans"):
        return MODE_RANS(int(name[5:]), wide=1)
    if name.startswith("rans"):
        return MODE_RANS(int(name[4:] or 8))
    raise ValueError(f"unknown container {name!r}: ac, rans<M>, wrans<M>, xrans<M>, xauto<M> (or \"auto\" where the image sizes are known)")


def name_of_mode(mode: int) -> str:
    if _mode_auto(mode):
        return "xauto%d" % (mode & 0xFF)
    return "ac" if mode == MODE_AC else ("rans%d", "wrans%d", "xrans%d")[_mode_wide(mode)] % (mode & 0xFF)


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    raise TypeError(type(t))


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class HipCodec:
    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise _lib.LlictiError(_lib.ENODEVICE, "no GPU visible to PyTorch-ROCm: the LLICTI hot path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.L = _lib.lib()
        ctx = C.c_void_p()
        _lib.check(self.L.llicti_create(C.byref(ctx), self.device.index))
        self.ctx = ctx
        self._ws = None
        self._ws_need = {}
        self._ws_grown = False
        self._mc = {}
        self.have_weights = False

    def close(self):
        if getattr(self, "ctx", None):
            self.L.llicti_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd):
        """Reference-keyed state_dict (or the canonical packed dict) -> device."""
        packed = sd if (isinstance(sd, dict) and 0 in sd) else pack_state_dict(sd)
        for b in range(3):
            d = packed[b]
            _lib.check(self.L.llicti_set_band_weights(self.ctx, b, int(d["K0"]), _ptr(d["w0"]), _ptr(d["b0"]),
                                                      _ptr(d["w1"]), _ptr(d["b1"]), _ptr(d["w2"]), _ptr(d["b2"])))
        self.have_weights = True

    # ------------------------------------------------------------------ kernel-level wrappers
    def lift(self, rgb):
        B, _, H, W = rgb.shape
        planes = torch.empty((B, 3, H, W), dtype=torch.int16, device=self.device)
        fplanes = torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device)
        mm = torch.empty((B, 4), dtype=torch.int32, device=self.device)
        _lib.check(self.L.llicti_lift_u8(self.ctx, _ptr(rgb), B, H, W, _ptr(planes), _ptr(fplanes), _ptr(mm), _stream_ptr(self.device)))
        return planes, fplanes, mm

    def unlift(self, planes):
        B, _, H, W = planes.shape
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=self.device)
        _lib.check(self.L.llicti_unlift_u8(self.ctx, _ptr(planes), B, H, W, _ptr(rgb), _stream_ptr(self.device)))
        return rgb

    def band_params(self, fplanes, lvl, band):
        B, _, H, W = fplanes.shape
        _, _, h, w, _, _, _, _ = _lib.level_geom(H, W, lvl, band)
        out = torch.empty((B, 64, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.L.llicti_band_params_f32(self.ctx, _ptr(fplanes), B, H, W, lvl, band, _ptr(out), _stream_ptr(self.device)))
        return out                             # channel-planar: 4 heads x 16 planes (15 used; the 16th is never written); params60() gives the reference's 60 channels

    def lift_train(self, rgb):
        """uint8 [B,3,H,W] -> float32 [B,3,H,W] planes of the training path's float lift (Y - 127/255, Co, Cg)."""
        B, _, H, W = rgb.shape
        fplanes = torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device)
        _lib.check(self.L.llicti_lift_train_f32(self.ctx, _ptr(rgb), B, H, W, _ptr(fplanes), _stream_ptr(self.device)))
        return fplanes

    def selfinfo(self, fplanes, params, lvl, band):
        """-> float32 [B, 3, h, w]: -log2 pmf (bits) of the band's Y, Co, Cg targets."""
        B, _, H, W = fplanes.shape
        _, _, h, w, _, _, _, _ = _lib.level_geom(H, W, lvl, band)
        out = torch.empty((B, 3, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.L.llicti_selfinfo_f32(self.ctx, _ptr(fplanes), _ptr(params), B, H, W, lvl, band, _ptr(out),
                                              _stream_ptr(self.device)))
        return out

    def forward_selfinfo(self, rgb):
        """LLICTI.forward: list of 5 tensors [B, 9, h, w] (scale 0 first; channel 3*band + colour)."""
        fplanes = self.lift_train(rgb)
        res = []
        for lvl in range(5):
            bands = [self.selfinfo(fplanes, self.band_params(fplanes, lvl, b), lvl, b) for b in range(3)]
            res.append(torch.cat(bands, dim=1))
        return res

    @staticmethod
    def params60(p64):
        """[B, 64, h, w] device layout (channel planes) -> [B, h, w, 60] in the reference's channel order (LLICTI_nets.py:381-387)."""
        B, _, h, w = p64.shape
        return p64.reshape(B, 4, 16, h, w)[:, :, :15].reshape(B, 60, h, w).permute(0, 2, 3, 1)

    def cdf_tables(self, planes, params, mm, lvl, band, clr, row_stride=512):
        B, _, H, W = planes.shape
        *_, hc, wc = _lib.level_geom(H, W, lvl, band)
        out = torch.empty((B, hc * wc, row_stride), dtype=torch.int16, device=self.device)
        _lib.check(self.L.llicti_cdf_u16(self.ctx, _ptr(planes), _ptr(params), _ptr(mm), B, H, W, lvl, band, clr,
                                         _ptr(out), row_stride, _stream_ptr(self.device)))
        return out

    def cdf_pairs(self, planes, params, mm, lvl, band):
        B, _, H, W = planes.shape
        *_, hc, wc = _lib.level_geom(H, W, lvl, band)
        out = torch.empty((3, B, hc * wc), dtype=torch.int32, device=self.device)
        _lib.check(self.L.llicti_cdf_pairs_u32(self.ctx, _ptr(planes), _ptr(params), _ptr(mm), B, H, W, lvl, band,
                                               _ptr(out), _stream_ptr(self.device)))
        return out

    def ac_encode(self, cdf, sym, Lp=None):
        """torchac seam: cdf int16/uint16 [S, N, stride] with Lp valid entries per row (default: stride), sym int16 [S, N]
        -> (streams uint8 [S, out_stride], lengths int32 [S])."""
        S, N, stride = cdf.shape
        Lp = stride if Lp is None else int(Lp)
        out_stride = (2 * N + 32 + 3) // 4 * 4      # slots are written 32 bits at a time
        out = torch.zeros((S, out_stride), dtype=torch.uint8, device=self.device)
        ln = torch.zeros((S,), dtype=torch.int32, device=self.device)
        _lib.check(self.L.llicti_ac_encode_u16cdf(self.ctx, _ptr(cdf), Lp, stride, _ptr(sym), S, N, _ptr(out), out_stride,
                                                  _ptr(ln), _stream_ptr(self.device)))
        _lib.check(self.L.llicti_check_status(self.ctx, _stream_ptr(self.device)))
        return out, ln

    def poison_workspace(self, value=0xA5):
        """Overwrite the cached workspace (tests / bench: a decode must not depend on what an earlier call left there)."""
        if self._ws is not None:
            self._ws.fill_(value)

    def ac_decode(self, cdf, Lp, streams, lens, N):
        S, _, stride = cdf.shape
        in_stride = streams.shape[1]
        sym = torch.empty((S, N), dtype=torch.int16, device=self.device)
        _lib.check(self.L.llicti_ac_decode_u16cdf(self.ctx, _ptr(cdf), int(Lp), stride, _ptr(streams), in_stride, _ptr(lens),
                                                  S, N, _ptr(sym), _stream_ptr(self.device)))
        return sym

    # ------------------------------------------------------------------ whole batch
    def _workspace_of(self, n):
        """The cached workspace, grown to the RUNNING MAXIMUM of what the calls needed (a data set of many image sizes must not
        re-allocate per call; the library validates the size it is given against what the call needs)."""
        if n == 0:
            _lib.check(_lib.EINVAL)
        if self._ws is None or self._ws.numel() < n:
            self._ws = None
            self._ws = torch.empty((int(n * 1.25) if self._ws_grown else n,), dtype=torch.uint8, device=self.device)
            self._ws_grown = True
        return self._ws

    def workspace(self, B, H, W, mode=MODE_AC):
        key = (B, H, W, mode)
        n = self._ws_need.get(key)
        if n is None:
            if len(self._ws_need) > 256:
                self._ws_need.clear()
            n = self._ws_need[key] = int(self.L.llicti_workspace_bytes(B, H, W, mode))
        return self._workspace_of(n)

    @staticmethod
    def _modes_arg(mode, B):
        """mode: one int, or one per image -> (int or None, int32 array or None)"""
        if isinstance(mode, (int, np.integer)):
            return int(mode), None
        m = np.ascontiguousarray(mode, dtype=np.int32)
        assert m.shape == (B,)
        return (int(m[0]), None) if (m == m[0]).all() else (None, m)

    def workspace_v(self, Hs, Ws, mode):
        Hs, Ws = np.ascontiguousarray(Hs, dtype=np.int32), np.ascontiguousarray(Ws, dtype=np.int32)
        one, per = self._modes_arg(mode, len(Hs))
        key = (Hs.tobytes(), Ws.tobytes(), one if per is None else per.tobytes())
        n = self._ws_need.get(key)
        if n is None:
            if len(self._ws_need) > 256:
                self._ws_need.clear()
            n = int(self.L.llicti_workspace_bytes_v(len(Hs), _ptr(Hs), _ptr(Ws), one)) if per is None else \
                int(self.L.llicti_workspace_bytes_vm(len(Hs), _ptr(Hs), _ptr(Ws), _ptr(per)))
            self._ws_need[key] = n
        return self._workspace_of(n)

    def max_container_bytes(self, H, W):
        n = self._mc.get((H, W))
        if n is None:
            n = int(self.L.llicti_max_container_bytes(H, W))
            if n == 0:
                _lib.check(_lib.EINVAL)
            if len(self._mc) > 1024:
                self._mc.clear()
            self._mc[(H, W)] = n
        return n

    def encode(self, rgb, mode=MODE_AC, out=None, seg_len=None):
        """rgb uint8 [B,3,H,W] on this device -> (containers uint8 [B, stride], seg_len int32 [B,49]), async."""
        assert rgb.dtype == torch.uint8 and rgb.is_cuda and rgb.dim() == 4 and rgb.shape[1] == 3
        rgb = rgb.contiguous()
        B, _, H, W = rgb.shape
        ws = self.workspace(B, H, W, mode)
        stride = self.max_container_bytes(H, W)
        if out is None:
            out = torch.empty((B, stride), dtype=torch.uint8, device=self.device)
        if seg_len is None:
            seg_len = torch.zeros((B, NSEG), dtype=torch.int32, device=self.device)
        _lib.check(self.L.llicti_encode_images(self.ctx, _ptr(rgb), B, H, W, mode, _ptr(ws), ws.numel(), _ptr(out), out.shape[1],
                                               _ptr(seg_len), _stream_ptr(self.device)))
        return out, seg_len

    def decode(self, containers, seg_len, H, W, mode=MODE_AC, out=None):
        """device containers -> uint8 [B,3,H,W], async."""
        B = containers.shape[0]
        ws = self.workspace(B, H, W, mode)
        if out is None:
            out = torch.empty((B, 3, H, W), dtype=torch.uint8, device=self.device)
        _lib.check(self.L.llicti_decode_images(self.ctx, _ptr(containers), containers.shape[1], _ptr(seg_len), B, H, W, mode,
                                               _ptr(ws), ws.numel(), _ptr(out), _stream_ptr(self.device)))
        return out

    # ---- batches of mixed sizes (llicti_encode_images_v / llicti_decode_images_v): images tightly packed in one flat uint8 buffer
    @staticmethod
    def flat_offsets(Hs, Ws):
        """Byte offsets of the images' [3][H][W] blocks in the flat RGB buffer of a mixed-size batch (tightly packed), and its size."""
        sizes = 3 * np.asarray(Hs, dtype=np.int64) * np.asarray(Ws, dtype=np.int64)
        offs = np.concatenate(([0], np.cumsum(sizes)))
        return offs[:-1], int(offs[-1])

    def encode_v(self, rgb_flat, Hs, Ws, mode, out=None, seg_len=None):
        """rgb_flat: uint8 device tensor holding B images of sizes Hs[b] x Ws[b] back to back ([3][H][W] each) -> (containers uint8
        [B, stride], seg_len int32 [B, 49]), async.  Image b's bytes are those of encode() on that image alone."""
        assert rgb_flat.dtype == torch.uint8 and rgb_flat.is_cuda and rgb_flat.dim() == 1 and rgb_flat.is_contiguous()
        Hs, Ws = np.ascontiguousarray(Hs, dtype=np.int32), np.ascontiguousarray(Ws, dtype=np.int32)
        B = len(Hs)
        assert rgb_flat.numel() >= self.flat_offsets(Hs, Ws)[1]
        ws = self.workspace_v(Hs, Ws, mode)
        stride = max(self.max_container_bytes(int(h), int(w)) for h, w in set(zip(Hs.tolist(), Ws.tolist())))
        if out is None:
            out = torch.empty((B, stride), dtype=torch.uint8, device=self.device)
        if seg_len is None:
            seg_len = torch.zeros((B, NSEG), dtype=torch.int32, device=self.device)
        assert out.shape[1] >= stride
        one, per = self._modes_arg(mode, B)         # one mode for the call, or one per image (stream counts may differ: llicti_encode_images_vm)
        if per is None:
            _lib.check(self.L.llicti_encode_images_v(self.ctx, _ptr(rgb_flat), None, B, _ptr(Hs), _ptr(Ws), one, _ptr(ws), ws.numel(),
                                                     _ptr(out), out.shape[1], _ptr(seg_len), _stream_ptr(self.device)))
        else:
            _lib.check(self.L.llicti_encode_images_vm(self.ctx, _ptr(rgb_flat), None, B, _ptr(Hs), _ptr(Ws), _ptr(per), _ptr(ws), ws.numel(),
                                                      _ptr(out), out.shape[1], _ptr(seg_len), _stream_ptr(self.device)))
        return out, seg_len

    def decode_v(self, containers, seg_len, Hs, Ws, mode, out=None):
        """device containers of B images of sizes Hs[b] x Ws[b] -> flat uint8 device tensor (the images back to back), async."""
        Hs, Ws = np.ascontiguousarray(Hs, dtype=np.int32), np.ascontiguousarray(Ws, dtype=np.int32)
        B = len(Hs)
        assert containers.shape[0] == B
        ws = self.workspace_v(Hs, Ws, mode)
        total = self.flat_offsets(Hs, Ws)[1]
        if out is None:
            out = torch.empty((total,), dtype=torch.uint8, device=self.device)
        assert out.numel() >= total
        one, per = self._modes_arg(mode, B)
        if per is None:
            _lib.check(self.L.llicti_decode_images_v(self.ctx, _ptr(containers), containers.shape[1], _ptr(seg_len), B, _ptr(Hs), _ptr(Ws), one,
                                                     _ptr(ws), ws.numel(), _ptr(out), None, _stream_ptr(self.device)))
        else:
            _lib.check(self.L.llicti_decode_images_vm(self.ctx, _ptr(containers), containers.shape[1], _ptr(seg_len), B, _ptr(Hs), _ptr(Ws), _ptr(per),
                                                      _ptr(ws), ws.numel(), _ptr(out), None, _stream_ptr(self.device)))
        return out

    def container_modes(self, containers):
        """The modes the headers of device containers [B, stride] name (a small download: 17 bytes per image; synchronises the current stream) --
        what decode() / decode_v() take for containers that came out of an "auto" encode (MODE_RANS_AUTO: the encoder picked the stream counts)."""
        hdr = containers[:, :17].contiguous().cpu().numpy()
        return [mode_of_header(hdr[b]) for b in range(hdr.shape[0])]

    def check(self):
        _lib.check(self.L.llicti_check_status(self.ctx, _stream_ptr(self.device)))

    def image_status(self, B):
        """Per-image status of the last decode: numpy int32 [B], 0 = ok, EFORMAT = that image's container was malformed."""
        out = np.zeros(B, dtype=np.int32)
        _lib.check(self.L.llicti_image_status(self.ctx, _ptr(out), B, _stream_ptr(self.device)))
        return out

    def set_tuning(self, key, value):
        _lib.check(self.L.llicti_set_tuning(self.ctx, key.encode(), int(value)))

    def encoded_fplanes(self, B, H, W, mode):
        """float32 [B,3,H,W] = (YCoCg-R - [127,0,0]) / 255 of the batch the LAST encode() / decode() of this shape and mode worked on, copied out of
        the workspace (llicti_workspace_planes): the `x_ycocg` of the reference's compress() without a second lift.  Enqueued on the
        current stream, behind that call."""
        o16, o32 = C.c_size_t(), C.c_size_t()
        _lib.check(self.L.llicti_workspace_planes(self.ctx, B, H, W, mode, C.byref(o16), C.byref(o32)))
        n = B * 3 * H * W * 4
        return self._ws[o32.value:o32.value + n].view(torch.float32).view(B, 3, H, W).clone()

    def last_params_v(self, Hs, Ws, mode, image):
        """float32 [64, h*w] CNN outputs of level 0, band x10 -- the LAST band-CNN launch -- of image `image` of the last encode_v() / decode_v() on
        images of these sizes in `mode` (one, or one per image), copied out of the workspace (llicti_workspace_params_v); params60() of its [1, 64, h, w]
        view gives the reference's 60 channels."""
        Hs, Ws = np.ascontiguousarray(Hs, dtype=np.int32), np.ascontiguousarray(Ws, dtype=np.int32)
        one, per = self._modes_arg(mode, len(Hs))
        modes = np.full(len(Hs), one, dtype=np.int32) if per is None else per
        off, npos = C.c_size_t(), C.c_long()
        _lib.check(self.L.llicti_workspace_params_v(self.ctx, len(Hs), _ptr(Hs), _ptr(Ws), _ptr(modes), int(image), C.byref(off), C.byref(npos)))
        n = 64 * npos.value * 4
        return self._ws[off.value:off.value + n].view(torch.float32).view(64, npos.value).clone()

    def counter(self, name):
        """llicti_get_counter: "device_syncs", "device_allocs", "plan_builds", "plan_hits", "block_waits", "plans_cached", "blocks_pooled"."""
        v = C.c_long()
        _lib.check(self.L.llicti_get_counter(self.ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def set_profiling(self, on=True):
        _lib.check(self.L.llicti_set_profiling(self.ctx, int(bool(on))))

    PROF_CATS = ("cnn", "rans_stage", "rans_tail", "cdf_pairs", "rans_encode", "ac", "misc")

    def last_timing_detail(self):
        """-> ({kernel group: ms}, [ms of every band-CNN launch, scale 4..0 x band 0..2]) of the last whole-batch call."""
        cat = (C.c_float * len(self.PROF_CATS))()
        per = (C.c_float * 64)()
        n = C.c_int()
        _lib.check(self.L.llicti_last_timing_detail(self.ctx, cat, per, 64, C.byref(n)))
        return dict(zip(self.PROF_CATS, [float(v) for v in cat])), [float(per[i]) for i in range(min(n.value, 64))]

    def last_cnn_level_ms(self):
        """-> [ms of the band-CNN launches of level 0 .. 4] of the last whole-batch call (profiling on)."""
        ms = (C.c_float * 5)()
        _lib.check(self.L.llicti_last_cnn_level_ms(self.ctx, ms))
        return [float(v) for v in ms]

    def last_timing(self):
        ms = (C.c_float * 4)()
        n = C.c_int()
        _lib.check(self.L.llicti_last_timing(self.ctx, ms, C.byref(n)))
        return list(ms), n.value


# ---------------------------------------------------------------------- container <-> bytestream_list
def container_to_bytestream_list(buf: np.ndarray, seg_len: np.ndarray):
    """Flat container of one image -> the reference's list of 6 lists x 9 `bytes`
    (LLICTI_nets.py:352-354, :411; loggers/rate.py:133 needs 9 entries per row)."""
    segs, pos = [], 0
    for n in seg_len:
        segs.append(bytes(buf[pos:pos + int(n)]))
        pos += int(n)
    em = b""
    bl = [[segs[0], segs[1], segs[2], segs[3], em, em, em, em, em]]
    for s in range(5):
        bl.append(segs[4 + 9 * s: 4 + 9 * (s + 1)])
    return bl


def bytestream_list_to_container(bl):
    if len(bl) != 6 or any(len(r) != 9 for r in bl):
        raise ValueError("bytestream_list must be 6 lists of 9 byte strings")
    segs = list(bl[0][:4])
    for s in range(1, 6):
        segs += list(bl[s])
    seg_len = np.array([len(s) for s in segs], dtype=np.int32)
    return np.frombuffer(b"".join(segs), dtype=np.uint8).copy(), seg_len


def header_dims(hdr17: bytes):
    H, W = C.c_int(), C.c_int()
    buf = (C.c_uint8 * 17).from_buffer_copy(bytes(hdr17[:17]).ljust(17, b"\0"))
    _lib.check(_lib.lib().llicti_header_dims(buf, C.byref(H), C.byref(W)))
    return H.value, W.value
