"""Drop-in for the reference's `graphs.models.LLICTI_nets.LLICTI` on the encode/decode path.

Same constructor, `nn.Module` surface (33 `state_dict` entries with the reference's key names, so
`load_state_dict(ckpt['state_dict'])` works -- agents/base.py:51-76), `compress(x)` and `decompres(...)`
(sic: the missing "s" is part of the reference API, LLICTI_nets.py:161) with the reference's
`bytestream_list` container.  All computation is done by the gfx950 HIP library through
`llicti_amd.codec.HipCodec`; the torch modules below only hold parameters.  There is no CPU path:
calling compress/decompres without a GPU raises.

Differences from the reference, by design:
  * fp32 arithmetic follows the numerics spec (same bits on every launch shape / device), so
    decode(encode(x)) == x by construction, where the reference relies on PyTorch determinism;
  * `compress` accepts B >= 1 only through `compress_batch` (the reference's decoder assumes B == 1,
    LLICTI_nets.py:429, :443); `compress` itself keeps the B == 1 contract.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ...codec import (MODE_AC, MODE_RANS, NSEG, HipCodec, auto_container, bytestream_list_to_container, container_to_bytestream_list,
                      header_dims, mode_of_header, mode_of_name)
from ...config import check_supported


class _LowerBound(nn.Module):
    """Parameter-less stand-in that only carries compressai's `bound` buffer name (state_dict parity)."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


class _CondProbModel(nn.Module):
    """Buffer names of GaussianConditionalLosslessGMM (entropy_layer_nets.py:145-158)."""

    def __init__(self):
        super().__init__()
        self.likelihood_lower_bound = _LowerBound(1e-9)
        self.lower_bound_scale = _LowerBound(0.11 / 255.0)
        self.lower_bound_weights = _LowerBound(1e-6)


class LLICTIEntropyModel4(nn.Module):
    """Parameter container of one band interpolator (LLICTI_nets.py:585-712, config A branch).
    Construction order matches the reference so a seeded default init gives the same weights."""

    def __init__(self, scale, band, config, Ev, Od, Ch, Ly):
        super().__init__()
        self.band = band
        self.num_mixtures = config.num_mixtures
        self.conditional_prob_model = _CondProbModel()
        grps = 4
        Ch = grps * Ch
        Co = (3 * self.num_mixtures) * 3 + (1 + 2) * self.num_mixtures
        c = 3
        if band == 0:
            self.layer0_00_11 = nn.Conv2d(c, Ch, kernel_size=(Ev, Ev))
        if band == 1:
            self.layer0_00_01 = nn.Conv2d(c, Ch, kernel_size=(Od, Ev))
            self.layer0_11_01 = nn.Conv2d(c, Ch, kernel_size=(Ev, Od))
        if band == 2:
            self.layer0_00_10 = nn.Conv2d(c, Ch, kernel_size=(Ev, Od))
            self.layer0_11_10 = nn.Conv2d(c, Ch, kernel_size=(Od, Ev))
            self.layer0_01_10 = nn.Conv2d(c, Ch, kernel_size=(Ev, Ev))
        layers = []
        for _ in range((Ly - 1) - 1):
            layers.append(nn.Conv2d(Ch, Ch, kernel_size=1, groups=grps))
            layers.append(nn.ReLU(inplace=True))
        layers.append(nn.Conv2d(Ch, Co, kernel_size=1, groups=grps))
        self.layers1toL = nn.Sequential(*layers)


class LLICTIEntropyLayer(nn.Module):
    """LLICTI_nets.py:255-316 with useprevlevNN=[F,T,T,T,T]: one scale entry of three band models,
    shared by all five levels."""

    def __init__(self, config):
        super().__init__()
        self.list_scales = config.dwtlevels
        self.num_scales = len(self.list_scales)
        bands = nn.ModuleList()
        for b in range(3):
            bands.append(LLICTIEntropyModel4(scale=0, band=b, config=config, Ev=config.Evens[0], Od=config.Odds[0],
                                             Ch=config.chs[0], Ly=config.conv_layers))
        self.entmdls_scale_band = nn.ModuleList([bands])


class LLICTI(nn.Module):
    """(L)earned (L)ossless (I)mage (C)ompression (T)hrough (I)nterpolation -- MI355X hot path."""

    def __init__(self, config):
        super().__init__()
        check_supported(config)
        self.config = config
        self.ycocg = config.ycocg
        self.clrchs = config.clrchs
        self.clrjnt = config.clr_joint_mode
        self.list_scales = config.dwtlevels
        self.num_scales = len(self.list_scales)
        self.entropymodel = LLICTIEntropyLayer(config)
        self._codec = None
        self._weights_version = None
        # container written by compress(): the reference's (torchac-compatible) one unless the config asks for a throughput container,
        # e.g. config.container = "xrans9", or "auto": the fastest one inside the north star's 0.001 bpp for the batch size of the call
        # (llicti_amd.codec.auto_container: xwide rANS streams, one decoder workgroup per stream and compute unit)
        self.container = str(config["container"]) if "container" in config else "ac"
        self.mode = None if self.container == "auto" else mode_of_name(self.container)
        self._stage = {}                # pinned host staging buffers of the batched path, by (tag, shape)
        self._xfer = {}                 # (upload, download) copy streams of the batched path, by device index

    # ------------------------------------------------------------------ plumbing
    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def codec(self, device=None) -> HipCodec:
        if self._codec is None:
            dev = device
            if dev is None:
                p = next(self.parameters())
                dev = p.device if p.is_cuda else None
            self._codec = HipCodec(dev)
        key = self._weights_key()
        if key != self._weights_version:
            self._codec.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._weights_version = key
        return self._codec

    @torch.no_grad()
    def forward(self, x):
        """Validation likelihood (LLICTI_nets.py:101-123): x float32 [B,3,H,W] in {k/255} (or uint8), H and W
        multiples of 32 (lazyDWT(pad=False) needs equal sub-band sizes; the reference's validate() pads first,
        llicti_agent.py:105-113) -> list of 5 tensors [B, 9, h, w] of -log2 pmf in bits, scale 0 first,
        channel = 3 * band + colour.  Inference only: the kernels do not produce gradients (training stays
        outside this package)."""
        assert x.dim() == 4 and x.shape[1] == 3
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError("forward() needs H and W to be multiples of 32 (pad first, as llicti_agent.py:105-113 does)")
        codec = self.codec(x.device if x.is_cuda else None)
        return codec.forward_selfinfo(self._to_u8(x).to(codec.device).contiguous())

    @staticmethod
    def _to_u8(x):
        # dataloader tensors are uint8/255 as float32 (image_dl.py ToTensor); round(x*255) is the
        # reference's own first step (LLICTI_nets.py:65)
        if x.dtype == torch.uint8:
            return x
        return (x * 255).round().clamp_(0, 255).to(torch.uint8)

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def compress(self, x):
        """x: float32 [1,3,H,W] in {k/255} (or uint8) -> (bytestream_list, x_ycocg)  (LLICTI_nets.py:125-159)."""
        assert x.dim() == 4 and x.shape[1] == 3     # ensure x has 3 colour channels (LLICTI_nets.py:63)
        if x.shape[0] != 1:
            raise ValueError("compress() codes one image (the reference's decoder assumes B == 1); use compress_batch()")
        lists, x_ycocg = self.compress_batch(x)
        return lists[0], x_ycocg

    def mode_for_batch(self, B, device=None):
        """Container mode of a call with B images: the configured one, or for "auto" the throughput container for that batch size."""
        if self.mode is not None:
            return self.mode
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        return mode_of_name(auto_container(B, torch.cuda.get_device_properties(dev).multi_processor_count))

    def _pinned(self, tag, shape, dtype):
        key = (tag, tuple(shape), dtype)
        t = self._stage.get(key)
        if t is None:
            if len(self._stage) >= 16:
                self._stage.clear()
            t = torch.empty(shape, dtype=dtype).pin_memory()
            self._stage[key] = t
        return t

    @torch.no_grad()
    def compress_batch(self, x):
        """x: [B,3,H,W] float32 in {k/255} or uint8 -> (list of B bytestream_lists, x_ycocg).  Synchronous."""
        enc = self.encode_batch_async(x, want_ycocg=True)
        return enc.lists(), enc.x_ycocg

    @torch.no_grad()
    def encode_batch_async(self, x, want_ycocg=False, slot=0):
        """Enqueue the encode of a batch and the download of its containers (pinned host buffers, only the bytes in use); returns an
        EncodedBatch whose lists() waits for the download and cuts the containers into the reference's bytestream_lists.  Nothing here
        blocks the host: a caller can enqueue the next batch before it converts this one (LLICTIAgent.eval_model with eval_batch > 1).
        `slot` selects one of the staging buffer sets (two batches in flight need two)."""
        codec = self.codec(x.device if x.is_cuda else None)
        cur = torch.cuda.current_stream(codec.device)
        up, down = self._copy_streams(codec.device)
        xu = self._to_u8(x)
        if xu.is_cuda:
            rgb = xu.to(codec.device).contiguous()
        else:
            # host input (pinned by the caller for a true async copy): uploaded on the copy stream, so that it runs under whatever the
            # compute stream is doing (the previous batch's decode); the compute stream waits for it, nothing else does
            with torch.cuda.stream(up):
                rgb = xu.to(codec.device, non_blocking=True).contiguous()
            cur.wait_stream(up)
            rgb.record_stream(cur)
        B, _, H, W = rgb.shape
        mode = self.mode_for_batch(B, codec.device)
        cont, seg = codec.encode(rgb, mode=mode)
        x_ycocg = codec.lift(rgb)[1] if want_ycocg else None     # x_ycocg = (YCoCg - [127,0,0]) / 255 (LLICTI_nets.py:143-144)
        seg_h = self._pinned(("seg", slot), (B, NSEG), torch.int32)
        cont_h = self._pinned(("cont_out", slot), tuple(cont.shape), torch.uint8)
        # the download runs on its own stream behind the encode: the compute stream goes straight on to the next enqueued call
        down.wait_stream(cur)
        with torch.cuda.stream(down):
            seg_h.copy_(seg, non_blocking=True)
            cont_h.copy_(cont, non_blocking=True)                  # one contiguous copy of the container strides
            ev = torch.cuda.Event()
            ev.record(down)
        cont.record_stream(down)
        seg.record_stream(down)
        return EncodedBatch(codec, rgb, cont_h, seg_h, ev, x_ycocg, mode)

    def _copy_streams(self, device):
        """(upload, download) HIP streams of a device for the batched calls' transfers (created once): PCIe copies next to the kernels, not
        between them -- on one stream they cost a 24-image encode + decode 2.6 ms of a compute queue that is otherwise never idle."""
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        if key not in self._xfer:
            self._xfer[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        return self._xfer[key]

    @torch.no_grad()
    def decompres(self, bytestream_list, devc=None, xorg=None):
        """bytestream_list -> float32 [1,3,H,W] (LLICTI_nets.py:161-179)."""
        out = self.decompres_batch([bytestream_list], devc)
        return out

    @torch.no_grad()
    def decompres_batch(self, lists, devc=None):
        rgb = self.decode_batch_async(lists, devc)
        self.codec().check()
        return rgb.to(torch.float32) / 255           # LLICTI_nets.py:87

    @torch.no_grad()
    def decode_batch_async(self, lists, devc=None, slot=0):
        """bytestream_lists of B same-size images -> uint8 [B,3,H,W] on the device, enqueued (upload from a pinned buffer + decode);
        device-side failures are reported by codec().check() / image_status()."""
        codec = self.codec(devc if (devc is not None and torch.device(devc).type == "cuda") else None)
        dims = None
        for bl in lists:
            if len(bl) != 6 or any(len(r) != 9 for r in bl):
                raise ValueError("bytestream_list must be 6 lists of 9 byte strings")
            if len(bl[0][0]) != 3 or len(bl[0][1]) != 12 or len(bl[0][2]) != 2:
                raise ValueError("malformed header streams")
            hdr = bytes(bl[0][0]) + bytes(bl[0][1]) + bytes(bl[0][2])
            mode = mode_of_header(hdr[0])               # AC container: hdr[0] == num_scales (LLICTI_nets.py:424)
            d = header_dims(hdr) + (mode,)
            if dims is None:
                dims = d
            elif d != dims:
                raise ValueError("all images of one decompres_batch call must have the same size and container")
        H, W, mode = dims
        B = len(lists)
        stride = codec.max_container_bytes(H, W)
        cont_h = self._pinned(("cont_in", slot), (B, stride), torch.uint8)
        seg_h = self._pinned(("seg_in", slot), (B, NSEG), torch.int32)
        cont_np, seg_np = cont_h.numpy(), seg_h.numpy()
        for i, bl in enumerate(lists):
            pos, k = 0, 0
            for r, row in enumerate(bl):
                for s in (row[:4] if r == 0 else row):
                    n = len(s)
                    if pos + n > stride:
                        raise ValueError("container larger than any valid stream set for this image size")
                    if n:
                        cont_np[i, pos:pos + n] = np.frombuffer(s, dtype=np.uint8)
                    seg_np[i, k] = n
                    pos += n
                    k += 1
        cur = torch.cuda.current_stream(codec.device)
        up, _ = self._copy_streams(codec.device)
        with torch.cuda.stream(up):
            cont_d = cont_h.to(codec.device, non_blocking=True)  # one contiguous copy; bytes past a container's own length are never read (validated lengths)
            seg_d = seg_h.to(codec.device, non_blocking=True)
        cur.wait_stream(up)
        cont_d.record_stream(cur)
        seg_d.record_stream(cur)
        return codec.decode(cont_d, seg_d, H, W, mode=mode)


class EncodedBatch:
    """Handle of one enqueued batch encode (LLICTI.encode_batch_async): the device input, the pinned host copies of the containers and
    their segment lengths, and the event recorded behind the download."""

    def __init__(self, codec, rgb, cont_h, seg_h, ev, x_ycocg, mode):
        self.codec, self.rgb, self.cont_h, self.seg_h, self.ev = codec, rgb, cont_h, seg_h, ev
        self.x_ycocg, self.mode = x_ycocg, mode
        self._lists = None

    def lists(self, check=True):
        """Wait for the download (NOT for anything enqueued after it) and cut the containers into bytestream_lists (6 lists x 9 `bytes`).
        check=False leaves the device-side status to a later codec.check() (which synchronises the whole stream)."""
        if self._lists is None:
            self.ev.synchronize()
            if check:
                self.codec.check()
            seg_np, cont_np = self.seg_h.numpy(), self.cont_h.numpy()
            self._lists = [container_to_bytestream_list(cont_np[b], seg_np[b]) for b in range(seg_np.shape[0])]
        return self._lists
