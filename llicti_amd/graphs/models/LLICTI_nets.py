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

from ...codec import (MODE_AC, MODE_RANS, NSEG, HipCodec, auto_modes, bytestream_list_to_container, container_to_bytestream_list, header_dims, mode_of_header, mode_of_name)
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
        # container written by compress().  config.container: "ac" = the reference's (torchac-compatible) format -- what compress() returns when the
        # config does not say; "xrans<M>" / "wrans<M>" / "rans<M>" = a rANS container with M streams per image; "auto" = xwide rANS streams (v4), their
        # number a function of the IMAGE'S SIZE alone (llicti_amd.codec.image_streams: 16 for 768x512 -- inside the north star's 0.001 bpp of the
        # reference format), so that an image's bytes do not depend on what it is coded with, next to or after.  LLICTIAgent.eval_model switches a
        # model whose config has NO container key -- the reference's own llicti_A.json -- to "auto" (set_container): there the north star's
        # "torchac replaced by a HIP rANS coder" is the default and the reference's byte format the opt-in.
        self.container_defaulted = "container" not in config
        self.set_container(str(config["container"]) if "container" in config else "ac")
        self._stage = {}                # pinned host staging buffers of the batched path, by (tag, slot): [buffer, event behind its last copy]
        self._xfer = {}                 # (upload, download) copy streams of the batched path, by device index

    def set_container(self, name):
        """Container of the following compress() / encode_batch_async() calls: "ac", "rans<M>", "wrans<M>", "xrans<M>" or "auto"."""
        self.container = str(name)
        self.mode = None if self.container == "auto" else mode_of_name(self.container)

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

    def mode_for_batch(self, B, device=None, sizes=None):
        """Container mode(s) of a call with B images of `sizes` [(H, W), ...] (one entry: all alike): the configured one, or for "auto" each image's
        own -- a function of its size (llicti_amd.codec.auto_modes), not of the batch, the device or anything coded before.  One int where all images
        get the same mode, else one per image (llicti_encode_images_vm)."""
        if self.mode is not None:
            return self.mode
        if not sizes:
            raise ValueError('container "auto" needs the image sizes')
        sz = list(sizes) if len(sizes) == B else [sizes[0]] * B
        modes = auto_modes(sz)
        return modes[0] if all(m == modes[0] for m in modes) else modes

    def _pinned(self, key, nbytes):
        """Pinned host staging buffer (flat uint8, at least nbytes, grown to the running maximum) of `key` = (tag, slot).  The buffer's last
        asynchronous copy -- _pinned_mark() records an event behind it -- has finished when this returns: the host may rewrite it."""
        ent = self._stage.get(key)
        if ent is not None and ent[1] is not None:
            ent[1].synchronize()
            ent[1] = None
        if ent is None or ent[0].numel() < nbytes:
            ent = self._stage[key] = [torch.empty((int(nbytes * (1.25 if ent is not None else 1.0)),), dtype=torch.uint8).pin_memory(), None]
        return ent[0]

    def _pinned_mark(self, key, stream):
        """An asynchronous copy from / into the buffer of `key` has just been enqueued on `stream`."""
        ev = torch.cuda.Event()
        ev.record(stream)
        self._stage[key][1] = ev

    @torch.no_grad()
    def compress_batch(self, x):
        """x: [B,3,H,W] float32 in {k/255} or uint8 -> (list of B bytestream_lists, x_ycocg).  Synchronous."""
        enc = self.encode_batch_async(x, want_ycocg=True)
        return enc.lists(), enc.x_ycocg

    @torch.no_grad()
    def encode_batch_async(self, x, want_ycocg=False, slot=0):
        """Enqueue the encode of a batch and the download of its containers (pinned host buffers); returns an EncodedBatch whose lists() waits
        for the download and cuts the containers into the reference's bytestream_lists.  Nothing here blocks the host: a caller can enqueue
        the next batch before it converts this one (LLICTIAgent.eval_model with eval_batch > 1).  `slot` selects one of the staging buffer
        sets (two batches in flight need two).
        x: a tensor [B,3,H,W] (float32 in {k/255} or uint8; host or device), or a LIST of B uint8 host arrays [3,H_b,W_b] whose sizes may
        differ (the reference's test loader yields arbitrary sizes, dataloaders/image_dl.py:40-45): one call, one upload, one download."""
        if isinstance(x, (list, tuple)):
            imgs = [np.ascontiguousarray(a) for a in x]
            for a in imgs:
                if a.dtype != np.uint8 or a.ndim != 3 or a.shape[0] != 3:
                    raise ValueError("a list batch holds uint8 [3, H, W] arrays")
            codec = self.codec(None)
            Hs, Ws = [int(a.shape[1]) for a in imgs], [int(a.shape[2]) for a in imgs]
            offs, total = codec.flat_offsets(Hs, Ws)
            host = self._pinned(("rgb_in", slot), total)
            hv = host.numpy()
            for a, o in zip(imgs, offs):
                hv[int(o):int(o) + a.size] = a.reshape(-1)
            cur = torch.cuda.current_stream(codec.device)
            up, down = self._copy_streams(codec.device)
            with torch.cuda.stream(up):
                rgb = host[:total].to(codec.device, non_blocking=True)
            self._pinned_mark(("rgb_in", slot), up)
            cur.wait_stream(up)
            rgb.record_stream(cur)
            t0 = torch.cuda.Event(enable_timing=True)              # the compute stream has the batch: what follows is encode time, not upload wait
            t0.record(cur)
            B = len(imgs)
            mode = self.mode_for_batch(B, codec.device, sizes=list(zip(Hs, Ws)))
            cont, seg = codec.encode_v(rgb, Hs, Ws, mode)
            x_ycocg = None
        else:
            t0 = None
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
            Hs, Ws = [H] * B, [W] * B
            mode = self.mode_for_batch(B, codec.device, sizes=[(H, W)])
            cont, seg = codec.encode(rgb, mode=mode)
            # x_ycocg = (YCoCg - [127,0,0]) / 255 (LLICTI_nets.py:143-144): the float planes the reference hands back beside the streams -- the encode
            # has just computed them: copied out of its workspace, not lifted a second time
            x_ycocg = codec.encoded_fplanes(B, H, W, mode) if want_ycocg else None
        nb = cont.numel()
        seg_h = self._pinned(("seg", slot), B * NSEG * 4)[:B * NSEG * 4].view(torch.int32).view(B, NSEG)
        cont_h = self._pinned(("cont_out", slot), nb)[:nb].view(tuple(cont.shape))
        # the download runs on its own stream behind the encode: the compute stream goes straight on to the next enqueued call
        down.wait_stream(cur)
        with torch.cuda.stream(down):
            seg_h.copy_(seg, non_blocking=True)
            cont_h.copy_(cont, non_blocking=True)                  # one contiguous copy of the container strides (bytes past a container's own length: undefined)
            ev = torch.cuda.Event()
            ev.record(down)
        self._pinned_mark(("seg", slot), down)
        self._pinned_mark(("cont_out", slot), down)
        cont.record_stream(down)
        seg.record_stream(down)
        enc = EncodedBatch(codec, rgb, cont_h, seg_h, ev, x_ycocg, mode, Hs, Ws)
        enc.t0 = t0
        return enc

    def _copy_streams(self, device):
        """(upload, download) HIP streams of a device for the batched calls' transfers (created once): PCIe copies next to the kernels, not
        between them -- on one stream they cost a 24-image encode + decode 2.6 ms of a compute queue that is otherwise never idle."""
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        if key not in self._xfer:
            self._xfer[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        return self._xfer[key]

    @torch.no_grad()
    def decompres(self, bytestream_list, devc=None, xorg=None):
        """bytestream_list -> float32 [1,3,H,W] (LLICTI_nets.py:161-179).  xorg: the reference's diagnostic (:167-171) -- the float planes
        compress() returned; the decoded planes are compared with them on the device and a difference of a grey level or more is reported."""
        rgb = self.decode_batch_async([bytestream_list], devc)
        codec = self.codec()
        codec.check()
        if xorg is not None:
            planes = codec.lift(rgb)[1]                            # (YCoCg - [127,0,0]) / 255 of the decoded image, as compress() returns it
            maxx_abserr = float((xorg.to(planes.device) - planes).abs().max()) * 255
            if maxx_abserr >= 1.0:                                 # LLICTI_nets.py:169-171
                print("Error: Decoded YCoCg img does NOT match original YCoCg image perfectly! The maximum of absolute error is {:.4f}".format(maxx_abserr))
        return rgb.to(torch.float32) / 255           # LLICTI_nets.py:87

    @torch.no_grad()
    def decompres_batch(self, lists, devc=None):
        """bytestream_lists of B images -> float32 [B,3,H,W] (equal sizes), or -- images of different sizes, rANS containers -- a LIST of B tensors
        [1,3,H_b,W_b], each what decompres() returns for that image."""
        res = self.decode_batch_async(lists, devc)
        self.codec().check()
        if isinstance(res, tuple):
            flat, Hs, Ws = res
            offs, _ = self.codec().flat_offsets(Hs, Ws)
            return [flat[int(o):int(o) + 3 * h * w].view(1, 3, h, w).to(torch.float32) / 255 for o, h, w in zip(offs, Hs, Ws)]
        return res.to(torch.float32) / 255           # LLICTI_nets.py:87

    @torch.no_grad()
    def decode_batch_async(self, lists, devc=None, slot=0, flat=False):
        """bytestream_lists of B images -> uint8 [B,3,H,W] on the device, enqueued (upload from a pinned buffer + decode); device-side
        failures are reported by codec().check() / image_status().  The images of a call share a container kind; in a rANS container their
        SIZES may differ -- then (or with flat=True) the result is (flat uint8 device tensor, Hs, Ws): the images back to back, [3][H][W] each."""
        codec = self.codec(devc if (devc is not None and torch.device(devc).type == "cuda") else None)
        Hs, Ws, modes = [], [], []
        for bl in lists:
            if len(bl) != 6 or any(len(r) != 9 for r in bl):
                raise ValueError("bytestream_list must be 6 lists of 9 byte strings")
            if len(bl[0][0]) != 3 or len(bl[0][1]) != 12 or len(bl[0][2]) != 2:
                raise ValueError("malformed header streams")
            hdr = bytes(bl[0][0]) + bytes(bl[0][1]) + bytes(bl[0][2])
            modes.append(mode_of_header(hdr))           # AC container: hdr[0] == num_scales (LLICTI_nets.py:424); rANS: its lane kind and stream count
            H, W = header_dims(hdr)
            Hs.append(H)
            Ws.append(W)
        if any((m & ~0xFF) != (modes[0] & ~0xFF) for m in modes):
            raise ValueError("all images of one decompres_batch call must be in the same kind of container (their stream counts may differ)")
        mode = modes[0] if all(m == modes[0] for m in modes) else modes
        B = len(lists)
        mixed = any(h != Hs[0] or w != Ws[0] for h, w in zip(Hs, Ws)) or isinstance(mode, list)
        if mixed and modes[0] == MODE_AC:
            raise ValueError("images of different sizes in the reference-format container decode one size per call")
        stride = max(codec.max_container_bytes(h, w) for h, w in set(zip(Hs, Ws)))
        cont_h = self._pinned(("cont_in", slot), B * stride)[:B * stride].view(B, stride)
        seg_h = self._pinned(("seg_in", slot), B * NSEG * 4)[:B * NSEG * 4].view(torch.int32).view(B, NSEG)
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
        self._pinned_mark(("cont_in", slot), up)
        self._pinned_mark(("seg_in", slot), up)
        cur.wait_stream(up)
        cont_d.record_stream(cur)
        seg_d.record_stream(cur)
        if mixed or flat:
            return codec.decode_v(cont_d, seg_d, Hs, Ws, mode), Hs, Ws
        return codec.decode(cont_d, seg_d, Hs[0], Ws[0], mode=mode)


class EncodedBatch:
    """Handle of one enqueued batch encode (LLICTI.encode_batch_async): the device input, the pinned host copies of the containers and
    their segment lengths, and the event recorded behind the download."""

    def __init__(self, codec, rgb, cont_h, seg_h, ev, x_ycocg, mode, Hs=None, Ws=None):
        self.codec, self.rgb, self.cont_h, self.seg_h, self.ev = codec, rgb, cont_h, seg_h, ev
        self.x_ycocg, self.mode = x_ycocg, mode
        self.Hs, self.Ws = Hs, Ws           # per image (a list batch: rgb is the flat device buffer, the images back to back)
        self.t0 = None                      # list batches: timing event on the compute stream behind the wait for the upload
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
