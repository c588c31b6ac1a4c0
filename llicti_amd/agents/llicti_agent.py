"""`LLICTIAgent.eval_model` on the MI355X hot path (reference: agents/llicti_agent.py:122-164,
agents/base.py:14-28, :102-123; graphs/losses/rate_dist.py:125-135).

Per test image: time `compress`, time `decompres`, bits per sub-pixel over ALL streams including the
header ones, and the lossless self-check `max|x - x_reco| * 255 < 0.5`, logged in the reference's line
format.  Training / validation modes are outside the hot path and raise.
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np
import torch

from ..graphs.losses.rate_dist import CompressionRLossList, TrainRLossList
from ..graphs.models.LLICTI_nets import LLICTI
from ..loggers.rate import RateLogger
from ..weights import load_reference_state_dict


def _iter_test_images_u8(config):
    """The test images as uint8 [3,H,W] host arrays (the batched eval path uploads uint8: a quarter of the float32 bytes)."""
    src = config.test_data
    if not isinstance(src, str):                       # an in-memory data set: any iterable of uint8 [3,H,W] arrays
        for rgb in src:
            rgb = np.asarray(rgb)
            if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[0] != 3:
                raise ValueError("in-memory test_data must yield uint8 [3, H, W] arrays")
            yield rgb
        return
    if src.startswith("synthetic:"):
        H, W, N = (int(v) for v in src.split(":")[1].split("x"))
        for i in range(N):
            yield np.random.default_rng(i).integers(0, 256, size=(3, H, W), dtype=np.uint8)
        return
    from ..fileio import read_image
    for f in sorted(f for f in os.listdir(src) if f.lower().endswith((".png", ".jpg", ".ppm"))):
        yield read_image(os.path.join(src, f))


def _iter_test_images(config, device):
    """Test loader, batch 1, no crop (dataloaders/image_dl.py:40-45, :106-111): float32 1x3xHxW = uint8/255.
    `test_data` is a directory of .png/.jpg/.ppm, "synthetic:HxWxN" (seeded uniform RGB) or an in-memory
    iterable of uint8 [3,H,W] arrays."""
    for rgb in _iter_test_images_u8(config):
        yield torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to(device)


class LLICTIAgent:
    def __init__(self, config):
        self.config = config
        self.logger = logging.getLogger("Agent")
        self.cuda = torch.cuda.is_available() and bool(config.cuda)
        if not self.cuda:
            raise RuntimeError("LLICTIAgent (MI355X hot path) needs a GPU: there is no CPU fallback")
        torch.cuda.set_device(config.gpu_device)
        self.device = torch.device("cuda", config.gpu_device)
        torch.manual_seed(config.seed)              # base.py:21-28 (one seed for the default init)
        assert config.wtr_type in ("lazydwt", "x")
        self.model = LLICTI(config).to(self.device)
        self.compr_loss = CompressionRLossList()
        self.train_loss = TrainRLossList()             # llicti_agent.py:22 (used by validate, :96)
        self.valid_logger = RateLogger()               # llicti_agent.py:39
        self.test_logger = RateLogger()                # llicti_agent.py:40
        self.results = []
        if config.mode in ("test", "validate", "debug", "eval_model"):
            self.load_checkpoint("model_best.pth.tar")
        self.model_size_estimation()                   # llicti_agent.py:46

    def load_checkpoint(self, filename):
        """base.py:51-81: a missing checkpoint is tolerated (the run continues with the seeded init)."""
        path = os.path.join(getattr(self.config, "checkpoint_dir", "") if "checkpoint_dir" in self.config else "", filename)
        try:
            ckpt = torch.load(path, map_location=self.device)
            # strict, as agents/base.py:60: only compressai's known extra buffers (scale_table, _offset, _quantized_cdf, ...)
            # are dropped; any other mismatch raises instead of silently keeping the seeded init
            load_reference_state_dict(self.model, ckpt["state_dict"])
            self.logger.info("Checkpoint loaded from '%s'", path)
        except OSError:
            self.logger.info("No checkpoint at '%s' -- running with the seeded default init", path)

    def model_size_estimation(self):
        """llicti_agent.py:167-192: bytes of parameters and of buffers, logged in MiB.  The reference's own run logs
        "0.750+0.000=0.750MB" (experiments/.../exp_debug.log:101): 196,596 fp32 parameters, nine 1-element buffers."""
        mib = float(1024 ** 2)
        n_par = sum(p.nelement() * p.element_size() for p in self.model.parameters())
        n_buf = sum(b.nelement() * b.element_size() for b in self.model.buffers())
        self.size_text = " model param+buffer=total size: {:.3f}+{:.3f}={:.3f}MB".format(n_par / mib, n_buf / mib, (n_par + n_buf) / mib)
        self.logger.info("------------------TOT----------------------------------------------")
        self.logger.info(self.size_text)
        self.logger.info("------------------END----------------------------------------------")
        return n_par, n_buf

    def run(self):
        if self.config.mode == "eval_model":
            return self.eval_model()
        if self.config.mode == "validate":
            return self.validate()
        raise NameError("'" + str(self.config.mode) + "' is not available on the MI355X hot path (eval_model, validate)")

    def _pad_img(self, x):
        """llicti_agent.py:105-113: replicate-pad right / bottom to multiples of 2**(max(dwtlevels)+1) = 32."""
        blk = 2 ** (max(self.config.dwtlevels) + 1)
        h, w = x.size(2), x.size(3)
        nh, nw = (h + blk - 1) // blk * blk, (w + blk - 1) // blk * blk
        return torch.nn.functional.pad(x, (0, nw - w, 0, nh - h), mode="replicate")

    @torch.no_grad()
    def validate(self):
        """llicti_agent.py:85-103 without the LR scheduler: estimated rate (self-information of LLICTI.forward) of the
        validation images -- here the same image source as eval_model -- logged as the scale x band x colour table."""
        self.model.eval()
        n = 0
        for x in _iter_test_images(self.config, self.device):
            x = self._pad_img(x)
            infos = self.model(x)
            _, rate1_list = self.train_loss.forward(torch.numel(x), infos)
            self.valid_logger(rate1_list)
            n += 1
        if n == 0:
            return 0.0
        rate, rate2 = self.valid_logger.display(lr=0.0, typ="va")
        return float(rate + rate2)

    def _log_image(self, idx, H, W, bpsp, enc_time, dec_time, maxx_abserr):
        """The reference's per-image line (llicti_agent.py:154-162), unchanged."""
        print_text = "{:3d} {:3d}x{:3d} ".format(idx, H, W)
        if maxx_abserr >= 0.5:
            self.logger.info(print_text + "bpsp= {:.3f} Enc/Dec-Times:{:.3f}/{:.3f} "
                             "(Error: Decoded img does NOT match original image perfectly! "
                             "The maximum of absolute error is {:.4f})".format(bpsp, enc_time, dec_time, maxx_abserr))
        else:
            self.logger.info(print_text + "bpsp= {:.3f} Enc/Dec-Times:{:.3f}/{:.3f} "
                             "(Check: Decoded img matches original)".format(bpsp, enc_time, dec_time))

    @torch.no_grad()
    def eval_model_batched(self, eval_batch):
        """eval_model for throughput: the test images are coded `eval_batch` at a time, IN THE ORDER THE LOADER YIELDS THEM AND WHATEVER THEIR
        SIZES (the reference's loader yields batch-1 images of arbitrary size, dataloaders/image_dl.py:40-45; its own 500-image test set has
        119 sizes, interleaved) through LLICTI.encode_batch_async / decode_batch_async on a list of images -- the same bytestream_lists,
        rates, lossless check and per-image log lines as the one-image loop (llicti_agent.py:122-164), in the same order.  (In the
        reference-format container, which codes one size per call, a batch closes where the size changes.)  The loop is software-pipelined
        one batch deep: while the GPU encodes batch k + 1 the host cuts batch k's containers into bytestream_lists, books their rates and
        packs them again for the decoder, so neither side waits for the other; uploads and downloads run on the model's copy streams, next
        to the kernels.  Enc/Dec-Times of an image are its batch's GPU time (HIP events around the enqueued calls on the compute stream,
        the encode's behind the wait for its upload) times the image's share of the batch's pixels.  With config.keep_streams the lists
        stay in self.results."""
        self.model.eval()
        self.results = []
        keep = bool(self.config["keep_streams"]) if "keep_streams" in self.config else False
        stream = torch.cuda.current_stream(self.device)
        one_size = self.model.mode is not None and self.model.mode == 0          # reference format: equal sizes per call

        def batches():
            cur = []
            for rgb in _iter_test_images_u8(self.config):
                if cur and (len(cur) == eval_batch or (one_size and rgb.shape != cur[0].shape)):
                    yield cur
                    cur = []
                cur.append(rgb)
            if cur:
                yield cur

        def start_encode(imgs, slot):
            B = len(imgs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            up, _ = self.model._copy_streams(self.device)
            enc = self.model.encode_batch_async(imgs, slot=slot)        # list of uint8 host arrays: staged in pinned memory, uploaded on the model's copy stream
            e1.record(stream)
            return {"enc": enc, "e_enc": (enc.t0 if enc.t0 is not None else e0, e1), "B": B, "Hs": enc.Hs, "Ws": enc.Ws}

        def finish(job, idx0):
            """host half of a batch: lists, rates, decode enqueue; then (synchronising) the lossless check and the log lines"""
            enc = job["enc"]
            lists = enc.lists(check=False)
            rates = [self.compr_loss.forward(3 * h * w, bl) for bl, h, w in zip(lists, job["Hs"], job["Ws"])]
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record(stream)
            rec, _, _ = self.model.decode_batch_async(lists, self.device, slot=job["slot"], flat=True)
            n = rec.numel()
            diff = torch.maximum(rec, enc.rgb[:n]) - torch.minimum(rec, enc.rgb[:n])    # |x - x_reco| * 255 in uint8 arithmetic, the images back to back
            offs, _ = enc.codec.flat_offsets(job["Hs"], job["Ws"])
            if len(set(zip(job["Hs"], job["Ws"]))) == 1:
                err = diff.view(job["B"], -1).amax(dim=1).to(torch.int16)
            else:
                err = torch.stack([diff[int(o):int(o) + 3 * h * w].amax() for o, h, w in zip(offs, job["Hs"], job["Ws"])]).to(torch.int16)
            d1.record(stream)
            err_h = self.model._pinned(("err", job["slot"]), 2 * job["B"])[:2 * job["B"]].view(torch.int16)
            err_h.copy_(err, non_blocking=True)                # read in report() behind its own event: no wait for later batches
            self.model._pinned_mark(("err", job["slot"]), stream)
            ev = torch.cuda.Event()
            ev.record(stream)
            return {"job": job, "lists": lists, "rates": rates, "err": err_h, "ev": ev, "e_dec": (d0, d1), "idx0": idx0}

        def report(fin):
            job = fin["job"]
            fin["ev"].synchronize()                            # this batch's decode and check are done; what was enqueued behind them keeps running
            err = fin["err"].numpy().astype(np.float64)
            enc_ms = job["e_enc"][0].elapsed_time(job["e_enc"][1])
            dec_ms = fin["e_dec"][0].elapsed_time(fin["e_dec"][1])
            pix = float(sum(h * w for h, w in zip(job["Hs"], job["Ws"])))
            for b in range(job["B"]):
                bl, rate1_list = fin["lists"][b], fin["rates"][b]
                H, W = job["Hs"][b], job["Ws"][b]
                self.test_logger(rate1_list)                   # llicti_agent.py:140
                bpsp = sum(len(s) * 8 for row in bl for s in row) / (3 * H * W)
                share = H * W / pix
                enc_t, dec_t = enc_ms / 1e3 * share, dec_ms / 1e3 * share
                self._log_image(fin["idx0"] + b, H, W, bpsp, enc_t, dec_t, float(err[b]))
                r = {"idx": fin["idx0"] + b, "H": H, "W": W, "bpsp": bpsp, "enc_s": enc_t, "dec_s": dec_t,
                     "max_abs_err": float(err[b]), "rates": rate1_list, "batch": job["B"]}
                if keep:
                    r["bytestream_list"] = bl
                self.results.append(r)

        idx, k, pending, prev_fin = 0, 0, None, None
        for imgs in batches():
            job = start_encode(imgs, k & 1)                    # GPU: encode batch k ...
            job["slot"] = k & 1
            if pending is not None:                            # ... host: lists / rates / repack of batch k - 1, decode k - 1 enqueued behind encode k
                if prev_fin is not None:
                    report(prev_fin)
                prev_fin = finish(pending, pending["idx0"])
            job["idx0"] = idx
            idx += job["B"]
            pending = job
            k += 1
        if pending is not None:
            if prev_fin is not None:
                report(prev_fin)
            prev_fin = finish(pending, pending["idx0"])
        if prev_fin is not None:
            report(prev_fin)
        # device-side failures (malformed container, overflow) are latched in the context: one check at the end -- a check per batch
        # would synchronise the whole stream and undo the pipelining; a failed image shows up in its own log line (max error) anyway
        self.model.codec().check()
        if self.results:
            self.test_logger.display(lr=0.0, typ="te")
        return self.results

    @torch.no_grad()
    def eval_model(self):
        eval_batch = int(self.config["eval_batch"]) if "eval_batch" in self.config else 1
        if eval_batch > 1:
            return self.eval_model_batched(eval_batch)
        self.model.eval()
        self.results = []
        for batch_idx, x in enumerate(_iter_test_images(self.config, self.device)):
            torch.cuda.synchronize()
            t0 = time.time()
            bytestream_list, xorg = self.model.compress(x)
            torch.cuda.synchronize()
            enc_time = time.time() - t0
            rate1_list = self.compr_loss.forward(torch.numel(x), bytestream_list)
            self.test_logger(rate1_list)               # llicti_agent.py:140
            total = sum(len(s) * 8 for row in bytestream_list for s in row)
            t0 = time.time()
            x_reco = self.model.decompres(bytestream_list, self.device)
            torch.cuda.synchronize()
            dec_time = time.time() - t0
            maxx_abserr = float(((x - x_reco) * 255).abs().max())
            bpsp = total / torch.numel(x)
            self._log_image(batch_idx, x.shape[2], x.shape[3], bpsp, enc_time, dec_time, maxx_abserr)
            self.results.append({"idx": batch_idx, "H": int(x.shape[2]), "W": int(x.shape[3]), "bpsp": bpsp,
                                 "enc_s": enc_time, "dec_s": dec_time, "max_abs_err": maxx_abserr, "rates": rate1_list})
            if "keep_streams" in self.config and self.config["keep_streams"]:
                self.results[-1]["bytestream_list"] = bytestream_list
        if self.results:
            self.test_logger.display(lr=0.0, typ="te")     # mean scale x band x channel table (llicti_agent.py:164)
        return self.results

    def finalize(self):
        self.logger.info("Please wait while finalizing the operation.. Thank you")
