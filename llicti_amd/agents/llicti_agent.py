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


def _iter_test_images(config, device):
    """Test loader, batch 1, no crop (dataloaders/image_dl.py:40-45, :106-111): float32 1x3xHxW = uint8/255.
    `test_data` is a directory of .png/.jpg or "synthetic:HxWxN" (seeded uniform RGB)."""
    src = config.test_data
    if isinstance(src, str) and src.startswith("synthetic:"):
        H, W, N = (int(v) for v in src.split(":")[1].split("x"))
        for i in range(N):
            rgb = np.random.default_rng(i).integers(0, 256, size=(3, H, W), dtype=np.uint8)
            yield torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to(device)
        return
    from ..fileio import read_image
    names = sorted(f for f in os.listdir(src) if f.lower().endswith((".png", ".jpg", ".ppm")))
    for f in names:
        rgb = read_image(os.path.join(src, f))
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

    @torch.no_grad()
    def eval_model(self):
        self.model.eval()
        self.results = []
        for batch_idx, x in enumerate(_iter_test_images(self.config, self.device)):
            print_text = "{:3d} {:3d}x{:3d} ".format(batch_idx, x.shape[2], x.shape[3])
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
            if maxx_abserr >= 0.5:
                self.logger.info(print_text + "bpsp= {:.3f} Enc/Dec-Times:{:.3f}/{:.3f} "
                                 "(Error: Decoded img does NOT match original image perfectly! "
                                 "The maximum of absolute error is {:.4f})".format(bpsp, enc_time, dec_time, maxx_abserr))
            else:
                self.logger.info(print_text + "bpsp= {:.3f} Enc/Dec-Times:{:.3f}/{:.3f} "
                                 "(Check: Decoded img matches original)".format(bpsp, enc_time, dec_time))
            self.results.append({"idx": batch_idx, "H": int(x.shape[2]), "W": int(x.shape[3]), "bpsp": bpsp,
                                 "enc_s": enc_time, "dec_s": dec_time, "max_abs_err": maxx_abserr, "rates": rate1_list})
        if self.results:
            self.test_logger.display(lr=0.0, typ="te")     # mean scale x band x channel table (llicti_agent.py:164)
        return self.results

    def finalize(self):
        self.logger.info("Please wait while finalizing the operation.. Thank you")
