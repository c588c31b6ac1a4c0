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
from .. import shard
from ..loggers.rate import RateLogger
from ..weights import load_reference_state_dict


def _iter_test_images_u8(config, rank=0, world=1, with_index=False):
    """The test images as uint8 [3,H,W] host arrays (the batched eval path uploads uint8: a quarter of the float32 bytes).  With several
    ranks: image i of the data set belongs to rank i mod world -- a rank reads and yields only its own, in order;
    with_index: (index in the whole data set, image) pairs."""
    src = config.test_data

    def out(i, rgb):
        return (i, rgb) if with_index else rgb
    if not isinstance(src, str):                       # an in-memory data set: any iterable of uint8 [3,H,W] arrays
        for i, rgb in enumerate(src):
            if i % world != rank:
                continue
            rgb = np.asarray(rgb)
            if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[0] != 3:
                raise ValueError("in-memory test_data must yield uint8 [3, H, W] arrays")
            yield out(i, rgb)
        return
    if src.startswith("synthetic:"):
        H, W, N = (int(v) for v in src.split(":")[1].split("x"))
        for i in range(rank, N, world):
            yield out(i, np.random.default_rng(i).integers(0, 256, size=(3, H, W), dtype=np.uint8))
        return
    from ..fileio import read_image
    files = sorted(f for f in os.listdir(src) if f.lower().endswith((".png", ".jpg", ".ppm")))
    for i in range(rank, len(files), world):
        yield out(i, read_image(os.path.join(src, files[i])))


def _iter_test_images(config, device, rank=0, world=1, with_index=False):
    """Test loader, batch 1, no crop (dataloaders/image_dl.py:40-45, :106-111): float32 1x3xHxW = uint8/255.
    `test_data` is a directory of .png/.jpg/.ppm, "synthetic:HxWxN" (seeded uniform RGB) or an in-memory
    iterable of uint8 [3,H,W] arrays."""
    for i, rgb in _iter_test_images_u8(config, rank, world, with_index=True):
        x = torch.from_numpy(rgb.astype(np.float32) / np.float32(255)).unsqueeze(0).to(device)
        yield (i, x) if with_index else x


class LLICTIAgent:
    """One process per GPU.  Started alone it is the reference's agent (agents/base.py:21-25: a single device).  Started by
    `python -m torch.distributed.run --nproc-per-node G ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the G agents shard the
    test set: image i -> rank i mod G, weights read once by rank 0 and broadcast (RCCL), no collective on the hot path,
    ONE all_gather of the per-image records at the end; rank 0 prints the reference's per-image lines in index order and the rate table
    -- the log of a 1-GPU run (llicti_amd/shard.py; tests/test_distributed_cpu.py::test_agent_two_ranks_equal_one_rank): container "auto" gives an
    image a container that depends on its size alone (llicti_amd.codec.image_streams), so its bytes -- and with them every logged rate -- are the same
    whatever the rank count, eval_batch or the coding order.

    Defaults: a config WITHOUT `container` / `eval_batch` keys -- the reference's own configs/llicti_A.json -- runs eval_model in
    container "auto" with eval_batch 24, and says so in one log line; `"container": "ac"` is the opt-in for reference-format bytes (one image at a
    time unless eval_batch says otherwise)."""

    DEFAULT_EVAL_BATCH = 24

    def __init__(self, config, model=None):
        self.config = config
        self.logger = logging.getLogger("Agent")
        self.cuda = torch.cuda.is_available() and bool(config.cuda)
        if model is None:
            if not self.cuda:
                raise RuntimeError("LLICTIAgent (MI355X hot path) needs a GPU: there is no CPU fallback")
            n_dev = max(1, torch.cuda.device_count())
            local = int(os.environ.get("LOCAL_RANK", "-1"))
            gpu = config.gpu_device if local < 0 else local % n_dev      # under the launcher: one rank per GPU
            torch.cuda.set_device(gpu)
            self.device = torch.device("cuda", gpu)
            # several ranks on one host (the launcher): this rank's host threads -- and with them its pinned staging buffers, placed by first touch --
            # go to the NUMA node its GPU hangs off (llicti_amd.shard.bind_to_gpu_numa); a lone process keeps the affinity it was given
            self.numa = shard.bind_to_gpu_numa(self.device) if (local >= 0 and int(os.environ.get("WORLD_SIZE", "1")) > 1) else None
        else:
            self.device = torch.device("cpu")          # an injected model (tests: the multi-rank plumbing without a GPU)
        self.rank, self.world = shard.init_from_env(self.device)
        torch.manual_seed(config.seed)              # base.py:21-28 (one seed for the default init)
        assert config.wtr_type in ("lazydwt", "x")
        self.model = LLICTI(config).to(self.device) if model is None else model
        self.compr_loss = CompressionRLossList()
        self.train_loss = TrainRLossList()             # llicti_agent.py:22 (used by validate, :96)
        self.valid_logger = RateLogger()               # llicti_agent.py:39
        self.test_logger = RateLogger()                # llicti_agent.py:40
        self.results = []
        if config.mode in ("test", "validate", "debug", "eval_model") and self.rank == 0 and model is None:
            self.load_checkpoint("model_best.pth.tar")
        if self.world > 1 and isinstance(self.model, torch.nn.Module):
            shard.broadcast_module_state(self.model, self.device)      # rank 0's weights (a checkpoint is read once), 0.79 MB
        if isinstance(self.model, torch.nn.Module):
            self.model_size_estimation()               # llicti_agent.py:46

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

    def _emit(self, idx, H, W, bl, rate1_list, enc_time, dec_time, maxx_abserr, batch=1, keep=False):
        """One coded image: alone, book its rates and print its line now (llicti_agent.py:140, :154-162); as one rank of several, keep its
        record -- rank 0 prints every rank's lines in index order after the gather (_finish_sharded)."""
        nbytes = [len(s) for row in bl for s in row]
        bpsp = sum(nbytes) * 8 / (3 * H * W)
        r = {"idx": idx, "H": H, "W": W, "bpsp": bpsp, "enc_s": enc_time, "dec_s": dec_time, "max_abs_err": maxx_abserr, "rates": rate1_list, "batch": batch}
        if keep:
            r["bytestream_list"] = bl
        self.results.append(r)
        if self.world == 1:
            self.test_logger(rate1_list)
            self._log_image(idx, H, W, bpsp, enc_time, dec_time, maxx_abserr)
        else:
            self._records.append([float(idx), float(H), float(W), float(enc_time), float(dec_time), float(maxx_abserr)] + [float(n) for n in nbytes])
        return r

    def _finish_sharded(self):
        """Several ranks: ONE all_gather of the per-image records (index, size, times, error, the 54 stream lengths); rank 0 books the rates and
        prints the lines of ALL images in index order, then the table -- byte for byte the log of a one-rank run but for the times."""
        if self.world == 1:
            if self.results:
                self.test_logger.display(lr=0.0, typ="te")     # mean scale x band x channel table (llicti_agent.py:164)
            return self.results
        coll_dev = self.device if (self.device.type == "cuda" and torch.distributed.get_backend() == "nccl") else "cpu"
        rec = shard.gather_records(self._records, 6 + 54, device=coll_dev)
        self.all_results = []
        if self.rank == 0:
            for row in sorted(rec.tolist(), key=lambda r: r[0]):
                idx, H, W = int(row[0]), int(row[1]), int(row[2])
                lens = [int(v) for v in row[6:]]
                rate1_list = [[n * 8 / (3 * H * W) * 3 for n in lens[9 * k:9 * k + 9]] for k in range(6)]    # CompressionRLossList on the lengths
                self.test_logger(rate1_list)
                bpsp = sum(lens) * 8 / (3 * H * W)
                self._log_image(idx, H, W, bpsp, row[3], row[4], row[5])
                self.all_results.append({"idx": idx, "H": H, "W": W, "bpsp": bpsp, "enc_s": row[3], "dec_s": row[4], "max_abs_err": row[5], "rates": rate1_list})
            if self.all_results:
                self.test_logger.display(lr=0.0, typ="te")
        return self.results

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
        self.results, self._records = [], []
        keep = bool(self.config["keep_streams"]) if "keep_streams" in self.config else False
        stream = torch.cuda.current_stream(self.device)
        one_size = self.model.mode is not None and self.model.mode == 0          # reference format: equal sizes per call

        def batches():
            """(indices in the whole data set, images): this rank's images (all of them when it is alone), eval_batch at a time"""
            cur, ids = [], []
            for i, rgb in _iter_test_images_u8(self.config, self.rank, self.world, with_index=True):
                if cur and (len(cur) == eval_batch or (one_size and rgb.shape != cur[0].shape)):
                    yield ids, cur
                    cur, ids = [], []
                cur.append(rgb)
                ids.append(i)
            if cur:
                yield ids, cur

        def start_encode(imgs, slot):
            B = len(imgs)
            e1 = torch.cuda.Event(enable_timing=True)
            enc = self.model.encode_batch_async(imgs, slot=slot)        # list of uint8 host arrays: staged in pinned memory, uploaded on the model's copy stream
            e1.record(stream)
            return {"enc": enc, "e_enc": (enc.t0, e1), "B": B, "Hs": enc.Hs, "Ws": enc.Ws}      # (t0: recorded by the list path behind the wait for its upload)

        def finish(job, idx0):
            """host half of a batch: lists, rates, decode enqueue; then (synchronising) the lossless check and the log lines"""
            enc = job["enc"]
            lists = enc.lists(check=False)
            rates = [self.compr_loss.forward(3 * h * w, bl) for bl, h, w in zip(lists, job["Hs"], job["Ws"])]
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record(stream)
            rec, _, _ = self.model.decode_batch_async(lists, self.device, slot=job["slot"], flat=True)
            n = rec.numel()
            # the lossless check (llicti_agent.py:151-162) on the device, the images back to back: ONE comparison pass says per image whether any sub-pixel
            # differs -- the answer is "no" for every image of every run that is not broken, so the error's SIZE (the reference prints it) is only
            # worked out, in report(), for an image that does differ (round 5: maximum - minimum - amax over every batch was four passes over 28 MB)
            ne = rec != enc.rgb[:n]
            offs, _ = enc.codec.flat_offsets(job["Hs"], job["Ws"])
            if len(set(zip(job["Hs"], job["Ws"]))) == 1:
                err = ne.view(job["B"], -1).any(dim=1).to(torch.int16)
            else:
                err = torch.stack([ne[int(o):int(o) + 3 * h * w].any() for o, h, w in zip(offs, job["Hs"], job["Ws"])]).to(torch.int16)
            d1.record(stream)
            err_h = self.model._pinned(("err", job["slot"]), 2 * job["B"])[:2 * job["B"]].view(torch.int16)
            err_h.copy_(err, non_blocking=True)                # read in report() behind its own event: no wait for later batches
            self.model._pinned_mark(("err", job["slot"]), stream)
            ev = torch.cuda.Event()
            ev.record(stream)
            return {"job": job, "lists": lists, "rates": rates, "err": err_h, "ev": ev, "e_dec": (d0, d1), "rec": rec, "offs": offs}

        def report(fin):
            job = fin["job"]
            fin["ev"].synchronize()                            # this batch's decode and check are done; what was enqueued behind them keeps running
            err = fin["err"].numpy().astype(np.float64)                # 0 / 1 per image: does any sub-pixel differ
            for b in np.nonzero(err)[0]:                               # (never, unless something is broken) how much: max |x - x_reco| * 255
                o, n_b = int(fin["offs"][b]), 3 * job["Hs"][b] * job["Ws"][b]
                a, c = fin["rec"][o:o + n_b].to(torch.int16), job["enc"].rgb[o:o + n_b].to(torch.int16)
                err[b] = float((a - c).abs().max())
            enc_ms = job["e_enc"][0].elapsed_time(job["e_enc"][1])
            dec_ms = fin["e_dec"][0].elapsed_time(fin["e_dec"][1])
            pix = float(sum(h * w for h, w in zip(job["Hs"], job["Ws"])))
            for b in range(job["B"]):
                bl, rate1_list = fin["lists"][b], fin["rates"][b]
                H, W = job["Hs"][b], job["Ws"][b]
                share = H * W / pix
                self._emit(job["ids"][b], H, W, bl, rate1_list, enc_ms / 1e3 * share, dec_ms / 1e3 * share, float(err[b]), batch=job["B"], keep=keep)

        k, pending, prev_fin = 0, None, None
        for ids, imgs in batches():
            job = start_encode(imgs, k & 1)                    # GPU: encode batch k ...
            job["slot"], job["ids"] = k & 1, ids
            if pending is not None:                            # ... host: lists / rates / repack of batch k - 1, decode k - 1 enqueued behind encode k
                if prev_fin is not None:
                    report(prev_fin)
                prev_fin = finish(pending, None)
            pending = job
            k += 1
        if pending is not None:
            if prev_fin is not None:
                report(prev_fin)
            prev_fin = finish(pending, None)
        if prev_fin is not None:
            report(prev_fin)
        # device-side failures (malformed container, overflow) are latched in the context: one check at the end -- a check per batch
        # would synchronise the whole stream and undo the pipelining; a failed image shows up in its own log line (max error) anyway
        self.model.codec().check()
        return self._finish_sharded()

    @torch.no_grad()
    def eval_model(self):
        # the reference's own config has neither key: the throughput path is the default, the reference-format loop the opt-in
        defaulted = getattr(self.model, "container_defaulted", False) and "container" not in self.config and hasattr(self.model, "set_container")
        prev = self.model.container if defaulted else None
        if defaulted:
            self.model.set_container("auto")          # for this run only: compress() on the model keeps returning the reference format unless the config says otherwise
        try:
            if "eval_batch" in self.config:
                eval_batch = int(self.config["eval_batch"])
            else:
                eval_batch = self.DEFAULT_EVAL_BATCH if getattr(self.model, "container", "ac") != "ac" else 1
            if defaulted or "eval_batch" not in self.config:
                self.logger.info('eval_model: container "%s", eval_batch %d (defaults of the MI355X path: rANS streams per image, their number from the image itself, '
                                 'batched and pipelined; "container": "ac" + "eval_batch": 1 in the config give the reference-format one-image loop)',
                                 getattr(self.model, "container", "?"), eval_batch)
            if eval_batch > 1:
                return self.eval_model_batched(eval_batch)
            return self._eval_model_one_by_one()
        finally:
            if defaulted:
                self.model.set_container(prev)

    @torch.no_grad()
    def _eval_model_one_by_one(self):
        """The reference's loop as it is (llicti_agent.py:122-164): one image per compress() / decompres() call."""
        self.model.eval()
        self.results, self._records = [], []
        keep = bool(self.config["keep_streams"]) if "keep_streams" in self.config else False
        sync = torch.cuda.synchronize if self.device.type == "cuda" else (lambda: None)
        for batch_idx, x in _iter_test_images(self.config, self.device, self.rank, self.world, with_index=True):
            sync()
            t0 = time.time()
            bytestream_list, xorg = self.model.compress(x)
            sync()
            enc_time = time.time() - t0
            rate1_list = self.compr_loss.forward(torch.numel(x), bytestream_list)
            t0 = time.time()
            x_reco = self.model.decompres(bytestream_list, self.device)
            sync()
            dec_time = time.time() - t0
            maxx_abserr = float(((x - x_reco) * 255).abs().max())
            self._emit(batch_idx, int(x.shape[2]), int(x.shape[3]), bytestream_list, rate1_list, enc_time, dec_time, maxx_abserr, keep=keep)
        return self._finish_sharded()

    def finalize(self):
        self.logger.info("Please wait while finalizing the operation.. Thank you")
