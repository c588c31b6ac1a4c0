"""Rate bookkeeping of the eval driver: mean of per-image rate arrays and the scale x band x channel table.

Mirrors the interface of the reference's `loggers/rate.py` (`RateLogger.__call__`, `.display(lr, typ)` ->
`(sum, 0.0)`, `state_dict` / `load_state_dict`; reference loggers/rate.py:7-75, table layout :120-168) so logs
of this build and of the reference can be diffed line for line (tests/golden/rate_table.json holds the
reference's own text for a fixed input).  A row is one list of the `bytestream_list`: 9 entries = 3 bands x
(Y, Co, Cg) -- which is why the header list is padded to 9 streams (LLICTI_nets.py:351-354)."""
from __future__ import annotations

import logging
from datetime import datetime

import numpy as np

_HEAD = {"tr": "  Train Epoch: {:3d}  Rates: scl", "te": "   Test Epoch: {:3d}  Rates: hdr ",
         "va": "  Valid Epoch: {:3d}  Rates: scl", "it": "Train Itera: {:3d}  Rates: scl"}
_CONT = {"tr": " " * 35 + "scl", "te": " " * 35 + "scl", "va": " " * 35 + "scl", "it": " " * 33 + "scl"}


def format_rate_table(epoch, rate, lr=0.0, typ="te", clock="") -> str:
    """One log record: a line per row of `rate` ([rows][9]); in 'te' mode row 0 is the header list."""
    rate = np.asarray(rate, dtype=np.float64)
    if rate.ndim != 2 or rate.shape[1] != 9:
        raise AssertionError("rate rows must hold 3 sub-bands x 3 colour channels")     # loggers/rate.py:133
    lines, total = [], 0.0
    for s, row in enumerate(rate):
        if typ == "te":
            tag, name = ("-> ", "hd") if s == 0 else (f"{s - 1:d}-> ", f"s{s - 1:d}")
        else:
            tag, name = f"{s:d}-> ", f"s{s:d}"
        bands = [row[3 * b:3 * b + 3] for b in range(3)]
        body = "".join("{:.2f}+{:.2f}+{:.2f}(b{:d}={:.3f}) ".format(v[0], v[1], v[2], b, v[0] + v[1] + v[2])
                       for b, v in enumerate(bands))
        row_sum = sum(float(v[0] + v[1] + v[2]) for v in bands)
        total += row_sum
        lines.append(tag + body + "({}={:.3f}) ".format(name, row_sum))
    text = _HEAD[typ].format(epoch) + ("\n" + _CONT[typ]).join(lines) + "(({:.3f})) ".format(total)
    text += "  (lr: {:.6f}) ({})".format(lr, clock) if typ in ("tr", "it") else " ({})".format(clock)
    return text


class RateMeter:
    def __init__(self):
        self.rate = []
        self.current_iteration = 0
        self.current_epoch = 0

    def append(self, rate):
        self.current_iteration += 1
        self.rate.append(rate)

    def reset(self):
        self.rate = []

    def mean(self):
        self.current_epoch += 1
        m = np.array(self.rate).mean(axis=0)
        self.reset()
        return m

    def state_dict(self):
        return {"rate": self.rate, "it": self.current_iteration, "ep": self.current_epoch}

    def load_state_dict(self, info):
        self.rate, self.current_iteration, self.current_epoch = info["rate"], info["it"], info["ep"]


class RateLogger(RateMeter):
    def __init__(self):
        super().__init__()
        self.logger = logging.getLogger("Rate Loss")

    def __call__(self, *args):
        self.append(*args)

    def _get_time_now_str(self):
        return datetime.now().strftime("%H:%M:%S")

    def display(self, lr=0.0, typ="tr"):
        rate = self.mean()
        self.logger.info(format_rate_table(self.current_epoch, rate, lr, typ, self._get_time_now_str()))
        return np.sum(rate), 0.0
