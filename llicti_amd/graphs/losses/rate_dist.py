"""Rate bookkeeping of the encode/decode path (reference: graphs/losses/rate_dist.py:79-103, :125-135).
Only the two classes `LLICTIAgent` uses on the hot path; the distortion / gradient losses are out of scope."""
from __future__ import annotations

import torch


class TrainRLossList:
    """graphs/losses/rate_dist.py:79-103: per scale the 9 (band, colour) sums of self-information / numel * 3."""

    def forward(self, numel_x, sinfoslist):
        self.rate1, self.rate1list = 0.0, []
        for t in sinfoslist:
            r = torch.sum(t, dim=(0, 2, 3)) / numel_x * 3
            self.rate1list.append([float(v) for v in r])
            self.rate1 = self.rate1 + float(torch.sum(r))
        return self.rate1, self.rate1list


class CompressionRLossList:
    """graphs/losses/rate_dist.py:125-135: bpp of every stream, len*8/numel*3."""

    def forward(self, numel_x, bytestream_list):
        self.rate1list = [[len(s) * 8 / numel_x * 3 for s in row] for row in bytestream_list]
        return self.rate1list
