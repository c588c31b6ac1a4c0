"""Second, independent reading of the torchac 0.9.3 coder -- TEST INFRASTRUCTURE, pure Python, bit at a time.

The reference calls `torchac.encode_int16_normalized_cdf` / `decode_int16_normalized_cdf`
(graphs/models/LLICTI_nets.py:406-407, :492-493; README.md:12 pins torchac==0.9.3).  torchac is a third-party
package that is neither under /root/reference nor installable in this image, so byte identity with it cannot be
shown ("parity unpinned").  What can be done is to read its published algorithm twice,
independently: `oracle/llicti_oracle.c` (C, word-level closed forms shared with nothing here) and this file, written
from SURVEY.md Appendix A alone as the most literal statement possible -- one Python int per C variable, one loop
iteration per renormalisation step, one list element per output bit -- and to require the two (and the HIP kernels)
to agree byte for byte on the reference's own recorded tables, on random tables and on edge rows, plus on three
vectors worked out by hand (tests/test_ref_ac.py).  Two readings agreeing is still not torchac itself.

Format handled (Appendix A): cdf rows of Lp uint16 words (the reference's int16 tensors reinterpreted), strictly
increasing over entries 0..Lp-2, entry Lp-1 wrapped to 0 and never read; max_symbol = Lp - 2; the top symbol's upper
bound is the constant 0x10000.
"""

M32 = 0xFFFFFFFF
HALF = 0x80000000
QUARTER = 0x40000000
THREEQ = 0xC0000000
PRECISION = 16


class _BitSink:
    """torchac's OutCacheString: bits enter a byte MSB first; flush() pads the open byte with zeros."""

    def __init__(self):
        self.bits = []

    def append(self, bit):
        self.bits.append(1 if bit else 0)

    def append_bit_and_pending(self, bit, pending):
        self.append(bit)
        for _ in range(pending):
            self.append(not bit)

    def to_bytes(self):
        bits = list(self.bits)
        while len(bits) % 8:
            bits.append(0)
        out = bytearray()
        for k in range(0, len(bits), 8):
            v = 0
            for b in bits[k:k + 8]:
                v = (v << 1) | b
            out.append(v)
        return bytes(out)


def _bounds(row, s, max_symbol):
    c_low = int(row[s]) & 0xFFFF
    c_high = 0x10000 if s == max_symbol else int(row[s + 1]) & 0xFFFF
    return c_low, c_high


def encode(cdf_rows, symbols):
    """cdf_rows: sequence of N rows of Lp ints (uint16 words); symbols: N ints in [0, Lp-2] -> bytes."""
    n = len(symbols)
    assert len(cdf_rows) == n
    low, high, pending = 0, M32, 0
    out = _BitSink()
    for i in range(n):
        row = cdf_rows[i]
        max_symbol = len(row) - 2
        s = int(symbols[i])
        assert 0 <= s <= max_symbol
        span = high - low + 1                                   # uint64 in torchac
        c_low, c_high = _bounds(row, s, max_symbol)
        high = ((low - 1) + ((span * c_high) >> PRECISION)) & M32
        low = (low + ((span * c_low) >> PRECISION)) & M32
        while True:
            if high < HALF:
                out.append_bit_and_pending(0, pending)
                pending = 0
                low = (low << 1) & M32
                high = ((high << 1) & M32) | 1
            elif low >= HALF:
                out.append_bit_and_pending(1, pending)
                pending = 0
                low = (low << 1) & M32
                high = ((high << 1) & M32) | 1
            elif low >= QUARTER and high < THREEQ:
                pending += 1
                low = (low << 1) & 0x7FFFFFFF
                high = ((high << 1) & M32) | 0x80000001
            else:
                break
    pending += 1
    out.append_bit_and_pending(0 if low < QUARTER else 1, pending)
    return out.to_bytes()


class _BitSource:
    """torchac's InCacheString: bits MSB first; past the end every read shifts in a 0."""

    def __init__(self, data):
        self.data = bytes(data)
        self.pos = 0

    def get(self, value):
        byte, bit = divmod(self.pos, 8)
        self.pos += 1
        b = (self.data[byte] >> (7 - bit)) & 1 if byte < len(self.data) else 0
        return ((value << 1) & M32) | b


def _binsearch(row, target, max_symbol):
    left, right = 0, max_symbol + 1
    while left + 1 < right:
        m = (left + right) // 2
        v = int(row[m]) & 0xFFFF
        if v < target:
            left = m
        elif v > target:
            right = m
        else:
            return m
    return left


def decode(cdf_rows, data):
    """-> list of N symbols (N = len(cdf_rows))."""
    n = len(cdf_rows)
    src = _BitSource(data)
    low, high, value = 0, M32, 0
    for _ in range(32):
        value = src.get(value)
    out = []
    for i in range(n):
        row = cdf_rows[i]
        max_symbol = len(row) - 2
        span = high - low + 1
        count = ((((value - low + 1) & 0xFFFFFFFFFFFFFFFF) * 0x10000 - 1) // span) & 0xFFFF      # cast to uint16
        s = _binsearch(row, count, max_symbol)
        out.append(s)
        if i == n - 1:
            break
        c_low, c_high = _bounds(row, s, max_symbol)
        high = ((low - 1) + ((span * c_high) >> PRECISION)) & M32
        low = (low + ((span * c_low) >> PRECISION)) & M32
        while True:
            if low >= HALF or high < HALF:
                low = (low << 1) & M32
                high = ((high << 1) & M32) | 1
                value = src.get(value)
            elif low >= QUARTER and high < THREEQ:
                low = (low << 1) & 0x7FFFFFFF
                high = ((high << 1) & M32) | 0x80000001
                value = (value - QUARTER) & M32
                value = src.get(value)
            else:
                break
    return out
