"""Single-file container and image I/O around the codec.

The reference never writes a compressed file: `compress` returns a Python list of lists of `bytes`
(LLICTI_nets.py:352-354, :411) and the test loader reads PNG/JPG through torchvision
(dataloaders/image_dl.py:106-111).  This module adds the two missing ends so the hot path is usable as a codec:

  * `.llic` file = magic b"LLIC", format version (1 byte), number of segments n (1 byte), n little-endian
    uint32 segment lengths, then the segments back to back -- exactly the device container of
    include/llicti_hip.h (header triplet, raw DC band, then the 45 AC streams or the M rANS streams) with its
    `seg_len` row in front, so a file maps to a `bytestream_list` and back without touching a byte of payload;
  * 8-bit RGB images as uint8 [3, H, W]: binary PPM (P6) natively, PNG/JPG through PIL when it is installed.
"""
from __future__ import annotations

import os
import struct

import numpy as np

MAGIC = b"LLIC"
VERSION = 1
NSEG = 49


def bytestream_list_to_segments(bl):
    if len(bl) != 6 or any(len(r) != 9 for r in bl):
        raise ValueError("bytestream_list must be 6 lists of 9 byte strings")
    return list(bl[0][:4]) + [s for row in bl[1:] for s in row]


def segments_to_bytestream_list(segs):
    if len(segs) != NSEG:
        raise ValueError(f"expected {NSEG} segments, got {len(segs)}")
    em = b""
    return [[segs[0], segs[1], segs[2], segs[3], em, em, em, em, em]] + [list(segs[4 + 9 * s: 13 + 9 * s]) for s in range(5)]


def dumps_llic(bl) -> bytes:
    segs = bytestream_list_to_segments(bl)
    head = MAGIC + bytes([VERSION, len(segs)]) + b"".join(struct.pack("<I", len(s)) for s in segs)
    return head + b"".join(bytes(s) for s in segs)


def loads_llic(buf: bytes):
    if len(buf) < 6 or buf[:4] != MAGIC:
        raise ValueError("not an LLIC file (bad magic)")
    if buf[4] != VERSION:
        raise ValueError(f"unsupported LLIC version {buf[4]}")
    n = buf[5]
    if n != NSEG or len(buf) < 6 + 4 * n:
        raise ValueError("truncated or malformed LLIC index")
    lens = struct.unpack("<%dI" % n, buf[6:6 + 4 * n])
    pos, segs = 6 + 4 * n, []
    if pos + sum(lens) != len(buf):
        raise ValueError("LLIC payload length does not match its index")
    for ln in lens:
        segs.append(bytes(buf[pos:pos + ln]))
        pos += ln
    return segments_to_bytestream_list(segs)


def write_llic(path, bl):
    with open(path, "wb") as fh:
        fh.write(dumps_llic(bl))


def read_llic(path):
    with open(path, "rb") as fh:
        return loads_llic(fh.read())


# ---------------------------------------------------------------------------------------------- images
def _read_ppm(buf: bytes):
    # P6 <ws> W <ws> H <ws> maxval <single ws> data ; '#' comments allowed in the header
    pos, toks = 0, []
    while len(toks) < 4:
        while pos < len(buf) and buf[pos:pos + 1].isspace():
            pos += 1
        if buf[pos:pos + 1] == b"#":
            while pos < len(buf) and buf[pos:pos + 1] != b"\n":
                pos += 1
            continue
        st = pos
        while pos < len(buf) and not buf[pos:pos + 1].isspace():
            pos += 1
        toks.append(buf[st:pos])
    if toks[0] != b"P6":
        raise ValueError("only binary PPM (P6) is supported")
    W, H, mx = int(toks[1]), int(toks[2]), int(toks[3])
    if mx != 255:
        raise ValueError("only 8-bit PPM (maxval 255) is supported")
    data = np.frombuffer(buf, dtype=np.uint8, count=3 * H * W, offset=pos + 1)
    return np.ascontiguousarray(data.reshape(H, W, 3).transpose(2, 0, 1))


def read_image(path) -> np.ndarray:
    """-> uint8 [3, H, W] (RGB).  The reference's loader does PIL .convert('RGB') + ToTensor (uint8 / 255)."""
    with open(path, "rb") as fh:
        head = fh.read(2)
    if head == b"P6":
        with open(path, "rb") as fh:
            return _read_ppm(fh.read())
    try:
        from PIL import Image
    except ImportError as e:                                     # pragma: no cover
        raise RuntimeError(f"{path}: PNG/JPG input needs PIL; binary PPM works without it") from e
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8).transpose(2, 0, 1))


def write_image(path, rgb: np.ndarray):
    rgb = np.asarray(rgb)
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[0] != 3:
        raise ValueError("expected uint8 [3, H, W]")
    hwc = np.ascontiguousarray(rgb.transpose(1, 2, 0))
    if os.path.splitext(path)[1].lower() in (".ppm", ".pnm"):
        with open(path, "wb") as fh:
            fh.write(b"P6\n%d %d\n255\n" % (rgb.shape[2], rgb.shape[1]))
            fh.write(hwc.tobytes())
        return
    from PIL import Image
    Image.fromarray(hwc, "RGB").save(path)
