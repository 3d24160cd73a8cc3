"""24-bit BMP <-> planar (3,H,W) uint8 RGB, the on-disk format either side of the path.

Mirrors what the reference reads and writes through CImg (load_bmp CImg.h:48376, save_bmp CImg.h:52605):
uncompressed 24 bpp, rows bottom-up, BGR byte order, rows padded to 4 bytes.  Host-side I/O only.
"""
import struct

import numpy as np


def load_bmp(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"BM":
        raise ValueError(f"{path}: not a BMP file")
    off = struct.unpack_from("<I", data, 10)[0]
    hdr_size, w, h, planes, bpp, comp = struct.unpack_from("<IiiHHI", data, 14)
    if bpp != 24 or comp != 0:
        raise ValueError(f"{path}: only uncompressed 24-bit BMP is supported (bpp={bpp}, compression={comp})")
    bottom_up = h > 0
    h = abs(h)
    stride = (3 * w + 3) & ~3
    rows = np.frombuffer(data, np.uint8, count=stride * h, offset=off).reshape(h, stride)[:, : 3 * w].reshape(h, w, 3)
    if bottom_up:
        rows = rows[::-1]
    return np.ascontiguousarray(rows[:, :, ::-1].transpose(2, 0, 1))  # BGR interleaved -> planar RGB


def save_bmp(path, img):
    img = np.asarray(img, np.uint8)
    assert img.ndim == 3 and img.shape[0] == 3
    _, h, w = img.shape
    stride = (3 * w + 3) & ~3
    rows = np.zeros((h, stride), np.uint8)
    bgr = img.transpose(1, 2, 0)[:, :, ::-1]          # H,W,(B,G,R)
    rows[:, : 3 * w] = bgr[::-1].reshape(h, 3 * w)    # bottom-up
    size = 54 + stride * h
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", size, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, stride * h, 2835, 2835, 0, 0))
        f.write(rows.tobytes())
