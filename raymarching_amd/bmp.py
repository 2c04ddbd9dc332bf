"""Decoder for uncompressed 24-bit BMP files (the reference's only asset format).

The reference decodes `assets/textures/Floor128.bmp` with the `image` crate
(texture.rs:14, `image::open(path).into_rgb8()`); for an uncompressed 24-bit
file that is a lossless reshuffle into top-down RGB8 rows, which is all this
module does.
"""
import struct

import numpy as np


def load_bmp_rgb8(path):
    """Return (rgb, width, height): rgb is a C-contiguous uint8 array [h][w][3], rows top-down."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"BM":
        raise ValueError(f"{path}: not a BMP file")
    pixel_offset = struct.unpack_from("<I", data, 10)[0]
    header_size = struct.unpack_from("<I", data, 14)[0]
    if header_size < 40:
        raise ValueError(f"{path}: unsupported BMP header ({header_size} bytes)")
    width, height, planes, bpp, compression = struct.unpack_from("<iiHHI", data, 18)
    if planes != 1 or bpp != 24 or compression != 0:
        raise ValueError(f"{path}: only uncompressed 24-bit BMP is supported")
    bottom_up = height > 0
    height = abs(height)
    stride = (width * 3 + 3) & ~3
    rows = np.frombuffer(data, dtype=np.uint8, count=stride * height, offset=pixel_offset)
    rows = rows.reshape(height, stride)[:, : width * 3].reshape(height, width, 3)
    if bottom_up:
        rows = rows[::-1]
    rgb = np.ascontiguousarray(rows[:, :, ::-1])  # file order is B,G,R
    return rgb, width, height
