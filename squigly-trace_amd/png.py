"""Minimal 8-bit RGB PNG writer (role of massiv-io's writeImage, src/Lib.hs:75)."""
import struct
import zlib

import numpy as np


def write_png(path, rgb: np.ndarray):
    """rgb: (rows, cols, 3) uint8, row-major — the layout `Array S Ix2 (Pixel RGB Word8)` has."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    rows, cols, _ = rgb.shape
    raw = np.concatenate([np.zeros((rows, 1), np.uint8), rgb.reshape(rows, cols * 3)], axis=1).tobytes()

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
