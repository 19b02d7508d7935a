"""Minimal PNG writer (numpy + zlib) for eyeballing renders."""
import struct
import zlib

import numpy as np


def write_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w, c = rgba.shape
    assert c == 4
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
    png += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    open(path, "wb").write(png)
