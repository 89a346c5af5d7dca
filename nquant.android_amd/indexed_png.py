"""Indexed-colour PNG writer for the index map + palette the quantizer emits (the reference stops at an ARGB
Bitmap and notes that Android cannot show indexed formats, /root/reference/README.md:24; palette + u8 indices, 5 B/pixel on the
GPU side, is the natural on-disk form).  Pure Python (zlib), no quantizer arithmetic here."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_indexed_png(path, index, palette_argb, compress_level=6):
    """index: (H, W) integer array with values < len(palette) <= 256; palette_argb: ARGB_8888 int32/uint32 entries.
    Writes colour type 3 (PLTE) with a tRNS chunk when any palette entry is not opaque."""
    index = np.ascontiguousarray(index)
    h, w = index.shape
    pal = np.asarray(palette_argb).astype(np.int64) & 0xFFFFFFFF
    if len(pal) > 256:
        raise ValueError("an indexed PNG holds at most 256 palette entries")
    if index.max(initial=0) >= len(pal):
        raise ValueError("index out of palette range")
    idx8 = index.astype(np.uint8)
    plte = bytearray()
    trns = bytearray()
    for c in pal:
        plte += bytes(((c >> 16) & 0xFF, (c >> 8) & 0xFF, c & 0xFF))
        trns.append((c >> 24) & 0xFF)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), idx8], axis=1).tobytes()     # filter type 0 per scanline
    png = b"\x89PNG\r\n\x1a\n"
    png += _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0))
    png += _chunk(b"PLTE", bytes(plte))
    if any(a != 255 for a in trns):
        png += _chunk(b"tRNS", bytes(trns))
    png += _chunk(b"IDAT", zlib.compress(raw, compress_level))
    png += _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)
    return len(png)
