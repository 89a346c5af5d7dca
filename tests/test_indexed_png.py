"""The indexed-PNG writer round-trips an index map + palette (decoded with PIL when available, else by parsing the chunks)."""
import struct
import zlib

import numpy as np


def _decode(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF)
        chunks.setdefault(tag, b"")
        chunks[tag] += body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (depth, ctype) == (8, 3)
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, w + 1)
    assert (raw[:, 0] == 0).all()
    pal = np.frombuffer(chunks[b"PLTE"], np.uint8).reshape(-1, 3)
    trns = np.frombuffer(chunks.get(b"tRNS", b""), np.uint8)
    return raw[:, 1:], pal, trns


def test_indexed_png_round_trip(tmp_path):
    from nquant.android_amd.indexed_png import write_indexed_png
    rng = np.random.default_rng(5)
    pal = (rng.integers(0, 1 << 24, 200, dtype=np.int64) | (0xFF << 24)).astype(np.uint32)
    pal[0] = 0x00FFFFFF          # a transparent entry -> tRNS
    idx = rng.integers(0, 200, (37, 53))
    p = tmp_path / "q.png"
    write_indexed_png(str(p), idx, pal.view(np.int32))
    got_idx, got_pal, trns = _decode(str(p))
    assert (got_idx == idx).all()
    want_rgb = np.stack([(pal >> 16) & 0xFF, (pal >> 8) & 0xFF, pal & 0xFF], axis=1)
    assert (got_pal == want_rgb).all()
    assert trns[0] == 0 and (trns[1:] == 255).all()
    try:
        from PIL import Image
        im = Image.open(str(p))
        assert im.mode == "P" and im.size == (53, 37)
        assert (np.asarray(im) == idx).all()
    except ImportError:
        pass
