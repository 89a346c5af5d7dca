#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference is Java and cannot run here; it ships no golden
vectors of its own).  Inputs come from the seeded generators in nquant.android_amd/synth.py, so the
fixtures hold only the expected outputs.  Run from the repo root: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O                                    # noqa: E402
from nquant.android_amd import synth                      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # BASELINE cfg 1: 64x64 uniform RGB seed 1, PnnQuantizer.convert(16, dither=false): fully deterministic
    "cfg1_rgb16_64x64": dict(kind=0, K=16, dither=False, img=lambda: synth.uniform_rgb(64, 64, 1), seed=0, tile=None),
    # LAB 256 colours, sequential reference semantics with an injected seed
    "lab256_seq_64x64": dict(kind=1, K=256, dither=True, img=lambda: synth.gradient_noise(64, 64, 2), seed=7, tile=None),
    # LAB 256 colours, the tiled decomposition the GPU runs (16x16 tiles)
    "lab256_tiled_96x80": dict(kind=1, K=256, dither=True, img=lambda: synth.gradient_noise(96, 80, 3), seed=7, tile=(16, 16)),
    "lab256_tiled_nodither_96x80": dict(kind=1, K=256, dither=False, img=lambda: synth.gradient_noise(96, 80, 3), seed=7, tile=(16, 16)),
    "lab64_alpha_tiled_64x64": dict(kind=1, K=64, dither=True, img=lambda: synth.with_alpha(synth.gradient_noise(64, 64, 4), 4), seed=5, tile=(16, 16)),
}


# palette only: 64 309 occupied bins (uniform random colours), the size at which the merge loop's LDS mirrors of the heap and of
# mtm are too small in every workgroup variant and all 1005 position blocks exist
PALETTE_CASES = {
    "lab256_palette_uniform_512x512": dict(kind=1, K=256, img=lambda: synth.uniform_rgb(512, 512, 7)),
    # the RGB kind at the same size (block pruning of the RGB scan, its unpruned fallback, the pruned initial pass), and with
    # semi-transparent pixels (alpha term of the gate, which the boxes ignore)
    "rgb256_palette_uniform_512x512": dict(kind=0, K=256, img=lambda: synth.uniform_rgb(512, 512, 7)),
    "rgb64_palette_alpha_gradient_384x384": dict(kind=0, K=64, img=lambda: synth.with_alpha(synth.gradient_noise(384, 384, 9), 9)),
}


# BASELINE cfg 2 end to end: 1024x1024 uniform random opaque RGB, seed 2 (all 65 536 bins occupied -> isNano), PnnLABQuantizer, 256
# colours, no diffusion: the palette of the image's OWN pnnquan and the per-pixel nearestColorIndex map (cache-miss semantics =
# MODE_LOOKUP_ONLY).  The fixture holds the palette, the scalars, the SHA-256 of the uint16 index map and every 61st index.
LOOKUP_CASES = {
    "cfg2_lab256_lookup_uniform_1024x1024": dict(kind=1, K=256, img=lambda: synth.uniform_rgb(1024, 1024, 2), stride=61),
}


# The reference's only input asset, decoded (tools/extract_sample_pixels.py -> sample_495x438.npz, pixels only): the picture the demo
# activity quantizes with `new PnnQuantizer(path).convert(256, true)` (app/src/main/java/nQuant/android/MainActivity.java:190-194) and
# the README's `new PnnLABQuantizer(path).convert(256, true)`.  A photographic histogram: 2970 bins -> weight 0.086 (colour-keyed
# caches, not isNano), quan_rt 1, GilbertCurve sorted-by-yDiff queue with DITHER_MAX 9 -- rungs no synthetic generator reaches at K = 256.
# Per case: palette + scalars, the whole convert in REFERENCE_SEQUENTIAL mode, and the tiled decomposition (8x8 tiles, ragged at the right
# and bottom edges of the 495 x 438 picture; the chains of the sorted queue start in its steady state, oracle: gilbert_run).
def sample_image():
    rgb = np.load(os.path.join(OUT, "sample_495x438.npz"))["rgb"].astype(np.uint32)
    return ((np.uint32(255) << np.uint32(24)) | (rgb[..., 0] << np.uint32(16)) | (rgb[..., 1] << np.uint32(8)) | rgb[..., 2]).view(np.int32)


SAMPLE_TILE = (8, 8)
SAMPLE_CASES = {
    "sample_rgb256_dither": dict(kind=0, K=256, dither=True, seed=7),
    "sample_lab256_dither": dict(kind=1, K=256, dither=True, seed=7),
    "sample_rgb256_nodither": dict(kind=0, K=256, dither=False, seed=7),
    "sample_lab256_nodither": dict(kind=1, K=256, dither=False, seed=7),
    "sample_lab16_dither": dict(kind=1, K=16, dither=True, seed=7),
}


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def run_sample_case(c):
    img = sample_image()
    q = O.OracleQuantizer(c["kind"], img, seed=c["seed"])
    q.prescan(c["K"])
    pal = q.pnnquan(c["K"])
    p = q.params
    scal = np.array([p.hasSemiTransparency, p.transparentPixelIndex, p.transparentColor, p.isNano, p.texicab, p.quan_rt,
                     p.maxbins, p.paletteLength], np.int64)
    dbl = np.array([p.PR, p.PG, p.PB, p.PA, p.ratio, p.weight], np.float64)
    q.set_seed(c["seed"])
    seq_argb, seq_idx = q.dither(pal, c["dither"])
    distinct_seq = np.int64(q.params.distinctColors)
    q2 = O.OracleQuantizer(c["kind"], img, seed=c["seed"])
    q2.prescan(c["K"])
    assert (q2.pnnquan(c["K"]) == pal).all()
    q2.set_seed(c["seed"])
    til_argb, til_idx = q2.dither(pal, c["dither"], tile=SAMPLE_TILE)
    return dict(palette=pal, scalars=scal, doubles=dbl, seq_index=seq_idx.astype(np.uint8), seq_argb_sha256=_sha(seq_argb),
                tiled_index=til_idx.astype(np.uint8), tiled_argb_sha256=_sha(til_argb), distinct_seq=distinct_seq,
                distinct_tiled=np.int64(q2.params.distinctColors))


# BASELINE cfg 5: 16384 x 16384 gradient + noise, seed 5, PnnLABQuantizer 256 colours: the palette and scalars of the WHOLE image
# (2^28 pixels through the oracle's histogram: minutes of CPU time, so the fixture is produced here and not on the GPU box), plus
# an order-sensitive checksum of the input so that the GPU-side test knows it quantizes the very same pixels.
BIG_CASES = {
    "cfg5_lab256_palette_16384x16384": dict(kind=1, K=256, width=16384, height=16384, seed=5),
}


def image_checksum(img):
    """(sum, position-weighted sum) of the pixels as uint64, wrapping: cheap on the host and with torch on the device."""
    with np.errstate(over="ignore"):
        v = img.reshape(-1).view(np.uint32).astype(np.uint64)
        w = (np.arange(v.size, dtype=np.uint64) & np.uint64(0xFFFF)) + np.uint64(1)
        return np.array([v.sum(dtype=np.uint64), (v * w).sum(dtype=np.uint64)], np.uint64)


def run_big_case(c):
    img = synth.gradient_noise_banded(c["width"], c["height"], c["seed"])
    q = O.OracleQuantizer(c["kind"], img)
    q.prescan(c["K"])
    pal = q.pnnquan(c["K"])
    p = q.params
    scal = np.array([p.hasSemiTransparency, p.transparentPixelIndex, p.transparentColor, p.isNano, p.texicab, p.quan_rt,
                     p.maxbins, p.paletteLength], np.int64)
    dbl = np.array([p.PR, p.PG, p.PB, p.PA, p.ratio, p.weight], np.float64)
    return dict(palette=pal, scalars=scal, doubles=dbl, distinct=np.int64(p.distinctColors), checksum=image_checksum(img))


def run_lookup_case(c):
    import hashlib
    img = c["img"]()
    q = O.OracleQuantizer(c["kind"], img)
    q.prescan(c["K"])
    pal = q.pnnquan(c["K"])
    p = q.params
    idx = q.nearest_index(pal, img.reshape(-1)).astype(np.uint16)
    return dict(palette=pal, scalars=np.array([p.maxbins, p.isNano, p.texicab, p.quan_rt], np.int64),
                doubles=np.array([p.ratio, p.weight], np.float64),
                index_sha256=np.frombuffer(hashlib.sha256(idx.tobytes()).digest(), np.uint8).copy(),
                index_sample=idx[::c["stride"]].copy(), index_histogram=np.bincount(idx, minlength=len(pal)).astype(np.int64))


def run_palette_case(c):
    q = O.OracleQuantizer(c["kind"], c["img"]())
    q.prescan(c["K"])
    pal = q.pnnquan(c["K"])
    p = q.params
    return dict(palette=pal, scalars=np.array([p.maxbins, p.isNano, p.texicab, p.quan_rt], np.int64),
                doubles=np.array([p.ratio, p.weight], np.float64))


def run_case(c):
    img = c["img"]()
    q = O.OracleQuantizer(c["kind"], img, seed=c["seed"])
    q.prescan(c["K"])
    pal = q.pnnquan(c["K"])
    p = q.params
    q.set_seed(c["seed"])
    argb, idx = q.dither(pal, c["dither"], tile=c["tile"])
    scal = np.array([p.hasSemiTransparency, p.transparentPixelIndex, p.transparentColor, p.isNano, p.texicab, p.quan_rt,
                     p.maxbins, p.paletteLength], np.int64)
    dbl = np.array([p.PR, p.PG, p.PB, p.PA, p.ratio, p.weight], np.float64)
    return dict(palette=pal, index=idx.astype(np.uint16), argb=argb, scalars=scal, doubles=dbl, distinct=np.int64(p.distinctColors))


if __name__ == "__main__":
    for name, c in CASES.items():
        r = run_case(c)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
        print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][6]))
    for name, c in SAMPLE_CASES.items():
        r = run_sample_case(c)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
        print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][6]), "distinct", int(r["distinct_seq"]), int(r["distinct_tiled"]))
    if "--big" in sys.argv:          # ~10 minutes, 4 GB
        for name, c in BIG_CASES.items():
            r = run_big_case(c)
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
            print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][6]), "distinct", int(r["distinct"]))
    for name, c in LOOKUP_CASES.items():
        r = run_lookup_case(c)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
        print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][0]))
    for name, c in PALETTE_CASES.items():
        r = run_palette_case(c)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
        print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][0]))
