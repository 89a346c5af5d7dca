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
    for name, c in LOOKUP_CASES.items():
        r = run_lookup_case(c)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
        print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][0]))
    for name, c in PALETTE_CASES.items():
        r = run_palette_case(c)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **r)
        print(name, "K", len(r["palette"]), "maxbins", int(r["scalars"][0]))
