"""Quality of the PARALLEL_TILED decomposition against the sequential reference semantics (CPU oracle on both sides):
per-pixel CIE76 deltaE against the SOURCE image (p50 / p99 / max) for the sequential and the tiled output, deltaE between
the two outputs after an 8x8 box filter (do the dither patterns integrate to the same colours?), and a tile-seam metric (mean
absolute Lab step across tile boundaries relative to the same step one pixel inside the tiles).
Usage: python tests/tiled_vs_sequential.py [size] [tile]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import oracle_lib
from nquant.android_amd import synth


def lab_image(argb):
    u, inv = np.unique(argb.reshape(-1), return_inverse=True)
    lab = np.array([oracle_lib.rgb2lab(int(c))[1:] for c in u], np.float64)
    return lab[inv].reshape(argb.shape + (3,))


def box(a, k):
    h, w = a.shape[0] // k * k, a.shape[1] // k * k
    return a[:h, :w].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


def seam_ratio(lab, tile):
    d = np.linalg.norm(np.diff(lab, axis=1), axis=2)            # horizontal steps, column c -> c + 1
    cols = np.arange(d.shape[1])
    on = (cols % tile) == tile - 1
    inside = (cols % tile) == tile // 2 - 1
    return float(d[:, on].mean() / max(d[:, inside].mean(), 1e-9))


def stats(img, K, tile, seed):
    q = oracle_lib.OracleQuantizer(1, img, seed=seed)
    q.prescan(K)
    pal = q.pnnquan(K)
    q.set_seed(seed)
    seq, _ = q.dither(pal, True)
    q.set_seed(seed)
    til, _ = q.dither(pal, True, tile=(tile, tile))
    src, ls, lt = lab_image(img), lab_image(seq), lab_image(til)
    es, et = np.linalg.norm(ls - src, axis=2).ravel(), np.linalg.norm(lt - src, axis=2).ravel()
    db = np.linalg.norm(box(ls, 8) - box(lt, 8), axis=2).ravel()
    dsrc_s = np.linalg.norm(box(ls, 8) - box(src, 8), axis=2).ravel()
    dsrc_t = np.linalg.norm(box(lt, 8) - box(src, 8), axis=2).ravel()
    return {"K": len(pal), "seq_vs_src p50/p99/max": [round(float(np.percentile(es, p)), 2) for p in (50, 99, 100)],
            "tiled_vs_src p50/p99/max": [round(float(np.percentile(et, p)), 2) for p in (50, 99, 100)],
            "box8 tiled_vs_seq p50/p99/max": [round(float(np.percentile(db, p)), 2) for p in (50, 99, 100)],
            "box8 seq_vs_src mean": round(float(dsrc_s.mean()), 3), "box8 tiled_vs_src mean": round(float(dsrc_t.mean()), 3),
            "seam ratio seq": round(seam_ratio(ls, tile), 3), "seam ratio tiled": round(seam_ratio(lt, tile), 3)}


if __name__ == "__main__":
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    full = synth.gradient_noise(4096, 4096, 3)
    cases = {"bench image crop (4096^2 gradient+noise, top-left %d^2)" % size: full[:size, :size].copy(),
             "gradient_noise %d^2" % size: synth.gradient_noise(size, size, 61),
             "uniform_rgb %d^2" % size: synth.uniform_rgb(size, size, 62)}
    for name, img in cases.items():
        print(name, stats(img, 256, tile, 9), flush=True)
