"""The specialised dither kernel (csrc/nq_dither_fast.hip: LAB, 32 < K <= 256, no semi-transparency, DITHER_MAX 25) and the
float32-filtered lookups it is built from, against the CPU oracle:

* the tile shapes the benchmark and the automatic rule use (8x8, 4x4, automatic), ragged image sizes, dither on and off, with the
  specialised kernel and with the generic one (NQ_OPT_FAST_DITHER off) -- bit for bit against the oracle's tiled restatement;
* nq_nearest_index / nq_closest_tuple over the WHOLE 2^24 opaque colour cube for a 256-colour palette (the float32 Lab
  pre-selection, the fast cube root of the exact path, the float32 closest filter: zero flips allowed);
* the benchmark image itself (4096^2 gradient + noise, 256 colours, automatic 8x8 tiles): a sample of tile rows against the
  oracle, the whole image against the generic kernel."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from nquant.android_amd import synth

pytestmark = pytest.mark.gpu

TILED = 1
OPT_FAST = 2


def _copy_params(src, dst_cls):
    p = dst_cls()
    for f, _ in dst_cls._fields_:
        setattr(p, f, getattr(src, f))
    return p


def _oracle_palette(oracle, kind, img, K):
    q = oracle.OracleQuantizer(kind, img)
    q.prescan(K)
    pal = q.pnnquan(K)
    return q, pal


def auto_tile(w, h):
    """The automatic rule of nq_dither_device (csrc/nq_abi.cpp): 8x8 when that yields >= 131072 chains, else 4x4 (non-sorted queue)."""
    if ((w + 7) // 8) * ((h + 7) // 8) >= 131072:
        return (min(8, w), min(8, h))
    return (min(4, w), min(4, h))


def _opaque_alpha0(img, seed, p=0.02):
    """a few fully transparent pixels, nothing semi-transparent: hasAlpha() without hasSemiTransparency"""
    return synth.with_alpha(img, seed, p_transparent=p, p_semi=0.0)


# K, dither, image, tile (None = automatic), weight (None = what pnnquan derived; a value = injected into BOTH the oracle object and the
# GPU handle -- `weight` = nMaxColors / maxbins selects the GilbertCurve ladder, and only images with > 17 000 histogram bins reach
# the DITHER_MAX = 25 rung of the benchmark configuration by themselves)
FAST_CASES = [
    (256, True, lambda: synth.gradient_noise(100, 52, 131), (8, 8), 0.0115),
    (256, True, lambda: synth.gradient_noise(100, 52, 132), (4, 4), 0.0115),
    (256, False, lambda: synth.gradient_noise(100, 52, 133), (8, 8), 0.0115),
    (256, False, lambda: synth.uniform_rgb(90, 70, 134), (4, 4), 0.0115),
    (256, True, lambda: synth.gradient_noise(203, 117, 135), None, 0.0115),
    (256, False, lambda: synth.gradient_noise(101, 99, 136), None, 0.0115),
    (256, True, lambda: _opaque_alpha0(synth.gradient_noise(96, 80, 138), 138), (8, 8), 0.0115),   # transparent pixels: tiles handed back
    (256, False, lambda: _opaque_alpha0(synth.gradient_noise(64, 80, 145), 145), (4, 4), 0.0115),
    (128, True, lambda: synth.gradient_noise(120, 88, 139), (8, 8), 0.006),
    (256, True, lambda: synth.gradient_noise(96, 96, 146), (8, 8), 0.003),         # margin 8: second stage of ditherPixel
    (48, True, lambda: synth.gradient_noise(96, 96, 140), (8, 8), 0.003),          # 32 < K <= 64, weight < .005: normalDistribution branch
    (40, True, lambda: synth.gradient_noise(80, 96, 147), (8, 8), 0.0026),         # small acceptedDiff: both Y_Diff tests are live
    (200, True, lambda: synth.gradient_noise(64, 48, 141), (7, 5), 0.0115),        # tile width not a multiple of 4: scalar write-out
    (256, True, lambda: synth.gradient_noise(50, 46, 142), (8, 8), 0.0115),        # image width not a multiple of 4
    (256, True, lambda: synth.uniform_rgb(192, 160, 137), (8, 8), None),           # 30 720 pixels, ~24 000 bins: the rung by itself
    (256, False, lambda: synth.uniform_rgb(256, 200, 148), (4, 4), None),
    # further natural cases (nothing injected): 19 819 bins of 1-5-5-5 keys WITH alpha-0 pixels (tiles are handed back to the generic
    # kernel); K = 100 / 21 403 bins (saliencies from the pnnquan pass, nMaxColors < 128); K = 128 / 26 544 bins, ragged 4x4 tiles
    (256, True, lambda: synth.with_alpha(synth.uniform_rgb(192, 160, 301), 301, p_transparent=0.01, p_semi=0.0), (8, 8), None),
    (100, True, lambda: synth.gradient_noise(320, 240, 302, noise=40), (8, 8), None),
    (128, False, lambda: synth.gradient_noise(398, 299, 303, noise=48), None, None),
    (256, True, lambda: synth.few_colors(96, 96, 143, 3000), (8, 8), None),        # sorted-by-yDiff queue: generic kernel only
    (16, True, lambda: synth.gradient_noise(64, 64, 144), (8, 8), None),           # K <= 32: generic kernel only
]


@pytest.mark.parametrize("K,dither,mk,tile,weight", FAST_CASES)
def test_bench_tile_shapes_bit_exact_vs_oracle(nq, oracle, K, dither, mk, tile, weight):
    img = mk()
    seed = 4321
    h, w = img.shape
    otile = tile if tile is not None else auto_tile(w, h)
    oq, pal = _oracle_palette(oracle, 1, img, K)
    op = oq.params
    if weight is not None:
        op.weight = weight
        op.isNano = 1 if weight <= .015 else 0
        oq.set_params(op)
    params = _copy_params(op, nq.Params)
    Kp = len(pal)
    expect_fast = 32 < Kp <= 256 and not op.hasSemiTransparency and 0.0025 < op.weight < 0.015 and op.ratio >= 0
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, dither, tile=otile)
    for fast in (1, 0):
        gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=tile)
        gq.set_params(params)
        gq.set_option(OPT_FAST, fast)
        got_argb, got_idx = gq.dither(pal, dither)
        ran_fast, handed_back = gq.dither_path()
        assert ran_fast == (1 if (fast and expect_fast) else 0), (fast, ran_fast, op.weight)
        if fast and expect_fast and (img.view(np.uint32) >> 24 == 0).any():
            assert handed_back > 0, "an image with alpha-0 pixels must hand tiles back to the generic kernel"
        bad = (got_idx.astype(np.int32) != want_idx).sum()
        assert bad == 0, "fast=%d: index mismatches %d of %d (tiles handed back: %d)" % (fast, bad, want_idx.size, handed_back)
        assert (got_argb != want_argb).sum() == 0


def test_repeated_dither_on_one_handle_is_idempotent(nq, oracle):
    """nq_dither twice on one handle (semi-transparent image: the reference negates `weight` inside dither()) gives the same
    result both times, equal to the oracle's."""
    img = synth.with_alpha(synth.gradient_noise(64, 64, 151), 151)
    seed = 5
    oq, pal = _oracle_palette(oracle, 1, img, 64)
    params = _copy_params(oq.params, nq.Params)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, True, tile=(8, 8))
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(8, 8))
    gq.set_params(params)
    a1, i1 = gq.dither(pal, True)
    a2, i2 = gq.dither(pal, True)
    assert (i1 == i2).all() and (a1 == a2).all()
    assert (i1.astype(np.int32) == want_idx).all() and (a1 == want_argb).all()
    assert gq.params.weight == params.weight


# ---- the whole colour cube ---------------------------------------------------------------------------------------------------
_W = {}


def _cube_worker_init(root, kind, img_bytes, shape, K):
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib
    img = np.frombuffer(img_bytes, np.int32).reshape(shape)
    q = oracle_lib.OracleQuantizer(kind, img)
    q.prescan(K)
    _W["q"] = q
    _W["pal"] = q.pnnquan(K)


def _cube_worker(chunk):
    lo, hi, alpha = chunk
    cols = (np.arange(lo, hi, dtype=np.uint32) | np.uint32(alpha << 24)).view(np.int32)
    q, pal = _W["q"], _W["pal"]
    return lo, hi, q.nearest_index(pal, cols), q.closest_tuple(pal, cols)


@pytest.mark.parametrize("mk,alpha0", [(lambda: synth.gradient_noise(160, 160, 3), False),
                                        (lambda: _opaque_alpha0(synth.uniform_rgb(112, 112, 161), 161), True)])
def test_whole_colour_cube_nearest_and_closest(nq, oracle, mk, alpha0):
    """Every one of the 2^24 opaque colours through nq_nearest_index and nq_closest_tuple (specialised lookups) == the oracle.
    Second palette: an image with transparent pixels (palette[0] transparent, visible colours start their nearest scan at 1)."""
    img = mk()
    K = 256
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    oq, pal = _oracle_palette(oracle, 1, img, K)
    gq = nq.PnnLABQuantizer(img)
    gq.set_params(_copy_params(oq.params, nq.Params))
    assert bool(oq.params.transparentPixelIndex >= 0) == alpha0
    step = 1 << 19
    chunks = [(lo, lo + step, 255) for lo in range(0, 1 << 24, step)]
    workers = max(2, min(12, (os.cpu_count() or 4) - 2))
    bad_n = bad_c = 0
    first = None
    ctx = mp.get_context("spawn")       # the workers only run the CPU oracle: fresh interpreters, no GPU state
    with ctx.Pool(workers, initializer=_cube_worker_init, initargs=(root, 1, img.tobytes(), img.shape, K)) as pool:
        for lo, hi, want_idx, want_tup in pool.imap_unordered(_cube_worker, chunks):
            cols = (np.arange(lo, hi, dtype=np.uint32) | np.uint32(0xFF000000)).view(np.int32)
            got_idx = gq.nearestColorIndex(pal, cols)
            got_tup = gq.closestTuple(pal, cols)
            mi = got_idx != want_idx
            mt = (got_tup != want_tup).any(axis=1)
            if (mi.any() or mt.any()) and first is None:
                j = int(np.flatnonzero(mi | mt)[0])
                first = (hex(int(cols[j]) & 0xFFFFFFFF), int(got_idx[j]), int(want_idx[j]), got_tup[j].tolist(), want_tup[j].tolist())
            bad_n += int(mi.sum())
            bad_c += int(mt.sum())
    assert bad_n == 0 and bad_c == 0, "flips over the colour cube: nearest %d, closest %d, first %s" % (bad_n, bad_c, first)


# ---- the benchmark image at full size ----------------------------------------------------------------------------------------
def test_bench_image_4096_rows_vs_oracle_and_whole_image_vs_generic(nq, oracle):
    """BASELINE cfg 3 exactly as bench.py runs it (4096^2 gradient + noise, seed 3, LAB 256 colours + dither, automatic tiles =
    8x8): the palette and six tile rows (first, last, four in between = 3072 tiles) bit for bit against the oracle, and the whole
    index map of the specialised kernel against the generic kernel's."""
    import torch
    W = H = 4096
    img = synth.gradient_noise(W, H, 3)
    seed = 3
    oq, pal = _oracle_palette(oracle, 1, img, 256)
    oq.set_seed(seed)
    tile = auto_tile(W, H)
    assert tile == (8, 8)
    d_in = torch.from_numpy(img.reshape(-1)).cuda()
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed)
    q.width, q.height = W, H
    outs = {}
    for fast in (1, 0):
        q.set_option(OPT_FAST, fast)
        d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
        gpal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
        torch.cuda.synchronize()
        assert (gpal == pal).all(), "palette differs from the oracle"
        assert q.dither_path()[0] == fast
        outs[fast] = (d_out.cpu().numpy().reshape(H, W), d_idx.cpu().numpy().view(np.uint16).reshape(H, W))
    assert (outs[1][1] == outs[0][1]).all() and (outs[1][0] == outs[0][0]).all(), "specialised kernel != generic kernel"
    rows = [0, 101, 255, 256, 377, 511]
    for r in rows:
        want_argb, want_idx = oq.dither_tile_rows(pal, True, tile, r, 1)
        ys = slice(r * 8, r * 8 + 8)
        assert (outs[1][1][ys].astype(np.int32) == want_idx[ys]).all(), "tile row %d differs from the oracle" % r
        assert (outs[1][0][ys] == want_argb[ys]).all()


# ---- PnnQuantizer (RGB kind) through the specialised kernel ---------------------------------------------------------------------
def _almost_opaque(img, seed, p=0.03):
    """a few pixels with alpha 0xF0 (>= 0xE0: neither transparent nor semi-transparent for the pre-scan): the accumulated alpha leaves 255
    around them, the opaque-colour candidate lists do not apply there and those tiles must be handed back to the generic kernel"""
    z = synth.splitmix64(seed ^ 0xA1FA, img.size).reshape(img.shape)
    out = img.view(np.uint32).copy()
    sel = (z % np.uint64(10000)) < np.uint64(int(p * 10000))
    out[sel] = (out[sel] & np.uint32(0x00FFFFFF)) | np.uint32(0xF0000000)
    return out.view(np.int32)


RGB_FAST_CASES = [  # K, image, tile (None = automatic), expect tiles handed back
    (256, lambda: synth.uniform_rgb(192, 160, 137), (8, 8), False),                     # ~24 000 bins: DITHER_MAX 25 by itself
    (100, lambda: synth.gradient_noise(320, 240, 302, noise=40), (4, 4), False),
    (64, lambda: synth.gradient_noise(200, 150, 511, noise=40), (8, 8), False),
    (128, lambda: synth.gradient_noise(398, 299, 303, noise=48), None, False),          # ragged automatic tiles
    (256, lambda: _almost_opaque(synth.uniform_rgb(192, 160, 512), 512), (8, 8), True),
]


@pytest.mark.parametrize("K,mk,tile,handed", RGB_FAST_CASES)
def test_rgb_kind_through_the_specialised_kernel(nq, oracle, K, mk, tile, handed):
    """PnnQuantizer.convert(K, dither = true) on images without transparency: the RGB instantiation of the specialised dither kernel
    (accumulate -> nearestColorIndex over the cell's candidate list -> limiter) == the oracle's tiled restatement == the generic kernel."""
    img = mk()
    seed = 77
    h, w = img.shape
    otile = tile if tile is not None else auto_tile(w, h)
    oq, pal = _oracle_palette(oracle, 0, img, K)
    op = oq.params
    params = _copy_params(op, nq.Params)
    expect_fast = 32 < len(pal) <= 256 and not op.hasSemiTransparency and op.transparentPixelIndex < 0 and 0.0025 < op.weight < 0.015
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, True, tile=otile)
    for fast in (1, 0):
        gq = nq.PnnQuantizer(img, mode=TILED, seed=seed, tile=tile)
        gq.set_params(params)
        gq.set_option(OPT_FAST, fast)
        got_argb, got_idx = gq.dither(pal, True)
        ran_fast, handed_back = gq.dither_path()
        assert ran_fast == (1 if (fast and expect_fast) else 0), (fast, ran_fast, op.weight, len(pal))
        if fast and expect_fast:
            assert (handed_back > 0) == handed, handed_back
        bad = (got_idx.astype(np.int32) != want_idx).sum()
        assert bad == 0, "fast=%d: index mismatches %d of %d (tiles handed back: %d)" % (fast, bad, want_idx.size, handed_back)
        assert (got_argb != want_argb).sum() == 0


def test_photo_workload_4096_palette_and_rows_vs_oracle(nq, oracle):
    """bench.py's `photo` workload at its full size -- the reference's sample photograph tiled to 4096^2 (2970 bins: weight 0.086, the
    sorted-by-yDiff queue with DITHER_MAX 9, generic kernel, automatic 8x8 tiles whose chains start in the queue's steady state, the
    LDS-resident PriorityQueue) -- palette, scalars and four tile rows (2048 tiles) against the oracle, for both quantizer kinds."""
    import os
    import torch
    W = H = 4096
    rgb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_495x438.npz"))["rgb"]
    img = synth.tile_photo(rgb, W, H, 3)
    seed = 3
    d_in = torch.from_numpy(img.reshape(-1)).cuda()
    for kind in (1, 0):
        oq, pal = _oracle_palette(oracle, kind, img, 256)
        oq.set_seed(seed)
        q = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(np.zeros((1, 1), np.int32), mode=TILED, seed=seed)
        q.width, q.height = W, H
        d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
        gpal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
        torch.cuda.synchronize()
        assert len(gpal) == len(pal) and (gpal == pal).all(), "kind %d: palette differs from the oracle" % kind
        po, pg = oq.params, q.params
        assert (po.maxbins, po.quan_rt, po.isNano, po.texicab, po.weight, po.ratio) == (pg.maxbins, pg.quan_rt, pg.isNano, pg.texicab, pg.weight, pg.ratio)
        assert q.dither_path()[0] == 0                    # sorted queue: the generic kernel
        idx = d_idx.cpu().numpy().view(np.uint16).reshape(H, W).astype(np.int32)
        out = d_out.cpu().numpy().reshape(H, W)
        for r in (0, 54, 300, 511):                       # (row 54: a tile row that straddles two copies of the picture)
            want_argb, want_idx = oq.dither_tile_rows(pal, True, (8, 8), r, 1)
            ys = slice(r * 8, r * 8 + 8)
            assert (idx[ys] == want_idx[ys]).all(), "kind %d tile row %d differs from the oracle" % (kind, r)
            assert (out[ys] == want_argb[ys]).all()
