"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.  Bit-exact bar for every
integer / index / palette output (SURVEY.md 8c contract items 1-5); PARALLEL_TILED is compared bit for bit against the
oracle's tiled restatement and its deviation from the sequential reference is reported as a CIE76 deltaE."""
import ctypes as C

import numpy as np
import pytest

from nquant.android_amd import synth

pytestmark = pytest.mark.gpu

SEQ, TILED, LOOKUP = 0, 1, 2


def _copy_params(src, dst_cls):
    p = dst_cls()
    for f, _ in dst_cls._fields_:
        setattr(p, f, getattr(src, f))
    return p


def _colors(n, seed, alpha_mix=False):
    z = synth.splitmix64(seed, n)
    c = (z & np.uint64(0xFFFFFF)).astype(np.uint32)
    a = np.full(n, 255, np.uint32)
    if alpha_mix:
        a = ((z >> np.uint64(24)) & np.uint64(0xFF)).astype(np.uint32)
        a[::7] = 255
        a[::11] = 0
    return (c | (a << np.uint32(24))).view(np.int32)


def _oracle_palette(oracle, kind, img, K):
    q = oracle.OracleQuantizer(kind, img)
    q.prescan(K)
    pal = q.pnnquan(K)
    return q, pal


CASES = [  # kind, K, image factory, alpha colours
    (1, 256, lambda: synth.gradient_noise(96, 96, 11), False),     # LAB K>32: |dL| + sqrt(dA^2+dB^2)
    (1, 24, lambda: synth.gradient_noise(64, 64, 12), False),      # LAB 16..32: CIEDE2000
    (1, 8, lambda: synth.uniform_rgb(48, 48, 13), False),          # LAB K<16: squared Lab
    (1, 4, lambda: synth.uniform_rgb(48, 48, 14), False),          # LAB K<=4: RGB
    (1, 64, lambda: synth.with_alpha(synth.gradient_noise(64, 64, 15), 15), True),   # semi-transparent
    (0, 16, lambda: synth.uniform_rgb(64, 64, 1), False),          # RGB
    (0, 256, lambda: synth.with_alpha(synth.uniform_rgb(96, 96, 16), 16), True),
]


@pytest.mark.parametrize("kind,K,mk,alpha", CASES)
def test_nearest_and_closest_bit_exact(nq, oracle, kind, K, mk, alpha):
    img = mk()
    oq, pal = _oracle_palette(oracle, kind, img, K)
    cols = _colors(40000, 100 + K, alpha)
    want_idx = oq.nearest_index(pal, cols)
    want_tup = oq.closest_tuple(pal, cols)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img)
    gq.set_params(_copy_params(oq.params, nq.Params))
    for use_lists in (1, 0):        # per-colour-cell candidate lists (exact acceleration) on, then the plain full scans
        gq.set_option(1, use_lists)
        got_idx = gq.nearestColorIndex(pal, cols)
        got_tup = gq.closestTuple(pal, cols)
        assert (got_idx != want_idx).sum() == 0, "nearest mismatches (lists=%d): %d" % (use_lists, (got_idx != want_idx).sum())
        assert (got_tup != want_tup).any(axis=1).sum() == 0, "closest mismatches (lists=%d)" % use_lists


PAL_CASES = [
    (0, 16, lambda: synth.uniform_rgb(64, 64, 1)),                              # BASELINE cfg 1
    (0, 256, lambda: synth.gradient_noise(128, 128, 21)),
    (0, 8, lambda: synth.with_alpha(synth.uniform_rgb(64, 64, 22), 22)),        # quan_rt = -1 (cbrt)
    (1, 256, lambda: synth.uniform_rgb(112, 112, 2)),
    (1, 256, lambda: synth.gradient_noise(160, 160, 3)),
    (1, 64, lambda: synth.with_alpha(synth.gradient_noise(96, 96, 23), 23)),
    (1, 16, lambda: synth.gradient_noise(96, 96, 24)),
    (1, 300, lambda: synth.uniform_rgb(96, 96, 25)),
    # the ratio ladder goes NEGATIVE here (4 colours out of 400 bins, NQ/PnnLABQuantizer.java:259-264): no interval bound holds,
    # the scans must fall back to the exact path (found by tests/fuzz_parity.py)
    (1, 4, lambda: synth.few_colors(128, 148, 508842683, 402)),
    # 64 histogram bins with ~2300 pixels each, every colour of a bin present: the LAB histogram's per-bin colour table
    # (hist_segments_kernel, nq_palette.inc) for 5-6-5 keys, for 1-5-5-5 keys (a fully transparent pixel), and with pixels whose
    # alpha (0xF0) is neither "semi-transparent" nor 255, which bypass the table inside a tabled bin
    (1, 16, lambda: _crowded_bins(0)),
    (1, 16, lambda: _crowded_bins(1)),
    (1, 16, lambda: _crowded_bins(2)),
]


def _crowded_bins(variant):
    img = synth.uniform_rgb(384, 384, 41 + variant)
    img = (img & np.int32(0x00C7C3C7)) | np.int32(-16777216)
    if variant == 1:
        img = (img & np.int32(0x00C7C7C7)) | np.int32(-16777216)
        img[5, 7] = 0x00FFFFFF
    if variant == 2:
        img[::3, ::5] = (img[::3, ::5] & 0x00FFFFFF) | np.int32(0xF0000000 - (1 << 32))
    return img


@pytest.mark.parametrize("kind,K,mk", PAL_CASES)
def test_pnnquan_palette_and_scalars_bit_exact(nq, oracle, kind, K, mk):
    img = mk()
    oq, want = _oracle_palette(oracle, kind, img, K)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img)
    got = gq.pnnquan(K)
    op, gp = oq.params, gq.params
    for f in ("hasSemiTransparency", "transparentPixelIndex", "transparentColor", "maxbins", "quan_rt", "isNano", "texicab",
              "paletteLength", "PR", "PG", "PB", "PA", "ratio", "weight"):
        assert getattr(op, f) == getattr(gp, f), (f, getattr(op, f), getattr(gp, f))
    assert len(got) == len(want)
    assert (got != want).sum() == 0, "palette mismatches: %d of %d" % ((got != want).sum(), len(want))


@pytest.mark.parametrize("variant,fat_min", [(0, 64), (1, 100), (2, 1000), (3, 64)])
def test_fat_bins_take_the_workgroup_per_bin_histogram(nq, oracle, monkeypatch, variant, fat_min):
    """RGB kind: bins of NQ_HIST_FAT_MIN pixels and more (default 16 384) leave hist_segments_kernel for hist_fat_rgb_kernel
    (csrc/nq_palette.inc), where a whole workgroup strides over the bin (exact integer sums).  Forced here on small images: ~2300-pixel
    bins under the three key forms, and a semi-transparent image (4-4-4-4 keys).  (The LAB kind keeps one wavefront per bin: its float32
    chain is sequential, nq_palette.inc.)"""
    monkeypatch.setenv("NQ_HIST_FAT_MIN", str(fat_min))
    img = _crowded_bins(variant) if variant < 3 else synth.with_alpha(_crowded_bins(0), 7)
    oq, want = _oracle_palette(oracle, 0, img, 16)
    gq = nq.PnnQuantizer(img)
    got = gq.pnnquan(16)
    assert oq.params.maxbins == gq.params.maxbins
    assert len(got) == len(want) and (got == want).all()
    monkeypatch.delenv("NQ_HIST_FAT_MIN")
    assert (nq.PnnQuantizer(img).pnnquan(16) == want).all()


def _lab_of(oracle, argb):
    u, inv = np.unique(argb.reshape(-1), return_inverse=True)
    lab = np.array([oracle.rgb2lab(int(c))[1:] for c in u], np.float64)
    return lab[inv]


DITHER_CASES = [
    (1, 256, True, lambda: synth.gradient_noise(128, 96, 31), (16, 16)),
    (1, 256, True, lambda: synth.uniform_rgb(80, 72, 32), (16, 16)),
    (1, 256, False, lambda: synth.gradient_noise(96, 96, 33), (16, 16)),      # + BlueNoise pass
    (1, 64, True, lambda: synth.with_alpha(synth.gradient_noise(96, 80, 34), 34), (16, 8)),
    (1, 16, True, lambda: synth.gradient_noise(64, 64, 35), (16, 16)),
    (1, 256, True, lambda: synth.few_colors(96, 96, 36, 3000), (16, 16)),      # few bins -> sorted-by-yDiff queue
    (0, 16, False, lambda: synth.uniform_rgb(64, 64, 1), (16, 16)),            # cfg 1 semantics
    (0, 256, True, lambda: synth.gradient_noise(96, 96, 37), (32, 8)),
    (0, 256, False, lambda: synth.with_alpha(synth.uniform_rgb(72, 72, 38), 38), (16, 16)),
]


@pytest.mark.parametrize("kind,K,dither,mk,tile", DITHER_CASES)
def test_dither_tiled_bit_exact_vs_oracle_tiled(nq, oracle, kind, K, dither, mk, tile):
    img = mk()
    seed = 1234
    oq, pal = _oracle_palette(oracle, kind, img, K)
    params = _copy_params(oq.params, nq.Params)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, dither, tile=tile)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=TILED, seed=seed, tile=tile)
    gq.set_params(params)
    got_argb, got_idx = gq.dither(pal, dither)
    bad = (got_idx.astype(np.int32) != want_idx).sum()
    assert bad == 0, "index mismatches: %d of %d" % (bad, want_idx.size)
    assert (got_argb != want_argb).sum() == 0


SEQ_CASES = [
    (1, 256, True, lambda: synth.gradient_noise(48, 40, 41)),
    (1, 64, True, lambda: synth.uniform_rgb(40, 40, 42)),
    (0, 16, False, lambda: synth.uniform_rgb(64, 64, 1)),
    (0, 256, False, lambda: synth.gradient_noise(48, 48, 43)),
    (1, 16, False, lambda: synth.gradient_noise(40, 48, 44)),
]


@pytest.mark.parametrize("kind,K,dither,mk", SEQ_CASES)
def test_reference_sequential_bit_exact_vs_oracle(nq, oracle, kind, K, dither, mk):
    img = mk()
    seed = 77
    oq, pal = _oracle_palette(oracle, kind, img, K)
    params = _copy_params(oq.params, nq.Params)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, dither)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=SEQ, seed=seed)
    gq.set_params(params)
    got_argb, got_idx = gq.dither(pal, dither)
    assert (got_idx.astype(np.int32) != want_idx).sum() == 0
    assert (got_argb != want_argb).sum() == 0


def test_lookup_only_1024_bit_exact(nq, oracle):
    """BASELINE cfg 2: 1024x1024, LAB, 256 colours, no dither: index map == per-pixel nearestColorIndex of the oracle."""
    img = synth.uniform_rgb(1024, 1024, 2)
    small = synth.uniform_rgb(96, 96, 2)
    oq, pal = _oracle_palette(oracle, 1, small, 256)
    params = _copy_params(oq.params, nq.Params)
    gq = nq.PnnLABQuantizer(img, mode=LOOKUP)
    gq.set_params(params)
    got_argb, got_idx = gq.dither(pal, False)
    want = oq.nearest_index(pal, img.reshape(-1)).reshape(img.shape)
    assert (got_idx.astype(np.int32) != want).sum() == 0
    assert (got_argb != pal[want]).sum() == 0


def test_convert_end_to_end_tiled(nq, oracle):
    """convert(256, true) on the GPU == oracle prescan + pnnquan + tiled dither."""
    img = synth.gradient_noise(144, 112, 51)
    seed = 5
    oq, pal = _oracle_palette(oracle, 1, img, 256)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, True, tile=(16, 16))
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(16, 16))
    out = gq.convert(256, True)
    assert (out.palette != pal).sum() == 0
    assert (out.index.astype(np.int32) != want_idx).sum() == 0
    assert (out.argb != want_argb).sum() == 0
    assert (out.argb == out.palette[out.index]).all()
    ms = gq.stage_ms()
    assert ms["total"] > 0


FEW_CASES = [
    (256, lambda: synth.few_colors(64, 64, 71, 100)),                                   # pixelMap.size() <= nMaxColors
    (256, lambda: synth.with_alpha(synth.few_colors(64, 64, 72, 60), 72)),              # + transparent / semi-transparent
    (16, lambda: synth.few_colors(48, 48, 73, 12)),
    (64, lambda: synth.few_colors(64, 48, 74, 64)),                                     # exactly nMaxColors colours
]


@pytest.mark.parametrize("K,mk", FEW_CASES)
def test_lab_few_colours_early_return(nq, oracle, K, mk):
    """NQ/PnnLABQuantizer.java:193-206: palette = the image's distinct colours in java.util.HashMap key order."""
    img = mk()
    seed = 3
    oq, want = _oracle_palette(oracle, 1, img, K)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(want, True, tile=(16, 16))
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(16, 16))
    out = gq.convert(K, True)
    assert len(out.palette) == len(want) and (out.palette == want).all()
    assert gq.params.distinctColors == oq.params.distinctColors
    assert (out.index.astype(np.int32) == want_idx).all() and (out.argb == want_argb).all()


def test_lab_no_dither_convert_computes_distinct_colours(nq, oracle):
    """convert(256, false): the BlueNoise weight needs pixelMap.size() (NQ/PnnLABQuantizer.java:512), counted on the GPU."""
    img = synth.gradient_noise(112, 96, 81)
    seed = 4
    oq, pal = _oracle_palette(oracle, 1, img, 256)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, False, tile=(16, 16))
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(16, 16))
    out = gq.convert(256, False)
    assert (out.palette == pal).all()
    assert gq.params.distinctColors == oq.params.distinctColors
    assert (out.index.astype(np.int32) == want_idx).all() and (out.argb == want_argb).all()


def _np_lab(argb):
    """CIELAB (D65) of ARGB pixels, float64 numpy -- for deltaE statistics only (the parity arithmetic lives in the oracle)."""
    u = argb.view(np.uint32)
    out = np.empty(argb.shape + (3,), np.float64)
    lin = np.array([(c / 255.0) / 12.92 if c / 255.0 < 0.04045 else ((c / 255.0 + 0.055) / 1.055) ** 2.4 for c in range(256)])
    r, g, b = lin[(u >> 16) & 0xFF], lin[(u >> 8) & 0xFF], lin[u & 0xFF]
    xyz = [(0.4124 * r + 0.3576 * g + 0.1805 * b) / 0.95047, 0.2126 * r + 0.7152 * g + 0.0722 * b, (0.0193 * r + 0.1192 * g + 0.9505 * b) / 1.08883]
    f = [np.where(t > 0.008856, np.cbrt(t), (903.3 * t + 16) / 116) for t in xyz]
    out[..., 0] = np.maximum(0, 116 * f[1] - 16); out[..., 1] = 500 * (f[0] - f[1]); out[..., 2] = 200 * (f[1] - f[2])
    return out


def _box8(a):
    h, w = a.shape[0] // 8 * 8, a.shape[1] // 8 * 8
    return a[:h, :w].reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))


TILED_VS_SEQ = [("gradient_noise 256^2", lambda: synth.gradient_noise(256, 256, 61), None, 4),       # name, image, seed, automatic tile
                ("uniform_rgb 224x160", lambda: synth.uniform_rgb(224, 160, 62), None, 4),
                ("few bins 192^2", lambda: synth.few_colors(192, 192, 63, 2000), None, 4),           # sorted-by-yDiff queue: tile chains start in its steady state
                ("bench image 4096^2", lambda: synth.gradient_noise(4096, 4096, 3), 3, 8)]


@pytest.mark.parametrize("name,mk,seed,tile", TILED_VS_SEQ)
def test_tiled_output_within_the_stated_deltaE_tolerance_of_the_sequential_reference(nq, oracle, name, mk, seed, tile):
    """The north star's "LAB-space dithered output within a stated per-pixel deltaE tolerance": PARALLEL_TILED (GPU, automatic
    tiles) against the SEQUENTIAL reference semantics (oracle, one curve over the image) on the same palette.  Two dithers of one
    image differ pixel by pixel by construction (each pixel takes one of two neighbouring palette colours), so the tolerance is
    stated per pixel against the SOURCE (CIE76: p50 / p99 of the tiled output no worse than the sequential output's by more than
    5 % + 0.1, p99.9 by more than 10 % + 0.5, the single worst pixel by more than 50 %), on the 8x8 local means between the two outputs (p99 <= 20 % of the sequential
    output's per-pixel p99, i.e. the patterns integrate to the same colours) and on the tile seams (mean Lab step across tile
    boundaries <= 1.05 x the step inside the tiles).  Measured (tests/tiled_vs_sequential.py): e.g. bench image p50/p99/max
    1.15/3.72/6.79 tiled vs 1.15/3.71/6.93 sequential, local means p99 0.31, seam ratio 1.008.
    And directly BETWEEN the two outputs, pixel by pixel (CIE76 of tiled[i] against sequential[i]): p99 <= 1.5 x the sequential output's
    p99 against the source + 0.5, worst pixel <= 2 x the sequential output's worst pixel against the source (two pixels that each sit
    within e of the source could be 2 e apart; measured 1.0-1.4 x / 1.1-1.8 x: noise 256^2 p99 19.5 max 32.9 against 18.8 / 31.0,
    bench image crop 6.9 / 15.7 against 5.0 / 10.4); 30-86 % of the pixels are identical.  The numbers are printed (pytest -s)."""
    img = mk()
    s = 9 if seed is None else seed
    H, W = img.shape
    oq, pal = _oracle_palette(oracle, 1, img, 256)
    params = _copy_params(oq.params, nq.Params)
    oq.set_seed(s)
    seq_argb, _ = oq.dither(pal, True)
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=s)
    gq.set_params(params)
    got_argb, _ = gq.dither(pal, True)
    src, ls, lt = _np_lab(img), _np_lab(seq_argb), _np_lab(got_argb)
    es, et = np.linalg.norm(ls - src, axis=2).ravel(), np.linalg.norm(lt - src, axis=2).ravel()
    ps, pt = [np.percentile(es, q) for q in (50, 99, 99.9, 100)], [np.percentile(et, q) for q in (50, 99, 99.9, 100)]
    assert pt[0] <= 1.05 * ps[0] + 0.1 and pt[1] <= 1.05 * ps[1] + 0.1 and pt[2] <= 1.10 * ps[2] + 0.5 and pt[3] <= 1.5 * ps[3], (name, ps, pt)
    dts = np.linalg.norm(lt - ls, axis=2).ravel()
    pd = [float(np.percentile(dts, q)) for q in (50, 99, 100)]
    print("%s: deltaE76 tiled vs sequential per pixel p50/p99/max %.2f/%.2f/%.2f (identical pixels %.1f %%); sequential vs source p50/p99/p99.9/max "
          "%.2f/%.2f/%.2f/%.2f; tiled vs source %.2f/%.2f/%.2f/%.2f" % ((name,) + tuple(pd) + (100.0 * float((got_argb == seq_argb).mean()),) + tuple(ps) + tuple(pt)))
    assert pd[1] <= 1.5 * ps[1] + 0.5 and pd[2] <= 2.0 * ps[3], (name, pd, ps)
    db = np.linalg.norm(_box8(ls) - _box8(lt), axis=2).ravel()
    assert np.percentile(db, 99) <= 0.2 * ps[1] + 0.05, (name, float(np.percentile(db, 99)), ps)
    d = np.linalg.norm(np.diff(lt, axis=1), axis=2)
    cols = np.arange(d.shape[1])
    seam = d[:, (cols % tile) == tile - 1].mean() / max(d[:, (cols % tile) == tile // 2 - 1].mean(), 1e-9)
    assert seam <= 1.05, (name, seam)


def test_full_size_properties_4096(nq):
    """BASELINE cfg 3 at full size through size-independent properties: every output pixel is the palette entry of its
    index, indices are in range, the palette has 256 distinct-bin survivors, and the run is reproducible."""
    import torch
    W = H = 4096
    img = synth.gradient_noise(W, H, 3)
    d_in = torch.from_numpy(img.reshape(-1)).cuda()
    d_out = torch.empty(W * H, dtype=torch.int32, device="cuda")
    d_idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=3)
    q.width, q.height = W, H
    pal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    assert len(pal) == 256
    idx = d_idx.cpu().numpy().view(np.uint16).astype(np.int64)
    out = d_out.cpu().numpy()
    assert idx.max() < 256
    assert (out == pal[idx]).all()
    assert len(np.unique(idx)) > 200
    first = out.copy()
    pal2 = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    assert (pal2 == pal).all() and (d_out.cpu().numpy() == first).all()
    # mean quantisation error in sRGB units stays small for a 256-colour palette on a smooth image
    def ch(a, s): return ((a.view(np.uint32) >> s) & 0xFF).astype(np.float64)
    err = np.mean([np.abs(ch(out, s) - ch(img.reshape(-1), s)).mean() for s in (0, 8, 16)])
    assert err < 12.0, err


def test_banded_pipeline_single_rank_equals_convert(nq, oracle, tmp_path):
    """The multi-GPU band pipeline (band scan -> reduce -> band histogram -> gather -> palette from histograms -> dither of the
    band, nquant.android_amd.parallel.convert_banded) with ONE rank and one band must equal the plain convert; the collectives
    run through torch.distributed (gloo here, RCCL on a multi-GPU node)."""
    import os
    import torch
    import torch.distributed as dist
    from nquant.android_amd import parallel
    img = synth.with_alpha(synth.gradient_noise(96, 80, 91), 91)
    seed = 6
    oq, pal = _oracle_palette(oracle, 1, img, 256)
    oq.set_seed(seed)
    want_argb, want_idx = oq.dither(pal, True, tile=(16, 16))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        H, W = img.shape
        d_band = torch.from_numpy(img.reshape(-1)).cuda()
        d_out = torch.empty(W * H, dtype=torch.int32, device="cuda")
        d_idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
        q = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(16, 16))
        got_pal = parallel.convert_banded(q, d_band, W, H, 0, 256, True, d_out, d_idx)
        torch.cuda.synchronize()
        assert (got_pal == pal).all()
        assert (d_idx.cpu().numpy().view(np.uint16).astype(np.int32).reshape(H, W) == want_idx).all()
        assert (d_out.cpu().numpy().reshape(H, W) == want_argb).all()
    finally:
        dist.destroy_process_group()


def test_two_band_rgb_palette_equals_whole_image(nq, oracle):
    """BASELINE cfg 5 semantics on one GPU: the image cut into two row bands, pre-scan partials reduced (last transparent pixel
    wins, counts add), per-band histograms combined in band order.  RGB sums are integers, so the palette must equal the
    oracle's whole-image palette bit for bit."""
    import ctypes as C
    import torch
    img = synth.with_alpha(synth.gradient_noise(96, 64, 93), 93)
    K = 64
    oq, want = _oracle_palette(oracle, 0, img, K)
    H, W = img.shape
    q = nq.PnnQuantizer(img)
    L = q._L
    bands = [(0, 37), (37, H)]
    d_img = torch.from_numpy(img.reshape(-1)).cuda()
    scans = []
    for (y0, y1) in bands:
        s3 = torch.empty(3, dtype=torch.int64, device="cuda")
        q._check(L.nq_band_scan_device(q._h, C.c_void_p(d_img[y0 * W:].data_ptr()), (y1 - y0) * W, y0 * W, K, C.c_void_p(s3.data_ptr())))
        torch.cuda.synchronize()
        scans.append(s3.cpu().numpy())
    win = max(scans, key=lambda s: s[0])
    q._check(L.nq_set_scan(q._h, K, int(win[0]), C.c_uint32(int(win[1]) & 0xFFFFFFFF), int(sum(s[2] for s in scans))))
    hists = torch.empty((2, 65536 * 5), dtype=torch.float64, device="cuda")
    for b, (y0, y1) in enumerate(bands):
        q._check(L.nq_band_histogram_device(q._h, C.c_void_p(d_img[y0 * W:].data_ptr()), (y1 - y0) * W, C.c_void_p(hists[b].data_ptr())))
    pal = np.zeros(K, np.int32)
    k = C.c_int32(0)
    q._check(L.nq_palette_from_histograms_device(q._h, C.c_void_p(hists.data_ptr()), 2, K, pal.ctypes.data, C.byref(k)))
    assert k.value == len(want) and (pal[:k.value] == want).all()
    op, gp = oq.params, q.params
    assert (op.transparentPixelIndex, op.transparentColor, op.hasSemiTransparency, op.maxbins) == \
           (gp.transparentPixelIndex, gp.transparentColor, gp.hasSemiTransparency, gp.maxbins)


def test_convert_batch_equals_single_converts(nq):
    """nq_convert_batch_device (all merge loops in one launch, shared scratch, one stream) == one nq_convert_device per image:
    mixed kinds, sizes, an image with alpha, a few-colours image (early return, no merge job)."""
    import torch
    imgs = [(1, synth.gradient_noise(144, 112, 81)), (0, synth.uniform_rgb(96, 80, 82)),
            (1, synth.with_alpha(synth.gradient_noise(80, 120, 83), 83)), (1, synth.few_colors(64, 64, 84, 100)),
            (0, synth.gradient_noise(128, 64, 85)), (1, synth.uniform_rgb(72, 72, 86))]
    cls = {0: nq.PnnQuantizer, 1: nq.PnnLABQuantizer}
    d_in = [torch.from_numpy(np.ascontiguousarray(im).reshape(-1)).cuda() for _, im in imgs]

    def fresh():
        qs = []
        for i, (kind, im) in enumerate(imgs):
            q = cls[kind](np.zeros((1, 1), np.int32), mode=TILED, seed=7 + i, tile=(16, 16))
            q.height, q.width = im.shape
            qs.append(q)
        return qs

    single = []
    for q, d in zip(fresh(), d_in):
        out = torch.empty_like(d)
        idx = torch.empty(d.numel(), dtype=torch.int16, device="cuda")
        pal = q.convert_device(d.data_ptr(), 256, True, out.data_ptr(), idx.data_ptr())
        single.append((pal, out.cpu().numpy(), idx.cpu().numpy()))
    qs = fresh()
    outs = [torch.empty_like(d) for d in d_in]
    idxs = [torch.empty(d.numel(), dtype=torch.int16, device="cuda") for d in d_in]
    pals = nq.convert_batch_device(qs, [d.data_ptr() for d in d_in], 256, True, [o.data_ptr() for o in outs],
                                   [x.data_ptr() for x in idxs])
    torch.cuda.synchronize()
    for i in range(len(imgs)):
        assert len(pals[i]) == len(single[i][0]) and (pals[i] != single[i][0]).sum() == 0, i
        assert (outs[i].cpu().numpy() != single[i][1]).sum() == 0, i
        assert (idxs[i].cpu().numpy() != single[i][2]).sum() == 0, i
        assert qs[i].stage_ms()["total"] > 0
    # the handles are usable one by one again afterwards (stream / scratch restored)
    again = torch.empty_like(d_in[0])
    pal0 = qs[0].convert_device(d_in[0].data_ptr(), 256, True, again.data_ptr())
    assert (pal0 != single[0][0]).sum() == 0 and (again.cpu().numpy() != single[0][1]).sum() == 0


_VARIANT_WANT = {}


@pytest.mark.parametrize("threads", [512, 256, 128])
def test_merge_loop_variants_agree_with_oracle(nq, oracle, threads, monkeypatch):
    """The merge loop is compiled for three workgroup sizes (csrc/nq_merge.inc; the batch size picks one): every variant must
    build the oracle's palette bit for bit -- LAB with > 16384 bins (more than the smallest LDS heap / mtm mirrors), LAB with
    alpha, RGB."""
    monkeypatch.setenv("NQ_MERGE_THREADS", str(threads))
    cases = [(1, synth.uniform_rgb(160, 160, 91), 256), (1, synth.with_alpha(synth.gradient_noise(96, 96, 92), 92), 64),
             (0, synth.gradient_noise(128, 128, 93), 256)]
    for ci, (kind, img, K) in enumerate(cases):
        if ci not in _VARIANT_WANT:                      # the oracle's answer does not depend on the variant
            _VARIANT_WANT[ci] = _oracle_palette(oracle, kind, img, K)[1]
        want = _VARIANT_WANT[ci]
        gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=TILED, seed=1)
        got = gq.pnnquan(K)
        assert len(got) == len(want) and (got != want).sum() == 0, (threads, kind)


def test_convert_batch_host_equals_device_batch(nq):
    """nq_convert_batch (host buffers, uploads / read-backs overlapped on a copy stream, ring of three device output buffers) ==
    nq_convert_batch_device, with page-locked and with pageable host memory; more images than ring slots."""
    import torch
    imgs = [(1, synth.gradient_noise(96 + 8 * i, 80, 120 + i)) for i in range(5)] + [(0, synth.uniform_rgb(64, 72, 130))]
    cls = {0: nq.PnnQuantizer, 1: nq.PnnLABQuantizer}

    def fresh():
        qs = []
        for i, (kind, im) in enumerate(imgs):
            q = cls[kind](np.zeros((1, 1), np.int32), mode=TILED, seed=11 + i, tile=(16, 16))
            q.height, q.width = im.shape
            qs.append(q)
        return qs

    d_in = [torch.from_numpy(np.ascontiguousarray(im).reshape(-1)).cuda() for _, im in imgs]
    d_out = [torch.empty_like(d) for d in d_in]
    d_idx = [torch.empty(d.numel(), dtype=torch.int16, device="cuda") for d in d_in]
    want_pal = nq.convert_batch_device(fresh(), [d.data_ptr() for d in d_in], 256, True, [o.data_ptr() for o in d_out],
                                       [x.data_ptr() for x in d_idx])
    torch.cuda.synchronize()
    for pinned in (True, False):
        h_in = [torch.from_numpy(np.ascontiguousarray(im).reshape(-1).copy()) for _, im in imgs]
        h_out = [torch.zeros(t.numel(), dtype=torch.int32) for t in h_in]
        h_idx = [torch.zeros(t.numel(), dtype=torch.int16) for t in h_in]
        if pinned:
            h_in = [t.pin_memory() for t in h_in]; h_out = [t.pin_memory() for t in h_out]; h_idx = [t.pin_memory() for t in h_idx]
        got_pal = nq.convert_batch_host(fresh(), [t.data_ptr() for t in h_in], 256, True, [t.data_ptr() for t in h_out],
                                        [t.data_ptr() for t in h_idx])
        for i in range(len(imgs)):
            assert (got_pal[i] != want_pal[i]).sum() == 0, (pinned, i)
            assert (h_out[i].numpy() != d_out[i].cpu().numpy()).sum() == 0, (pinned, i)
            assert (h_idx[i].numpy() != d_idx[i].cpu().numpy()).sum() == 0, (pinned, i)


@pytest.mark.parametrize("kind,K,dither,alpha", [(0, 2, True, False), (0, 2, False, True), (1, 2, True, True), (1, 2, False, False),
                                                 (1, 1, True, False), (0, 16, False, False), (1, 8, True, True), (0, 300, True, False),
                                                 # LAB, no dither, K > 32: the BlueNoise weight needs pixelMap.size() after the gilbert pass (:512)
                                                 (1, 256, False, False), (1, 64, False, True), (1, 100, False, False)])
def test_whole_convert_sequential_equals_oracle_convert(nq, oracle, kind, K, dither, alpha):
    """convert(n, dither) end to end in REFERENCE_SEQUENTIAL mode == the oracle's convert(), including nMaxColors <= 2 (fixed
    two-colour palette, alpha-0 pixels read as the transparent colour, NQ/PnnQuantizer.java:424,441-452) and RGB K > 256."""
    img = synth.gradient_noise(56, 44, 140 + K)
    if alpha:
        img = synth.with_alpha(img, 141 + K)
    seed = 9
    oq = oracle.OracleQuantizer(kind, img, seed=seed)
    want_argb, want_idx, want_pal = oq.convert(K, dither)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=SEQ, seed=seed)
    out = gq.convert(K, dither)
    assert len(out.palette) == len(want_pal) and (out.palette != want_pal).sum() == 0
    assert (out.argb != want_argb).sum() == 0
    assert (out.index.astype(np.int32) != want_idx).sum() == 0


def test_negative_ratio_is_handled_literally(nq, oracle):
    """NQ/PnnLABQuantizer.java:259-264 can make `ratio` NEGATIVE (5 colours out of ~690 bins: .036 - .0072 e^1.632): the CIEDE2000
    terms then lower the find_nn sums and the YUV terms lower the closest error, so neither the interval bounds, nor the candidate
    lists, nor the gate-free closest evaluation apply -- palette, closest tuples and the dithered image must still equal the
    oracle's (found by tests/fuzz_parity.py)."""
    img = synth.few_colors(112, 96, 37, 700)
    K, seed = 5, 4
    oq, want_pal = _oracle_palette(oracle, 1, img, K)
    assert oq.params.ratio < 0, oq.params.ratio
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(16, 16))
    pal = gq.pnnquan(K)
    assert gq.params.ratio == oq.params.ratio
    assert (pal != want_pal).sum() == 0
    cols = img.reshape(-1)[:6000]
    assert (gq.closestTuple(pal, cols) != oq.closest_tuple(want_pal, cols)).sum() == 0
    assert (gq.nearestColorIndex(pal, cols) != oq.nearest_index(want_pal, cols)).sum() == 0
    for dither in (True, False):
        oq.set_seed(seed)
        want_argb, want_idx = oq.dither(want_pal, dither, tile=(16, 16))
        got_argb, got_idx = gq.dither(pal, dither)
        assert (got_idx.astype(np.int32) != want_idx).sum() == 0 and (got_argb != want_argb).sum() == 0


def test_banded_pipeline_few_colours_early_return(nq, oracle):
    """Split pipeline, image with <= nMaxColors distinct colours: the bands' distinct colours are exchanged
    (nq_band_distinct_device -> parallel.merge_distinct -> nq_set_distinct) and the palette is the reference's early return
    (NQ/PnnLABQuantizer.java:193-206) -- one rank through convert_banded, and two bands driven by hand on one GPU."""
    import ctypes as C
    import os
    import torch
    import torch.distributed as dist
    from nquant.android_amd import parallel
    img = synth.with_alpha(synth.few_colors(64, 64, 72, 60), 72)
    H, W = img.shape
    K = 256
    oq, want = _oracle_palette(oracle, 1, img, K)
    assert len(want) <= K and oq.params.maxbins <= K        # the early-return case
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        d_band = torch.from_numpy(img.reshape(-1)).cuda()
        d_out = torch.empty(W * H, dtype=torch.int32, device="cuda")
        q = nq.PnnLABQuantizer(img, mode=TILED, seed=3, tile=(16, 16))
        got = parallel.convert_banded(q, d_band, W, H, 0, K, True, d_out)
        torch.cuda.synchronize()
        assert len(got) == len(want) and (got == want).all()
    finally:
        dist.destroy_process_group()
    # two bands by hand: per-band lists concatenated in band order, repeats dropped == the whole image's first-occurrence order
    q = nq.PnnLABQuantizer(img, mode=TILED, seed=3)
    L = q._L
    halves = [(0, H // 2), (H // 2, H)]
    d_bands = [torch.from_numpy(np.ascontiguousarray(img[a:b]).reshape(-1)).cuda() for a, b in halves]
    scans = []
    for (a, b), d in zip(halves, d_bands):
        s3 = torch.empty(3, dtype=torch.int64, device="cuda")
        q._check(L.nq_band_scan_device(q._h, C.c_void_p(d.data_ptr()), d.numel(), a * W, K, C.c_void_p(s3.data_ptr())))
        scans.append(s3.cpu())
    allv = torch.stack(scans)
    win = int(torch.argmax(allv[:, 0]))
    q._check(L.nq_set_scan(q._h, K, int(allv[win, 0]), C.c_uint32(int(allv[win, 1]) & 0xFFFFFFFF), int(allv[:, 2].sum())))
    hists, lists = [], []
    for d in d_bands:
        hst = torch.empty(65536 * 5, dtype=torch.float64, device="cuda")
        q._check(L.nq_band_histogram_device(q._h, C.c_void_p(d.data_ptr()), d.numel(), C.c_void_p(hst.data_ptr())))
        hists.append(hst)
        lists.append(parallel.band_distinct(q, d, d.numel(), K))
    merged, seen = [], set()
    for lst in lists:
        for c in lst:
            if c not in seen:
                seen.add(c); merged.append(c)
    cols = np.asarray(merged, np.int32)
    q._check(L.nq_set_distinct(q._h, len(cols), cols.ctypes.data))
    hs = torch.stack(hists)
    pal = np.zeros(K, np.int32)
    Kout = C.c_int32(0)
    q._check(L.nq_palette_from_histograms_device(q._h, C.c_void_p(hs.data_ptr()), 2, K, pal.ctypes.data, C.byref(Kout)))
    assert Kout.value == len(want) and (pal[:Kout.value] == want).all()
