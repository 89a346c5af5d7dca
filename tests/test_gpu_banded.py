"""One image tiled over GPUs (BASELINE cfg 5) and batches of frames (cfg 4), on the one GPU the test box has:

* the band pipeline by hand (one handle per band, the reductions done in the test exactly as nquant.android_amd.parallel does
  them) against the oracle's banded restatement (nqo_set_bands: float32 partial sums per band, added in band order) -- palette
  bit for bit, and the dithered bands against the oracle's WHOLE-image tiled dither (band origin = global tile indices, blue-noise
  phase and position gates), with 2 and 8 LAB bands, transparent pixels, dither on and off (image-wide distinct-colour count);
* parallel.convert_banded end to end in two real processes (gloo, both on GPU 0);
* cfg 4: a 1920x1080 frame through nq_convert_batch_device against the oracle, and the 64-frame batch through properties;
* cfg 5: 16384x16384 through properties, 8 bands by hand against the single-call dither of the same palette."""
import ctypes as C
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from nquant.android_amd import synth

pytestmark = pytest.mark.gpu

TILED = 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bands_by_hand(nq, img, kind, K, dither, row_starts, seed, tile=None):
    """The band pipeline with the reductions done here (what parallel.convert_banded does across ranks).  Returns palette, argb, index."""
    import torch
    H, W = img.shape
    Q = nq.PnnLABQuantizer if kind else nq.PnnQuantizer
    bounds = list(row_starts) + [H]
    nb = len(row_starts)
    qs = [Q(np.zeros((1, 1), np.int32), mode=TILED, seed=seed, tile=tile) for _ in range(nb)]
    L = qs[0]._L
    d_img = torch.from_numpy(img.reshape(-1).copy()).cuda()
    bands = [d_img[bounds[b] * W: bounds[b + 1] * W] for b in range(nb)]
    scans = []
    for b in range(nb):
        s3 = torch.empty(3, dtype=torch.int64, device="cuda")
        qs[b]._check(L.nq_band_scan_device(qs[b]._h, C.c_void_p(bands[b].data_ptr()), bands[b].numel(), bounds[b] * W, K, C.c_void_p(s3.data_ptr())))
        torch.cuda.synchronize()
        scans.append(s3.cpu().numpy())
    win = max(scans, key=lambda s: s[0])
    idx = int(win[0]); color = int(win[1]) if idx >= 0 else -1; semi = int(sum(s[2] for s in scans))
    hists = torch.zeros((nb, 65536 * 5), dtype=torch.float64, device="cuda")
    for b in range(nb):
        qs[b]._check(L.nq_set_scan(qs[b]._h, K, idx, C.c_uint32(color & 0xFFFFFFFF), semi))
        qs[b]._check(L.nq_band_histogram_device(qs[b]._h, C.c_void_p(bands[b].data_ptr()), bands[b].numel(), C.c_void_p(hists[b].data_ptr())))
    torch.cuda.synchronize()
    occupied = int((hists.view(nb, 65536, 5)[:, :, 0].sum(0) > 0).sum())
    merged = None
    if kind == 1 and occupied <= K:
        out, seen, many = [], set(), False
        for b in range(nb):
            cnt = C.c_int64(0); cols = np.zeros(K, np.int32)
            qs[b]._check(L.nq_band_distinct_device(qs[b]._h, C.c_void_p(bands[b].data_ptr()), bands[b].numel(), K, C.byref(cnt), cols.ctypes.data))
            if cnt.value > K:
                many = True
            for c in cols[:min(cnt.value, K)].tolist():
                if c not in seen:
                    seen.add(c); out.append(c)
        merged = None if (many or len(out) > K) else out
    pals = []
    for b in range(nb):
        if kind == 1 and occupied <= K:
            cols = np.asarray(merged if merged is not None else [], np.int32)
            qs[b]._check(L.nq_set_distinct(qs[b]._h, len(cols) if merged is not None else -1, cols.ctypes.data))
        pal = np.zeros(max(K, 2), np.int32); k = C.c_int32(0)
        qs[b]._check(L.nq_palette_from_histograms_device(qs[b]._h, C.c_void_p(hists.data_ptr()), nb, K, pal.ctypes.data, C.byref(k)))
        pals.append(pal[:k.value].copy())
    for b in range(1, nb):
        assert (pals[b] == pals[0]).all(), "ranks disagree on the palette"
    pal = pals[0]
    if kind == 1 and not dither and len(pal) > 32:
        presence = torch.zeros(1 << 24, dtype=torch.uint8, device="cuda")      # one table for all bands == byte-wise MAX of per-band tables
        others = set()
        for b in range(nb):
            cnt = C.c_int64(0); other = np.zeros(65536, np.uint32)
            qs[b]._check(L.nq_band_color_presence_device(qs[b]._h, C.c_void_p(bands[b].data_ptr()), bands[b].numel(),
                                                         C.c_void_p(presence.data_ptr()), 65536, C.byref(cnt), other.ctypes.data))
            assert cnt.value >= 0
            others.update(other[:cnt.value].tolist())
        total = int(torch.count_nonzero(presence)) + len(others)
        for b in range(nb):
            p = qs[b].params; p.distinctColors = total; qs[b].set_params(p)
    d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    for b in range(nb):
        rows = bounds[b + 1] - bounds[b]
        qs[b].width, qs[b].height = W, rows
        qs[b].set_band(bounds[b], H)
        qs[b].dither_device(bands[b].data_ptr(), pal, dither, d_out[bounds[b] * W:].data_ptr(), d_idx[bounds[b] * W:].data_ptr())
    torch.cuda.synchronize()
    return pal, d_out.cpu().numpy().reshape(H, W), d_idx.cpu().numpy().view(np.uint16).reshape(H, W), qs[0].params


BAND_CASES = [  # kind, K, dither, image, band starts, tile
    (1, 256, True, lambda: synth.gradient_noise(64, 128, 201), [0, 64], (8, 8)),
    (1, 256, True, lambda: synth.gradient_noise(48, 512, 202), [0, 64, 128, 192, 256, 320, 384, 448], (8, 8)),
    (1, 256, False, lambda: synth.gradient_noise(48, 512, 203), [0, 64, 128, 192, 256, 320, 384, 448], None),
    (1, 64, True, lambda: synth.with_alpha(synth.gradient_noise(64, 192, 204), 204), [0, 64, 128], (16, 16)),
    (1, 256, False, lambda: synth.with_alpha(synth.uniform_rgb(64, 144, 205), 205, p_semi=0.0), [0, 64, 128], (4, 4)),   # last band 16 rows
    (1, 256, True, lambda: synth.few_colors(64, 128, 206, 90), [0, 64], (8, 8)),           # few-colours early return across bands
    (0, 256, False, lambda: synth.gradient_noise(64, 128, 207), [0, 64], (16, 16)),        # RGB: closestColorIndex uses pos % 2
    (0, 64, True, lambda: synth.with_alpha(synth.gradient_noise(96, 128, 208), 208), [0, 64], (8, 8)),
]


@pytest.mark.parametrize("kind,K,dither,mk,starts,tile", BAND_CASES)
def test_bands_by_hand_equal_oracle_banded(nq, oracle, kind, K, dither, mk, starts, tile):
    img = mk()
    seed = 77
    H, W = img.shape
    oq = oracle.OracleQuantizer(kind, img, seed=seed)
    oq.set_bands(starts)
    oq.prescan(K)
    want_pal = oq.pnnquan(K)
    otile = tile
    if otile is None:
        otile = (4, 4)          # automatic rule at this size (every queue form: the sorted queue's tile chains start in its steady state)
    want_argb, want_idx = oq.dither(want_pal, dither, tile=otile)
    pal, argb, idx, gparams = _bands_by_hand(nq, img, kind, K, dither, starts, seed, tile)
    assert len(pal) == len(want_pal) and (pal == want_pal).all(), "banded palette differs from the oracle's banded restatement"
    if kind == 1 and not dither and len(pal) > 32:
        assert gparams.distinctColors == oq.params.distinctColors, "image-wide distinct-colour count"
    bad = int((idx.astype(np.int32) != want_idx).sum())
    assert bad == 0, "banded dither differs from the whole-image tiled dither in %d pixels" % bad
    assert (argb == want_argb).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _banded_worker(rank, world, port, tmp, dither, backend="gloo", give_height=True):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import nquant.android_amd as nq
    from nquant.android_amd import parallel
    torch.cuda.set_device(0)                   # all ranks share the one GPU of the box; the collectives run over gloo (or RCCL, world 1)
    dist.init_process_group(backend, rank=rank, world_size=world)
    data = np.load(os.path.join(tmp, "case.npz"))
    img = data["img"]
    H, W = img.shape
    y0, y1 = parallel.band_bounds(H, rank, world)
    d_band = torch.from_numpy(img[y0:y1].reshape(-1).copy()).cuda()
    d_out = torch.zeros(max((y1 - y0) * W, 1), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros(max((y1 - y0) * W, 1), dtype=torch.int16, device="cuda")
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=int(data["seed"]), tile=(8, 8))
    pal = parallel.convert_banded(q, d_band, W, y1 - y0, y0, 256, bool(dither), d_out, d_idx, image_height=H if give_height else None)
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, "out%d.npz" % rank), pal=pal, y0=y0, y1=y1, argb=d_out.cpu().numpy()[:(y1 - y0) * W],
             idx=d_idx.cpu().numpy().view(np.uint16)[:(y1 - y0) * W])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dither,world", [(True, 2), (False, 2), (True, 3)])
def test_convert_banded_two_processes_on_one_gpu(nq, oracle, tmp_path, dither, world):
    """parallel.convert_banded end to end with world_size ranks (separate processes, gloo collectives, all on GPU 0) == the oracle's
    banded restatement; world 3 on a 128-row image leaves the last rank an EMPTY band (it must still join the collectives)."""
    img = synth.gradient_noise(64, 128, 211)
    seed = 13
    H, W = img.shape
    np.savez(tmp_path / "case.npz", img=img, seed=seed)
    from nquant.android_amd import parallel
    spans = [parallel.band_bounds(H, r, world) for r in range(world)]
    starts = [y0 for (y0, y1) in spans if y1 > y0]
    oq = oracle.OracleQuantizer(1, img, seed=seed)
    oq.set_bands(starts)
    oq.prescan(256)
    want_pal = oq.pnnquan(256)
    want_argb, want_idx = oq.dither(want_pal, dither, tile=(8, 8))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_banded_worker, args=(r, world, port, str(tmp_path), dither)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in range(world):
        o = np.load(tmp_path / ("out%d.npz" % r))
        assert (o["pal"] == want_pal).all()
        y0, y1 = int(o["y0"]), int(o["y1"])
        assert (y0, y1) == spans[r]
        if y1 > y0:
            assert (o["idx"].reshape(y1 - y0, W).astype(np.int32) == want_idx[y0:y1]).all(), "rank %d" % r
            assert (o["argb"].reshape(y1 - y0, W) == want_argb[y0:y1]).all()


RCCL_CASES = [  # name, image, dither
    ("lab_no_dither_k256", lambda: synth.gradient_noise(64, 128, 221), False),      # image-wide distinct-colour count: presence-table all-reduce
    ("few_colours", lambda: synth.few_colors(64, 128, 222, 90), True),               # few-colours early return: exchange of the distinct colours
    ("lab_dither_k256", lambda: synth.gradient_noise(64, 128, 223), True),
]


@pytest.mark.parametrize("name,mk,dither", RCCL_CASES)
def test_convert_banded_under_rccl_world_1(nq, oracle, tmp_path, name, mk, dither):
    """parallel.convert_banded with the collectives over RCCL ("nccl", a world of ONE rank on the one GPU of the box -- a one-rank RCCL
    group already refuses host tensors, which is what the branches below used to hand it) and WITHOUT image_height (the band rows are
    gathered): LAB dither=false K=256 (BlueNoise weight from the image-wide distinct-colour count), the few-colours early return
    (NQ/PnnLABQuantizer.java:193-206, :511-515) and the plain dithered case, each == the oracle's banded restatement with one band."""
    img = mk()
    seed = 29
    H, W = img.shape
    np.savez(tmp_path / "case.npz", img=img, seed=seed)
    oq = oracle.OracleQuantizer(1, img, seed=seed)
    oq.set_bands([0])
    oq.prescan(256)
    want_pal = oq.pnnquan(256)
    want_argb, want_idx = oq.dither(want_pal, dither, tile=(8, 8))
    ctx = mp.get_context("spawn")
    pr = ctx.Process(target=_banded_worker, args=(0, 1, _free_port(), str(tmp_path), dither, "nccl", False))
    pr.start()
    pr.join(300)
    assert pr.exitcode == 0, "convert_banded failed under the nccl backend (exit code %r)" % pr.exitcode
    o = np.load(tmp_path / "out0.npz")
    assert len(o["pal"]) == len(want_pal) and (o["pal"] == want_pal).all()
    assert (o["idx"].reshape(H, W).astype(np.int32) == want_idx).all()
    assert (o["argb"].reshape(H, W) == want_argb).all()


# ---- BASELINE cfg 4: batch of 1920x1080 frames -------------------------------------------------------------------------------
def test_cfg4_one_1080p_frame_vs_oracle_and_batch_properties(nq, oracle):
    import torch
    W, H = 1920, 1080
    seed0 = 100
    frames = 64
    img0 = synth.gradient_noise(W, H, seed0)
    oq = oracle.OracleQuantizer(1, img0, seed=seed0)
    oq.prescan(256)
    want_pal = oq.pnnquan(256)
    want_argb, want_idx = oq.dither(want_pal, True, tile=(4, 4))         # the automatic rule at 1920x1080: 4x4 (129 600 tiles)
    qs, ins, outs, idxs = [], [], [], []
    for f in range(frames):
        d_in = torch.from_numpy(img0.reshape(-1)).cuda() if f == 0 else synth.gradient_noise_torch(W, H, seed0 + f)
        q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed0 + f)
        q.width, q.height = W, H
        qs.append(q); ins.append(d_in)
        outs.append(torch.zeros(W * H, dtype=torch.int32, device="cuda")); idxs.append(torch.zeros(W * H, dtype=torch.int16, device="cuda"))
    pals = nq.convert_batch_device(qs, [t.data_ptr() for t in ins], 256, True, [t.data_ptr() for t in outs], [t.data_ptr() for t in idxs])
    torch.cuda.synchronize()
    assert (pals[0] == want_pal).all(), "frame 0: palette differs from the oracle"
    got_idx = idxs[0].cpu().numpy().view(np.uint16).reshape(H, W).astype(np.int32)
    assert (got_idx == want_idx).all(), "frame 0: %d index mismatches" % int((got_idx != want_idx).sum())
    assert (outs[0].cpu().numpy().reshape(H, W) == want_argb).all()
    for f in range(frames):
        pal = torch.from_numpy(pals[f]).cuda()
        assert len(pals[f]) == 256
        ix = idxs[f].to(torch.int64) & 0xFFFF
        assert int(ix.max()) < 256
        assert bool((pal[ix] == outs[f]).all()), "frame %d: output pixel != palette[index]" % f
    assert not (pals[1] == pals[2]).all()
    # one frame again, alone: the batch result of that frame
    f = 17
    o2 = torch.zeros(W * H, dtype=torch.int32, device="cuda"); i2 = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    p2 = qs[f].convert_device(ins[f].data_ptr(), 256, True, o2.data_ptr(), i2.data_ptr())
    torch.cuda.synchronize()
    assert (p2 == pals[f]).all() and bool((o2 == outs[f]).all()) and bool((i2 == idxs[f]).all())


# ---- BASELINE cfg 5: 16384 x 16384 ------------------------------------------------------------------------------------------
def test_cfg5_16384_properties_and_eight_bands(nq):
    """2^28 pixels: (1) one image on one GPU -- every output pixel is palette[index], reproducible; (2) the band pipeline with 8
    bands of 2048 rows (bands_by_hand's steps on device data) -- same palette on every band handle, and the dithered bands equal
    the single-call dither of the whole image with that palette (global tile indices / blue-noise phase at 2^28-pixel indexing)."""
    import torch
    W = H = 16384
    seed = 5
    d_in = synth.gradient_noise_torch(W, H, seed)
    torch.cuda.synchronize()
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed)
    q.width, q.height = W, H
    d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    pal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    assert len(pal) == 256
    palt = torch.from_numpy(pal).cuda()
    ix = d_idx.to(torch.int64) & 0xFFFF
    assert int(ix.max()) < 256
    assert bool((palt[ix] == d_out).all())
    del ix
    first = d_out.clone()
    pal2 = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    assert (pal2 == pal).all() and bool((first == d_out).all())
    del first
    # 8 bands by hand
    L = q._L
    nb = 8
    rows = H // nb
    qs = [nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed) for _ in range(nb)]
    bands = [d_in[b * rows * W:(b + 1) * rows * W] for b in range(nb)]
    hists = torch.zeros((nb, 65536 * 5), dtype=torch.float64, device="cuda")
    for b in range(nb):
        s3 = torch.empty(3, dtype=torch.int64, device="cuda")
        qs[b]._check(L.nq_band_scan_device(qs[b]._h, C.c_void_p(bands[b].data_ptr()), rows * W, b * rows * W, 256, C.c_void_p(s3.data_ptr())))
        torch.cuda.synchronize()
        assert s3.cpu().tolist() == [-1, -1, 0]              # opaque image
        qs[b]._check(L.nq_set_scan(qs[b]._h, 256, -1, C.c_uint32(0xFFFFFFFF), 0))
        qs[b]._check(L.nq_band_histogram_device(qs[b]._h, C.c_void_p(bands[b].data_ptr()), rows * W, C.c_void_p(hists[b].data_ptr())))
    torch.cuda.synchronize()
    bpal = None
    for b in range(nb):
        p = np.zeros(256, np.int32); k = C.c_int32(0)
        qs[b]._check(L.nq_palette_from_histograms_device(qs[b]._h, C.c_void_p(hists.data_ptr()), nb, 256, p.ctypes.data, C.byref(k)))
        assert k.value == 256
        if bpal is None:
            bpal = p.copy()
        assert (p == bpal).all()
    # pixel counts of the gathered histograms add up to the image
    assert int(hists.view(nb, 65536, 5)[:, :, 0].sum().item()) == W * H
    b_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    b_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    for b in range(nb):
        qs[b].width, qs[b].height = W, rows
        qs[b].set_band(b * rows, H)
        qs[b].dither_device(bands[b].data_ptr(), bpal, True, b_out[b * rows * W:].data_ptr(), b_idx[b * rows * W:].data_ptr())
    torch.cuda.synchronize()
    q.dither_device(d_in.data_ptr(), bpal, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    assert bool((b_idx == d_idx).all()) and bool((b_out == d_out).all()), "8 dithered bands != the whole image dithered with the same palette"


def _torch_checksum(t):
    """make_golden.image_checksum on a device tensor (int64 arithmetic wraps like the uint64 sums of the fixture)."""
    import torch
    v = t.to(torch.int64) & 0xFFFFFFFF
    w = (torch.arange(v.numel(), dtype=torch.int64, device=t.device) & 0xFFFF) + 1
    s0, s1 = int(v.sum().item()), int((v * w).sum().item())
    return np.array([s0 & 0xFFFFFFFFFFFFFFFF, s1 & 0xFFFFFFFFFFFFFFFF], np.uint64)


def test_cfg5_16384_palette_and_tile_rows_vs_oracle(nq, oracle):
    """BASELINE cfg 5 at FULL size against the oracle, not against the GPU itself: the palette and scalars of the 2^28-pixel image
    equal the committed fixture (tests/golden/cfg5_lab256_palette_16384x16384.npz: the oracle's whole-image pnnquan, produced in the build
    container by make_golden.py --big -- minutes of CPU time), and three tile rows of the dithered output (first, middle, last:
    3 x 2048 tiles of 8x8) equal the oracle's tiled restatement run here (nqo_dither_tile_rows with the fixture's palette and scalars).
    The input is identified by a checksum held in the fixture; if the device-generated image differs from it in a single pixel (a
    device sin() on a rounding boundary) the image is regenerated with numpy."""
    import torch
    want = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_lab256_palette_16384x16384.npz"))
    W = H = 16384
    seed = 5
    d_in = synth.gradient_noise_torch(W, H, seed)
    torch.cuda.synchronize()
    if not (_torch_checksum(d_in) == want["checksum"]).all():
        img = synth.gradient_noise_banded(W, H, seed)
        d_in = torch.from_numpy(img.reshape(-1)).cuda()
        assert (_torch_checksum(d_in) == want["checksum"]).all(), "cannot reproduce the fixture's input image"
    else:
        img = d_in.cpu().numpy().reshape(H, W)
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed)
    q.width, q.height = W, H
    d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    pal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    assert len(pal) == len(want["palette"]) and (pal == want["palette"]).all(), "%d palette entries differ from the oracle's" % int((pal != want["palette"]).sum())
    p = q.params
    assert [p.hasSemiTransparency, p.transparentPixelIndex, p.transparentColor, p.isNano, p.texicab, p.quan_rt, p.maxbins,
            p.paletteLength] == list(want["scalars"])
    assert (np.array([p.PR, p.PG, p.PB, p.PA, p.ratio, p.weight]) == want["doubles"]).all()
    del d_in
    # three tile rows against the oracle (it holds its own copy of the image; prescan sets PR..PA and the transparency fields)
    oq = oracle.OracleQuantizer(1, img, seed=seed)
    oq.prescan(256)
    op = oq.params
    for f, v in zip(("hasSemiTransparency", "transparentPixelIndex", "transparentColor", "isNano", "texicab", "quan_rt", "maxbins", "paletteLength"),
                    want["scalars"].tolist()):
        setattr(op, f, int(v))
    op.nMaxColors = 256
    op.PR, op.PG, op.PB, op.PA, op.ratio, op.weight = [float(v) for v in want["doubles"]]
    op.distinctColors = int(want["distinct"])
    oq.set_params(op)
    oq.set_seed(seed)
    for r in (0, 1027, 2047):
        ys = slice(r * 8, r * 8 + 8)
        want_argb, want_idx = oq.dither_tile_rows(pal, True, (8, 8), r, 1)
        got_idx = d_idx[r * 8 * W:(r * 8 + 8) * W].cpu().numpy().view(np.uint16).reshape(8, W).astype(np.int32)
        got_argb = d_out[r * 8 * W:(r * 8 + 8) * W].cpu().numpy().reshape(8, W)
        assert (got_idx == want_idx[ys]).all(), "tile row %d: %d indices differ from the oracle" % (r, int((got_idx != want_idx[ys]).sum()))
        assert (got_argb == want_argb[ys]).all()
        del want_argb, want_idx
