"""world_size-2 gloo test of the multi-GPU plumbing (no GPU needed): frame sharding, band split, the pre-scan reduction and
the band-ordered histogram gather; the band partials are produced with numpy from the same bin function the oracle uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _numpy_band(img_band, index_offset):
    """scan3 and the RGB partial histogram of one band, computed independently with numpy."""
    u = img_band.reshape(-1).view(np.uint32)
    a = (u >> 24).astype(np.int64)
    zero = np.nonzero(a == 0)[0]
    idx = int(zero.max() + index_offset) if zero.size else -1
    color = int(u[zero.max()]) if zero.size else -1
    semi = int(((a > 0xF) & (a < 0xE0)).sum())
    return np.array([idx, color, semi], np.int64)


def _hist_rgb(img, transparent_color, has_semi, has_transp):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    L = oracle_lib.lib()
    u = img.reshape(-1).view(np.uint32).copy()
    a = u >> 24
    u[a <= 0xF] = np.uint32(transparent_color & 0xFFFFFFFF)
    uniq, inv = np.unique(u, return_inverse=True)
    bins_u = np.array([L.nqo_get_color_index(int(np.int32(np.uint32(c).view(np.int32))), has_semi, has_transp) for c in uniq])
    bins = bins_u[inv]
    h = np.zeros((65536, 5), np.float64)
    np.add.at(h[:, 0], bins, 1)
    for k, sh in enumerate((24, 16, 8, 0)):
        np.add.at(h[:, 1 + k], bins, ((u >> sh) & 0xFF).astype(np.float64))
    return h


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nquant.android_amd import parallel, synth
    assert parallel.shard_frames(7, rank, world) == list(range(rank, 7, world))
    H, W = 37, 24
    img = synth.with_alpha(synth.uniform_rgb(W, H, 9), 9)
    y0, y1 = parallel.band_bounds(H, rank, world, align=8)      # [0, 24) and [24, 37)
    assert y1 > y0
    band = img[y0:y1]
    scan3 = torch.from_numpy(_numpy_band(band, y0 * W))
    idx, color, semi = parallel.reduce_scan(scan3)
    whole = _numpy_band(img, 0)
    assert (idx, color & 0xFFFFFFFF, semi) == (int(whole[0]), int(whole[1]) & 0xFFFFFFFF, int(whole[2]))
    hist = torch.from_numpy(_hist_rgb(band, color, semi > 0, True).reshape(-1))
    hists = parallel.gather_histograms(hist)
    assert hists.shape == (world, 65536 * 5)
    total = hists.sum(0).numpy().reshape(65536, 5)
    want = _hist_rgb(img, color, semi > 0, True)
    assert (total == want).all()                       # integer sums: exact in any order
    assert (hists[rank].numpy() == hist.numpy()).all()  # band order == rank order
    t = parallel.max_over_ranks(1.0 + rank)
    assert t == float(world)
    # distinct-colour exchange of the few-colours early return: band lists in rank order, repeats dropped, None = "too many"
    few = synth.few_colors(W, H, 11, 9)
    def first_seen(a):
        out, seen = [], set()
        for c in a.reshape(-1).tolist():
            if c not in seen:
                seen.add(c); out.append(c)
        return out
    mine = first_seen(few[y0:y1])
    assert parallel.merge_distinct(mine, 16) == first_seen(few)
    assert parallel.merge_distinct(mine, 4) is None                        # more than cap colours overall
    assert parallel.merge_distinct(None if rank == 1 else mine, 16) is None  # one band reported "too many"
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")


def test_two_rank_gloo(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_band_bounds_cover_image():
    sys.path.insert(0, ROOT)
    from nquant.android_amd import parallel
    for h in (1, 7, 8, 1080, 16384):
        for world in (1, 2, 3, 8):
            spans = [parallel.band_bounds(h, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == h
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            # bands start on multiples of 64 rows (tile heights 16 / 8 / 4 and the blue-noise period divide it); with fewer
            # 64-row blocks than ranks the last ranks get an empty band instead of a misaligned one
            assert all(y0 % 64 == 0 or y0 == h for y0, _ in spans)
            assert all(y1 >= y0 for y0, y1 in spans)
    assert parallel.band_bounds(37, 1, 2) == (37, 37)
    assert parallel.band_bounds(16384, 3, 8) == (6144, 8192)
