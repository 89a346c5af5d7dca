"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  The path shards two ways:

* independent images (BASELINE cfg 4): frame f -> rank f mod world, no data-path collective;
* one image cut into row bands (cfg 5): ONE exchange step -- the pre-scan scalars (max / sum) and the 65536x5 f64
  histogram partials (2.6 MB per rank, all-gather so that every rank adds them in band order) -- then every rank builds
  the same palette and dithers its own band.

Only tensors cross this module; the quantizer calls go through the C ABI (host.py)."""
import numpy as np
import torch
import torch.distributed as dist

HIST_BINS, HIST_STRIDE = 65536, 5


def shard_frames(n_frames, rank, world):
    """Frames of a batch owned by `rank` (round robin)."""
    return list(range(rank, n_frames, world))


BAND_ALIGN = 64      # rows: a multiple of every tile height (16, 8, 4) and of the 64-row blue-noise period


def band_bounds(height, rank, world, align=BAND_ALIGN):
    """Row band [y0, y1) of `rank`: the image is cut at multiples of `align` rows (a band must start on a tile boundary for its
    dither to equal the same rows of the single-GPU result); the blocks are dealt out evenly, the first bands take the remainder,
    the last block may be short.  With fewer blocks than ranks the last ranks get an EMPTY band (y0 == y1): they still take part in
    every collective of convert_banded."""
    blocks = (height + align - 1) // align
    base, rem = divmod(blocks, world)
    b0 = rank * base + min(rank, rem)
    b1 = b0 + base + (1 if rank < rem else 0)
    return min(b0 * align, height), min(b1 * align, height)


def reduce_scan(scan3, group=None):
    """scan3 = int64[3] {max global index of an alpha==0 pixel or -1, its colour, count of semi-transparent pixels} of this
    band.  Returns the image-wide triple: the LAST transparent pixel wins (NQ/PnnQuantizer.java:419-422), counts add."""
    allv = _gather_small(scan3, group)
    winner = int(torch.argmax(allv[:, 0]))
    idx = int(allv[winner, 0])
    color = int(allv[winner, 1]) if idx >= 0 else -1
    semi = int(allv[:, 2].sum())
    return idx, color, semi


def gather_histograms(hist, group=None):
    """hist = f64[65536*5] partial of this band -> f64[world, 65536*5] in band (= rank) order on every rank."""
    world = dist.get_world_size(group)
    src = hist.contiguous().to(_collective_device(group))
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return torch.stack(parts).to(hist.device)


def _collective_device(group=None):
    """Device the tensors of a collective must live on: RCCL ("nccl") only moves device memory -- a CPU tensor handed to it raises
    "No backend type associated with device type cpu" --, gloo (CPU tests) only host memory."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _gather_small(t, group=None):
    """All-gather of a small tensor (any device) -> CPU tensor [world, ...], staged on the device the backend needs."""
    world = dist.get_world_size(group)
    src = t.to(_collective_device(group)).contiguous()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return torch.stack(parts).cpu()


def band_distinct(q, d_band, n, cap):
    """Distinct colours of this band (first-occurrence order) or None when there are more than `cap`."""
    import ctypes as C
    cnt = C.c_int64(0)
    cols = np.zeros(cap, np.int32)
    q._check(q._L.nq_band_distinct_device(q._h, C.c_void_p(d_band.data_ptr()), n, cap, C.byref(cnt), cols.ctypes.data))
    return None if cnt.value > cap else [int(c) for c in cols[:cnt.value]]


def merge_distinct(mine, cap, group=None):
    """mine: this band's distinct colours in first-occurrence order, or None (= more than cap).  Returns the image-wide list in
    first-occurrence order (bands in rank order, repeats dropped), or None when it is longer than cap.  Fixed-size exchange:
    int64[cap + 1] = {count or -1, colours...}."""
    t = torch.full((cap + 1,), -1, dtype=torch.int64)
    if mine is not None and len(mine) <= cap:
        t[0] = len(mine)
        if mine:
            t[1:1 + len(mine)] = torch.tensor(mine, dtype=torch.int64)
    allv = _gather_small(t, group)
    out, seen = [], set()
    for r in range(allv.shape[0]):
        k = int(allv[r, 0])
        if k < 0:
            return None
        for c in allv[r, 1:1 + k].tolist():
            if c not in seen:
                seen.add(c)
                out.append(int(c))
    return out if len(out) <= cap else None


def max_over_ranks(seconds, device=None, group=None):
    t = torch.tensor([seconds], dtype=torch.float64, device=_collective_device(group) if device is None else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def image_distinct_count(q, d_band, n, device, group=None, cap_other=65536):
    """Distinct colours of the WHOLE image (as the histogram sees them) from the bands of all ranks: byte-wise MAX all-reduce of
    the 2^24-byte opaque-colour tables (16 MB) + union of the short non-opaque lists.  Returns None when some band has more than
    cap_other non-opaque colours (the caller then keeps the band-local count: stated deviation for such images)."""
    import ctypes as C
    presence = torch.zeros(1 << 24, dtype=torch.uint8, device=device)
    other = np.zeros(cap_other, np.uint32)
    cnt = C.c_int64(0)
    if n > 0:
        q._check(q._L.nq_band_color_presence_device(q._h, C.c_void_p(d_band.data_ptr()), n, C.c_void_p(presence.data_ptr()), cap_other,
                                                    C.byref(cnt), other.ctypes.data))
    red = presence.to(_collective_device(group))
    dist.all_reduce(red, op=dist.ReduceOp.MAX, group=group)
    opaque = int(torch.count_nonzero(red))
    t = torch.full((cap_other + 1,), -1, dtype=torch.int64)
    k = int(cnt.value)
    t[0] = k
    if k > 0:
        t[1:1 + k] = torch.from_numpy(other[:k].astype(np.int64))
    allv = _gather_small(t, group)
    if bool((allv[:, 0] < 0).any()):
        return None
    others = set()
    for r in range(allv.shape[0]):
        others.update(allv[r, 1:1 + int(allv[r, 0])].tolist())
    return opaque + len(others)


def convert_banded(q, d_band, width, band_rows, y0, nMaxColors, dither, d_out_argb, d_out_index=None, group=None, image_height=None,
                   timings=None):
    """One image tiled over the ranks of `group` (GPU only).  q: a PnnQuantizer/PnnLABQuantizer of this rank; d_band: this
    rank's rows [y0, y0 + band_rows) as a CUDA int32 tensor (band_rows may be 0: the rank then only takes part in the collectives);
    image_height: rows of the whole image (default: the sum of the bands' rows).  Bands must start on multiples of the tile height
    (band_bounds cuts at multiples of 64).  Returns the (shared) palette; the band's pixels equal rows [y0, y0 + band_rows) of the
    single-GPU PARALLEL_TILED convert of the whole image given the same palette.  timings (optional dict): seconds spent in the
    collectives / the palette build / the band dither are added to its "collectives" / "palette" / "dither" entries (bench.py)."""
    import ctypes as C
    import time as _time
    L = q._L

    def _tick(key, t0):
        if timings is not None:
            torch.cuda.synchronize()
            timings[key] = timings.get(key, 0.0) + (_time.perf_counter() - t0)
    
    n = width * band_rows
    dev = d_out_argb.device if band_rows == 0 else d_band.device
    if image_height is None:
        t = torch.tensor([band_rows], dtype=torch.int64)
        image_height = int(_gather_small(t, group).sum())
    scan3 = torch.tensor([-1, -1, 0], dtype=torch.int64, device=dev)
    if n > 0:
        q._check(L.nq_band_scan_device(q._h, C.c_void_p(d_band.data_ptr()), n, y0 * width, nMaxColors, C.c_void_p(scan3.data_ptr())))
    if timings is not None:
        torch.cuda.synchronize()
    t0 = _time.perf_counter()
    idx, color, semi = reduce_scan(scan3, group)
    _tick("collectives", t0)
    q._check(L.nq_set_scan(q._h, nMaxColors, idx, C.c_uint32(color & 0xFFFFFFFF), semi))
    hist = torch.zeros(HIST_BINS * HIST_STRIDE, dtype=torch.float64, device=dev)
    if n > 0:
        q._check(L.nq_band_histogram_device(q._h, C.c_void_p(d_band.data_ptr()), n, C.c_void_p(hist.data_ptr())))
    if timings is not None:
        torch.cuda.synchronize()
    t0 = _time.perf_counter()
    hists = gather_histograms(hist, group)
    _tick("collectives", t0)
    t0 = _time.perf_counter()
    if q.KIND == 1 and int((hists.view(hists.shape[0], HIST_BINS, HIST_STRIDE)[:, :, 0].sum(0) > 0).sum()) <= nMaxColors:
        # NQ/PnnLABQuantizer.java:193-206: with so few occupied bins the image may hold <= nMaxColors distinct colours, and the
        # reference then returns them as they are -- every rank takes this branch together (the gathered histograms are identical)
        mine = band_distinct(q, d_band, n, nMaxColors) if n > 0 else []
        merged = merge_distinct(mine, nMaxColors, group)
        cols = np.asarray(merged if merged is not None else [], np.int32)
        q._check(L.nq_set_distinct(q._h, len(cols) if merged is not None else -1, cols.ctypes.data))
    pal = np.zeros(max(nMaxColors, 2), np.int32)
    K = C.c_int32(0)
    q._check(L.nq_palette_from_histograms_device(q._h, C.c_void_p(hists.data_ptr()), hists.shape[0], nMaxColors,
                                                 pal.ctypes.data, C.byref(K)))
    pal = pal[:K.value].copy()
    _tick("palette", t0)
    t0 = _time.perf_counter()
    if q.KIND == 1 and not dither and K.value > 32:
        # BlueNoise weight of convert(n, false): delta = K^2 / pixelMap.size() over the WHOLE image (NQ/PnnLABQuantizer.java:511-515)
        total = image_distinct_count(q, d_band, n, dev, group)
        if total is not None:
            p = q.params
            p.distinctColors = total
            q.set_params(p)
    if band_rows > 0:
        q.width, q.height = width, band_rows
        q.set_band(y0, image_height)
        try:
            q.dither_device(d_band.data_ptr(), pal, dither, d_out_argb.data_ptr(), d_out_index.data_ptr() if d_out_index is not None else 0)
        finally:
            q.set_band(0, 0)
    _tick("dither", t0)
    return pal
