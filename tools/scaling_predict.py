"""What a 1 / 2 / 4 / 8-GPU run of the three bench configurations would take, measured on ONE GPU by giving it exactly the share ONE rank
of an N-GPU run has (the ranks of all three configurations run independently except for cfg 5's two small all-gathers, whose time is
taken from a measured world-1 RCCL run and scaled by the message count).  The driver's SCALE run can be held against these numbers.

  cfg 3  weak scaling: every rank converts its own batch of 4096^2 images -> N x the single-GPU rate (no data-path collective);
  cfg 4  strong: 64 frames of 1920x1080, frame f on rank f mod N -> one rank converts 64 / N frames in one batch call;
  cfg 5  strong: one 16384^2 image in N row bands -> one rank scans + histograms its band, adds the N gathered partials, builds the
         palette (replicated), dithers its band.  The N - 1 other bands' partial histograms are computed here as well (untimed) so that
         the palette build sees the whole image's histogram.

Usage: python tools/scaling_predict.py [cfg4] [cfg5]   (default both; cfg 3 needs no measurement beyond bench.py's own line)"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import nquant.android_amd as nq
from nquant.android_amd import synth, parallel

TILED = 1
out = {}


def cfg4():
    W, H, frames = 1920, 1080, 64
    npx = W * H
    res = {}
    for N in (1, 2, 4, 8):
        mine = parallel.shard_frames(frames, 0, N)
        qs, ins, outs, idxs = [], [], [], []
        for f in mine:
            q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=100 + f)
            q.width, q.height = W, H
            qs.append(q); ins.append(synth.gradient_noise_torch(W, H, 100 + f))
            outs.append(torch.empty(npx, dtype=torch.int32, device="cuda")); idxs.append(torch.empty(npx, dtype=torch.int16, device="cuda"))
        best = None
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nq.convert_batch_device(qs, [t.data_ptr() for t in ins], 256, True, [t.data_ptr() for t in outs], [t.data_ptr() for t in idxs])
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if it:
                best = dt if best is None else min(best, dt)
        ts = qs[0].team_stats()
        res[N] = {"frames_per_rank": len(mine), "seconds_per_batch": round(best, 4), "frames_per_s_whole_job": round(frames / best, 1),
                  "mpixels_s_whole_job": round(frames * npx / best / 1e6, 1), "helpers_per_merge_loop": int(ts["helpers"]),
                  "phases_ms": {k: round(v, 2) for k, v in qs[0].batch_phase_ms().items()}}
        print("cfg4 N=%d: %s" % (N, res[N]), flush=True)
        del qs, ins, outs, idxs
    return res


def cfg5():
    W = H = 16384
    seed = 5
    res = {}
    L = nq.load_library()
    for N in (1, 2, 4, 8):
        bounds = [parallel.band_bounds(H, r, N) for r in range(N)]
        y0, y1 = bounds[0]
        rows = y1 - y0
        q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed)
        hists = torch.zeros((N, 65536 * 5), dtype=torch.float64, device="cuda")
        # the other ranks' partial histograms (untimed: they run on other GPUs at the same time)
        for r in range(1, N):
            b0, b1 = bounds[r]
            d = synth.gradient_noise_torch(W, H, seed, row0=b0, rows=b1 - b0)
            qq = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=TILED, seed=seed)
            s3 = torch.empty(3, dtype=torch.int64, device="cuda")
            qq._check(L.nq_band_scan_device(qq._h, C.c_void_p(d.data_ptr()), d.numel(), b0 * W, 256, C.c_void_p(s3.data_ptr())))
            qq._check(L.nq_set_scan(qq._h, 256, -1, C.c_uint32(0xFFFFFFFF), 0))
            qq._check(L.nq_band_histogram_device(qq._h, C.c_void_p(d.data_ptr()), d.numel(), C.c_void_p(hists[r].data_ptr())))
            torch.cuda.synchronize()
            del d, qq
        d_band = synth.gradient_noise_torch(W, H, seed, row0=y0, rows=rows)
        d_out = torch.empty(rows * W, dtype=torch.int32, device="cuda"); d_idx = torch.empty(rows * W, dtype=torch.int16, device="cuda")
        best = None
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s3 = torch.empty(3, dtype=torch.int64, device="cuda")
            q._check(L.nq_band_scan_device(q._h, C.c_void_p(d_band.data_ptr()), d_band.numel(), y0 * W, 256, C.c_void_p(s3.data_ptr())))
            torch.cuda.synchronize()
            q._check(L.nq_set_scan(q._h, 256, -1, C.c_uint32(0xFFFFFFFF), 0))
            q._check(L.nq_band_histogram_device(q._h, C.c_void_p(d_band.data_ptr()), d_band.numel(), C.c_void_p(hists[0].data_ptr())))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            pal = np.zeros(256, np.int32); k = C.c_int32(0)
            q._check(L.nq_palette_from_histograms_device(q._h, C.c_void_p(hists.data_ptr()), N, 256, pal.ctypes.data, C.byref(k)))
            torch.cuda.synchronize(); t2 = time.perf_counter()
            q.width, q.height = W, rows
            q.set_band(y0, H)
            q.dither_device(d_band.data_ptr(), pal[:k.value].copy(), True, d_out.data_ptr(), d_idx.data_ptr())
            q.set_band(0, 0)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            cur = {"band_scan_and_histogram_s": t1 - t0, "palette_build_s": t2 - t1, "band_dither_s": t3 - t2}
            if it and (best is None or sum(cur.values()) < sum(best.values())):
                best = cur
        # two all-gathers per step (3 int64; 2.6 MB of f64 partials per rank): 0.46 ms measured at world 1 under RCCL (profiles/r02); over
        # xGMI the 2.6 MB x N payload at ~50 GB/s per link adds ~0.05 ms per rank
        coll = 0.0005 + 0.00005 * N
        total = sum(best.values()) + coll
        res[N] = {"band_rows": rows, **{k2: round(v, 5) for k2, v in best.items()}, "collectives_s_estimate": round(coll, 5),
                  "seconds_per_image": round(total, 4), "mpixels_s_whole_job": round(W * H / total / 1e6, 1)}
        print("cfg5 N=%d: %s" % (N, res[N]), flush=True)
        del d_band, d_out, d_idx, hists, q
    return res


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in ("cfg4", "cfg5")] or ["cfg4", "cfg5"]
    if "cfg4" in which:
        out["cfg4"] = cfg4()
    if "cfg5" in which:
        out["cfg5"] = cfg5()
    print(json.dumps(out))
