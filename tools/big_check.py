"""Robustness at the largest BASELINE size (16384x16384 = 268 M pixels, LAB, 256 colours + dither) and an RGB run at 4096^2:
size-independent properties only (output == palette[index], indices in range, reproducible palette)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth

def run(kind, W, H, label):
    t0 = time.perf_counter()
    band = 2048
    d_in = torch.empty(W * H, dtype=torch.int32, device="cuda")
    for y in range(0, H, band):                       # generate band by band (host memory)
        hh = min(band, H - y)
        # same generator, rows y..y+hh of the full image: gradient_noise is defined per pixel index
        part = synth.gradient_noise(W, H, 5)[y:y + hh] if H <= 4096 else None
        if part is None:
            n = W * hh
            z = synth.splitmix64(5, n, offset=y * W)
            yy, xx = np.divmod(np.arange(n, dtype=np.int64) + y * W, W)
            fx, fy = xx / (W - 1), yy / (H - 1)
            nz = [(((z >> np.uint64(s)) & np.uint64(0xFF)).astype(np.float64) / 255.0 - 0.5) for s in (0, 8, 16)]
            r = np.clip(np.rint(255.0 * fx + 24 * nz[0]), 0, 255).astype(np.uint32)
            g = np.clip(np.rint(255.0 * fy + 24 * nz[1]), 0, 255).astype(np.uint32)
            b = np.clip(np.rint(127.5 * (1.0 + np.sin(2.0 * np.pi * (0.75 * fx + 0.5 * fy))) + 24 * nz[2]), 0, 255).astype(np.uint32)
            part = ((np.uint32(255) << np.uint32(24)) | (r << np.uint32(16)) | (g << np.uint32(8)) | b).view(np.int32)
        d_in[y * W:(y + hh) * W] = torch.from_numpy(np.ascontiguousarray(part).reshape(-1)).cuda()
    d_out = torch.empty(W * H, dtype=torch.int32, device="cuda")
    d_idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
    q = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(np.zeros((1, 1), np.int32), mode=1, seed=5)
    q.width, q.height = W, H
    print(label, "generated in %.1f s" % (time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    pal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    idx = d_idx.view(torch.int16).to(torch.int64) & 0xFFFF
    assert int(idx.max()) < len(pal)
    palt = torch.from_numpy(pal).cuda()
    assert bool((palt[idx] == d_out).all())
    print(label, "ok: K=%d maxbins=%d convert %.2f s (%.1f Mpx/s) stages %s" % (len(pal), q.params.maxbins, dt, W * H / dt / 1e6,
          {k: round(v, 1) for k, v in q.stage_ms().items()}), flush=True)

run(0, 4096, 4096, "RGB 4096^2")
run(1, 16384, 16384, "LAB 16384^2")
