"""PCIe-inclusive rate of the host-buffer batch entry (nq_convert_batch): B 4096x4096 images in page-locked host memory,
uploads and read-backs overlapped with the per-image stages.  Not the headline metric (bench.py keeps inputs resident in HBM)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = H = 4096
npx = W * H
qs, h_in, h_out, h_idx = [], [], [], []
for b in range(B):
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=3 + b)
    q.width, q.height = W, H
    qs.append(q)
    h_in.append(synth.gradient_noise_torch(W, H, 3 + b).cpu().pin_memory())
    h_out.append(torch.empty(npx, dtype=torch.int32).pin_memory())
    h_idx.append(torch.empty(npx, dtype=torch.int16).pin_memory())
for it in range(2):
    t0 = time.perf_counter()
    pals = nq.convert_batch_host(qs, [t.data_ptr() for t in h_in], 256, True, [t.data_ptr() for t in h_out], [t.data_ptr() for t in h_idx])
    dt = time.perf_counter() - t0
    print("batch %d host buffers (pinned): %.2f s, %.1f Mpx/s PCIe-inclusive, %.2f ms per image; moved %.1f GB" % (
        B, dt, B * npx / dt / 1e6, dt / B * 1e3, B * npx * 10 / 1e9), flush=True)
pal = torch.from_numpy(pals[0])
assert bool((pal[(h_idx[0].to(torch.int64) & 0xFFFF)] == h_out[0]).all())
