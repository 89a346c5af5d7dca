"""The other BASELINE.json configurations on one GPU (timings for DESIGN.md; parity for them is in tests/):
cfg 2 1024x1024 LAB 256 colours LOOKUP_ONLY; cfg 4 batch of 64 x 1920x1080 frames, LAB 256 + dither (one nq_convert_batch_device call);
cfg 5 16384x16384 LAB 256 + dither as ONE image (single GPU here; the 8-band split is tests/ + parallel.py)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth

def sync(): torch.cuda.synchronize()

# cfg 2
W = H = 1024
img = synth.uniform_rgb(W, H, 2)
q = nq.PnnLABQuantizer(img, mode=nq.MODE_PARALLEL_TILED, seed=2)
pal = q.pnnquan(256)
d_in = torch.from_numpy(img.reshape(-1)).cuda(); d_out = torch.empty_like(d_in); d_idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
for it in range(3):
    sync(); t0 = time.perf_counter()
    q.dither_device(d_in.data_ptr(), pal, False, d_out.data_ptr(), d_idx.data_ptr(), mode=nq.MODE_LOOKUP_ONLY)
    sync(); dt = time.perf_counter() - t0
print("cfg2 1024^2 uniform, LOOKUP_ONLY (palette of %d bins image built in the same object): %.3f ms = %.0f Mpx/s" % (q.params.maxbins, dt * 1e3, W * H / dt / 1e6), flush=True)

# cfg 4
W, H, B = 1920, 1080, 64
qs, ins, outs, idxs = [], [], [], []
for f in range(B):
    qq = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=100 + f); qq.width, qq.height = W, H
    qs.append(qq); ins.append(synth.gradient_noise_torch(W, H, 100 + f))
    outs.append(torch.empty(W * H, dtype=torch.int32, device="cuda")); idxs.append(torch.empty(W * H, dtype=torch.int16, device="cuda"))
for it in range(2):
    sync(); t0 = time.perf_counter()
    nq.convert_batch_device(qs, [t.data_ptr() for t in ins], 256, True, [t.data_ptr() for t in outs], [t.data_ptr() for t in idxs])
    sync(); dt = time.perf_counter() - t0
print("cfg4 64 x 1920x1080 frames, LAB 256 + dither, one batch call on ONE GPU: %.2f s = %.1f frames/s = %.0f Mpx/s (maxbins %d)" % (
    dt, B / dt, B * W * H / dt / 1e6, qs[0].params.maxbins), flush=True)
del qs, ins, outs, idxs
torch.cuda.empty_cache()

# cfg 5 (one image, one GPU)
W = H = 16384
d_in = torch.cat([synth.gradient_noise_torch(W, 2048, 5 + b) for b in range(8)])      # 8 bands generated separately (memory)
d_out = torch.empty_like(d_in); d_idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=5); q.width, q.height = W, H
for it in range(2):
    sync(); t0 = time.perf_counter()
    pal = q.convert_device(d_in.data_ptr(), 256, True, d_out.data_ptr(), d_idx.data_ptr())
    sync(); dt = time.perf_counter() - t0
print("cfg5 16384^2 as one image on ONE GPU: %.2f s = %.0f Mpx/s, stages %s" % (dt, W * H / dt / 1e6, {k: round(v, 1) for k, v in q.stage_ms().items()}), flush=True)
