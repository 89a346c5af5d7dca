"""A/B of the dither pass on the bench image (4096^2 gradient+noise, LAB K=256): specialised kernel vs generic kernel --
equality of the outputs, time of each, tiles handed back.  Usage: python tools/dither_ab.py [size] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import nquant.android_amd as nq
from nquant.android_amd import synth, host

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
img = synth.gradient_noise(W, H, 3)
d_in = torch.from_numpy(img.reshape(-1)).cuda()
q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=3)
q.width, q.height = W, H
pal = q.pnnquan_device(d_in.data_ptr(), 256)
print("K", len(pal), "maxbins", q.params.maxbins, flush=True)
res = {}
for dither in (True, False):
    for tile in ((8, 8), (16, 16), (4, 4)):
        q.set_tile(*tile)
        outs = {}
        for fast in (1, 0):
            q.set_option(host.OPT_FAST_DITHER, fast)
            d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
            d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
            best = 1e9
            for it in range(reps):
                torch.cuda.synchronize(); t = time.perf_counter()
                q.dither_device(d_in.data_ptr(), pal, dither, d_out.data_ptr(), d_idx.data_ptr())
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            path = q.dither_path()
            outs[fast] = (d_out.cpu().numpy(), d_idx.cpu().numpy())
            print("dither=%d tile=%s fast=%d: call %.3f ms (lists + saliency + dither kernels), path %s" % (dither, tile, fast, best * 1e3, path), flush=True)
        same = (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
        nd = int((outs[0][1] != outs[1][1]).sum())
        print("   fast == generic: %s (differing indices %d)" % (same, nd), flush=True)
