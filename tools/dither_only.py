"""Runs the dither pass of the bench image `reps` times (for rocprofv3 --kernel-trace --stats / --pmc).
Usage: python tools/dither_only.py [size] [reps] [tile] [fast 0|1] [dither 0|1] [mode 1 = PARALLEL_TILED | 2 = LOOKUP_ONLY]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import nquant.android_amd as nq
from nquant.android_amd import synth, host

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 8
fast = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dither = int(sys.argv[5]) if len(sys.argv) > 5 else 1
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 1
img = synth.gradient_noise(W, H, 3)
d_in = torch.from_numpy(img.reshape(-1)).cuda()
q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=3)
q.width, q.height = W, H
pal = q.pnnquan_device(d_in.data_ptr(), 256)
q.set_tile(tile, tile)
q.set_option(host.OPT_FAST_DITHER, fast)
d_out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
d_idx = torch.zeros(W * H, dtype=torch.int16, device="cuda")
acc = []
for it in range(reps):
    q.dither_device(d_in.data_ptr(), pal, bool(dither), d_out.data_ptr(), d_idx.data_ptr(), mode=mode)
    torch.cuda.synchronize()
    acc.append(q.stage_ms()["dither"])
import hashlib
print("done", q.dither_path(), "dither stage ms: min %.4f median %.4f" % (min(acc), sorted(acc)[len(acc) // 2]),
      "index sha", hashlib.sha256(d_idx.cpu().numpy().tobytes()).hexdigest()[:16], "NQ_FAST_NO_OPAQUE" in os.environ)
