"""Per-find_nn breakdown of one merge loop while a whole batch runs (contention of the co-resident merge workgroups included)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = H = 4096
qs, ins, outs = [], [], []
for b in range(B):
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), mode=1, seed=3 + b)
    q.width, q.height = W, H
    qs.append(q)
    ins.append(synth.gradient_noise_torch(W, H, 3 + b))
out = torch.empty(W * H, dtype=torch.int32, device="cuda")      # one output buffer: this tool only looks at the counters
for it in range(2):
    nq.convert_batch_device(qs, [t.data_ptr() for t in ins], 256, True, [out.data_ptr()] * B)
for k in (0, B // 2, B - 1):
    st = qs[k].merge_stats(); n = max(st["find_nn_calls"], 1)
    print("image %4d: finds %d | per find us: total %.2f ctrl %.2f bound %.2f (seed %.2f) exact %.2f replay %.2f | chunks %.1f" % (
        k, n, st["find_ticks_100MHz"] / n / 100, st["ctrl_ticks_100MHz"] / n / 100, st["bound_ticks"] / n / 100,
        st["seed_round_ticks"] / n / 100, st["exact_ticks"] / n / 100, st["replay_ticks"] / n / 100, st["chunks"] / n))
