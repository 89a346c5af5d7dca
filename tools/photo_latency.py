"""Single convert of the `photo` workload (the reference's sample.jpg tiled to side^2) for either kind, with the merge counters of the
stamped build.  python tools/photo_latency.py [side] [kind 0|1]"""
import os, sys, time
os.environ.setdefault("NQ_MERGE_STATS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rgb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sample_495x438.npz"))["rgb"]
d_in = torch.from_numpy(synth.tile_photo(rgb, W, H, 0).reshape(-1)).cuda()
out = torch.empty_like(d_in); idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
q = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(np.zeros((1, 1), np.int32), mode=1, seed=3)
q.width, q.height = W, H
for it in range(2):
    t0 = time.perf_counter()
    pal = q.convert_device(d_in.data_ptr(), 256, True, out.data_ptr(), idx.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
st = q.merge_stats(); n = max(st["find_nn_calls"], 1); ts = q.team_stats()
print("kind %d photo %dx%d: latency %.1f ms  stages %s" % (kind, W, H, dt * 1e3, {k: round(v, 2) for k, v in q.stage_ms().items()}))
print("maxbins %d finds %d merges %d | per find us: total %.2f ctrl %.2f | chunks %.1f exact %.2f overflows %d rebuilds %d" % (
    q.params.maxbins, n, st["merges"], st["find_ticks_100MHz"] / n / 100, st["ctrl_ticks_100MHz"] / n / 100, st["chunks"] / n, st["exact_evals"] / n, st["overflows"], st["rebuilds"]))
print("team: helpers %d published %d used %d timeouts %d declined %d gave up %d" % (ts["helpers"], ts["published"], ts["used"], ts["timeouts"], ts["declined"], ts["gave_up"]))
