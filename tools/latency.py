"""Single-convert latency and merge-loop counters on the bench image (4096^2 gradient+noise, LAB, 256 colours).
python tools/latency.py [side] [kind 0|1] [uniform]"""
import sys, time
import os; os.environ.setdefault("NQ_MERGE_STATS", "1")          # the stamped merge kernel: the per-phase ticks below need it (~4 % slower)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d_in = synth.gradient_noise_torch(W, H, 3) if (len(sys.argv) <= 3 or sys.argv[3] != "uniform") else torch.from_numpy(synth.uniform_rgb(W, H, 3).reshape(-1)).cuda()
out = torch.empty_like(d_in); idx = torch.empty(W * H, dtype=torch.int16, device="cuda")
q = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(np.zeros((1, 1), np.int32), mode=1, seed=3)
q.width, q.height = W, H
for it in range(2):
    t0 = time.perf_counter()
    pal = q.convert_device(d_in.data_ptr(), 256, True, out.data_ptr(), idx.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
st = q.merge_stats(); n = max(st["find_nn_calls"], 1)
print("latency %.1f ms  stages %s" % (dt * 1e3, {k: round(v, 2) for k, v in q.stage_ms().items()}))
print("finds %d merges %d | per find us: total %.2f ctrl %.2f bound %.2f (seed %.2f) exact %.2f replay %.2f | chunks %.1f l1 %.1f l2 %.1f listed %.1f exact %.2f" % (
    n, st["merges"], st["find_ticks_100MHz"] / n / 100, st["ctrl_ticks_100MHz"] / n / 100, st["bound_ticks"] / n / 100,
    st["seed_round_ticks"] / n / 100, st["exact_ticks"] / n / 100, st["replay_ticks"] / n / 100, st["chunks"] / n, st["chunks_l1"] / n,
    st["chunks_l2"] / n, st["chunks_listed"] / n, st["exact_evals"] / n))
print("overflows %d rebuilds %d ratio %.4f" % (st["overflows"], st["rebuilds"], q.params.ratio))
ts = q.team_stats()
print("team: helpers %d published %d used %d timeouts %d wait %.2f us per used result, speculating at end %d" % (
    ts["helpers"], ts["published"], ts["used"], ts["timeouts"], ts["wait_ticks_100MHz"] / max(ts["used"] + ts["timeouts"], 1) / 100, ts["speculating_at_end"]))
print("bin-info cache: top hits %d | results of virtual merges used %d | helpers declined %d results, given up on %d times" % (ts["cache_hits_top"], ts["virtual_merges_used"], ts["declined"], ts["gave_up"]))
print("control ms: total %.1f | sifts %.1f (pops %d) merges %.1f top fetch %.1f result waits %.1f find epilogues %.1f (incl. their sift) | wavefront 1 selection %.1f" % (
    st["ctrl_ticks_100MHz"] / 1e5, ts["sift_ticks"] / 1e5, ts["pops"], ts["merge_ticks"] / 1e5, ts["top_fetch_ticks"] / 1e5, ts["wait_ticks_100MHz"] / 1e5,
    ts["epilogue_ticks"] / 1e5, ts["select_ticks"] / 1e5))
import hashlib
print("palette sha", hashlib.sha256(pal.tobytes()).hexdigest()[:16])
