"""Throughput of nq_convert_batch_device for either kind: python tools/batch_rate.py <batch> <kind 0|1> [side]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
B = int(sys.argv[1]); kind = int(sys.argv[2]); W = H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cls = nq.PnnLABQuantizer if kind else nq.PnnQuantizer
free_b, total_b = torch.cuda.mem_get_info()
print("device memory: free %.1f GiB of %.1f GiB; batch needs %.1f GiB" % (free_b / 2**30, total_b / 2**30, B * (10 * W * H + (12 << 20)) / 2**30), flush=True)
if B * (10 * W * H + (12 << 20)) + (8 << 30) > free_b:
    raise SystemExit("batch does not fit")
qs, ins, outs = [], [], []
for b in range(B):
    q = cls(np.zeros((1, 1), np.int32), mode=1, seed=3 + b)
    q.width, q.height = W, H
    qs.append(q); ins.append(synth.gradient_noise_torch(W, H, 3 + b)); outs.append(torch.empty(W * H, dtype=torch.int32, device="cuda"))
idx = [torch.empty(W * H, dtype=torch.int16, device="cuda") for _ in range(B)]
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pals = nq.convert_batch_device(qs, [t.data_ptr() for t in ins], 256, True, [t.data_ptr() for t in outs], [t.data_ptr() for t in idx])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("kind %d batch %d: %.2f s, %.1f Mpx/s, %.2f ms per image, maxbins %d, stages %s" % (kind, B, dt, B * W * H / dt / 1e6, dt / B * 1e3,
          qs[0].params.maxbins, {k: round(v, 2) for k, v in qs[0].stage_ms().items()}), flush=True)
st = qs[0].merge_stats(); n = max(st["find_nn_calls"], 1)
print("image 0 in the batch, per find us: find %.2f ctrl %.2f | bound %.2f (seed %.2f) exact %.2f replay %.2f | chunks %.1f exact evals %.2f overflows %d" % (
    st["find_ticks_100MHz"] / n / 100, st["ctrl_ticks_100MHz"] / n / 100, st["bound_ticks"] / n / 100, st["seed_round_ticks"] / n / 100,
    st["exact_ticks"] / n / 100, st["replay_ticks"] / n / 100, st["chunks"] / n, st["exact_evals"] / n, st["overflows"]))
ts = qs[0].team_stats()
print("image 0 team: helpers %d published %d used %d timeouts %d wait %.2f us per waited result, speculating at end %d" % (
    ts["helpers"], ts["published"], ts["used"], ts["timeouts"], ts["wait_ticks_100MHz"] / max(ts["used"] + ts["timeouts"], 1) / 100, ts["speculating_at_end"]))
tss = [q.team_stats() for q in qs]
print("loops that gave up on their helpers at least once: %d of %d (%d times in all); without them at the end: %d" % (
    sum(1 for t in tss if t["gave_up"]), B, sum(t["gave_up"] for t in tss), sum(1 for t in tss if t["helpers"] > 0 and not t["speculating_at_end"])))
if not st["find_ticks_100MHz"]:
    print("(the batch variants of the merge kernel carry no phase stamps: NQ_MERGE_STATS=1 selects the stamped 128-thread build)")
print("batch phases %s" % qs[0].batch_phase_ms())
if st["find_ticks_100MHz"] > 0 and st["ctrl_ticks_100MHz"] > 0:          # (stamped build only: without stamps the tick fields hold no times)
    busy = sorted((q.merge_stats()["find_ticks_100MHz"] + q.merge_stats()["ctrl_ticks_100MHz"]) / 1e5 for q in qs)
    print("loop busy time (find + control) ms: min %.0f median %.0f p99 %.0f max %.0f" % (busy[0], busy[len(busy) // 2], busy[int(len(busy) * 0.99)], busy[-1]))
print("image 0 control ms: total %.1f | sifts %.1f (pops %d) merges %.1f top fetch %.1f find epilogues %.1f (incl. their sift)" % (
    st["ctrl_ticks_100MHz"] / 1e5, ts["sift_ticks"] / 1e5, ts["pops"], ts["merge_ticks"] / 1e5, ts["top_fetch_ticks"] / 1e5, ts["epilogue_ticks"] / 1e5))
