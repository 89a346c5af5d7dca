#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: Mpixels/s, 4096x4096 RGBA -> 256-colour PnnLAB + dither.

One "step" = one whole convert(256, dither=true) of a 4096x4096 ARGB image that is already resident in HBM:
alpha pre-scan, histogram, find_nn, merge loop, palette fill, gilbert-curve error diffusion (PARALLEL_TILED).
Multi-GPU (driver: torch.distributed.run, one rank per GPU): every rank converts its own image (independent units,
no data-path collective; RCCL only for the barrier / max-over-ranks of the time) -> "scaling": "weak".

Prints ONE JSON line on rank 0 (contract in the round prompt) with two extra objects:
  roofline     -- dominant per-pixel kernel (gilbert_kernel = nearest-colour + dither pass), algorithmic bytes
                  8 B/pixel (4 B ARGB read + 4 B ARGB write) / its average duration measured with HIP
                  events on the launch stream inside the timed region, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (C restatement of the reference's sequential Java path, 1 core) on a bounded sample
                  of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# One hardware queue per in-flight convert: ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and
# a short kernel queued behind another stream's long merge kernel in the same hardware queue would wait for it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "gilbert_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command
BYTES_PER_PIXEL = 8             # 4 B ARGB read + 4 B ARGB write


def cpu_baseline(workload, sample):
    """Times the oracle (sequential C restatement of the reference, one core) on a sample x sample image of the same
    generator; ~10-30 s of CPU work."""
    import oracle_lib
    from nquant.android_amd import synth
    img = synth.gradient_noise(sample, sample, 3) if workload == "gradient_noise" else synth.uniform_rgb(sample, sample, 3)
    q = oracle_lib.OracleQuantizer(1, img, seed=3)
    t0 = time.perf_counter()
    q.convert(256, True)
    dt = time.perf_counter() - t0
    st = q.stage_seconds()
    return {"value": round(sample * sample / dt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%dx%d %s, whole convert(256,true), sequential C restatement of the reference Java path, %.1f s "
                      "(pnnquan %.1f s, dither %.1f s)" % (sample, sample, workload, dt,
                                                            st["histogram"] + st["nn_init"] + st["merge"], st["gilbert"]),
            "nproc": os.cpu_count()}


def measured_traffic(w, h):
    """HBM bytes per gilbert_kernel launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    `bench.py --steps 1 --concurrency 1`, summaries under profiles/): 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B; confirmed here on prescan_kernel: 32 787 KB for a 64 MiB read).
    Only valid for the image size it was measured on."""
    try:
        t = json.load(open(TRAFFIC_FILE))
        if t.get("width") == w and t.get("height") == h:
            return int(2 * t["FETCH_SIZE_KB"] * 1024 + t["WRITE_SIZE_KB"] * 1024)
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--workload", default="gradient_noise", choices=["gradient_noise", "uniform"])
    ap.add_argument("--tile", type=int, default=0, help="tile side of the PARALLEL_TILED decomposition (0 = automatic)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="side of the CPU-baseline sample image (0 = skip)")
    ap.add_argument("--concurrency", type=int, default=24,
                    help="independent converts in flight per GPU, each on its own HIP stream and quantizer handle (the merge "
                         "loop of one convert is a sequential chain on one CU; other converts fill the rest of the chip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import nquant.android_amd as nq
    from nquant.android_amd import synth
    nq.build_library()

    W = H = args.size
    npx = W * H
    seed = 3 + rank
    img = synth.gradient_noise(W, H, seed) if args.workload == "gradient_noise" else synth.uniform_rgb(W, H, seed)
    d_in = torch.from_numpy(img.reshape(-1)).cuda()
    d_out = torch.empty(npx, dtype=torch.int32, device="cuda")
    d_idx = torch.empty(npx, dtype=torch.int16, device="cuda")
    tile = args.tile
    if tile <= 0:      # the library's automatic rule (nq_set_tile): largest of 16, 8, 4 with >= 131072 tiles
        tile = next((c for c in (16, 8) if ((W + c - 1) // c) * ((H + c - 1) // c) >= 131072), 4)
    import threading
    C = max(1, min(args.concurrency, max(args.steps, 1)))
    lanes = []
    for c in range(C):
        st = torch.cuda.Stream()
        qq = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), device=local_rank, mode=nq.MODE_PARALLEL_TILED, seed=seed,
                                tile=(tile, tile))
        qq.width, qq.height = W, H
        qq.set_stream(st.cuda_stream)
        lanes.append({"q": qq, "stream": st,
                      "out": torch.empty(npx, dtype=torch.int32, device="cuda"),
                      "idx": torch.empty(npx, dtype=torch.int16, device="cuda"),
                      "stages": {}, "n": 0, "pal": None})
    q = lanes[0]["q"]

    def run_lane(ln, nsteps, record):
        # one host thread per lane: the C ABI blocks only on its own stream (ctypes releases the GIL)
        for _ in range(nsteps):
            ln["pal"] = ln["q"].convert_device(d_in.data_ptr(), 256, True, ln["out"].data_ptr(), ln["idx"].data_ptr())
            if record:
                for k, v in ln["q"].stage_ms().items():      # HIP events recorded on the launch stream, per stage
                    ln["stages"][k] = ln["stages"].get(k, 0.0) + v
                ln["n"] += 1

    def run_all(total, record):
        share = [total // C + (1 if c < total % C else 0) for c in range(C)]
        th = [threading.Thread(target=run_lane, args=(lanes[c], share[c], record)) for c in range(C) if share[c] > 0]
        for t in th:
            t.start()
        for t in th:
            t.join()

    torch.cuda.synchronize()
    # single-convert latency (one stream, nothing else in flight), untimed part of the warm-up
    t0 = time.perf_counter()
    run_lane(lanes[0], 1, False)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t0) * 1e3
    run_all(max(args.warmup, C), False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_all(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stages = {}
    nrec = sum(ln["n"] for ln in lanes)
    for ln in lanes:
        for k, v in ln["stages"].items():
            stages[k] = stages.get(k, 0.0) + v
    stages = {k: v / max(nrec, 1) for k, v in stages.items()}
    pal = lanes[0]["pal"]

    if rank == 0:
        p = q.params
        kernel_ms = stages["dither"]
        achieved = BYTES_PER_PIXEL * npx / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        line = {
            "metric": "Mpixels/sec, 4096x4096 RGBA -> 256-colour PnnLAB + dither",
            "value": round(world * args.steps * npx / dt / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d ARGB_8888 %s (seed 3+rank), PnnLABQuantizer.convert(256, dither=true), "
                                   "PARALLEL_TILED %dx%d tiles, one image per rank per step" % (W, H, args.workload, tile, tile),
                       "palette": int(len(pal)), "maxbins": int(p.maxbins), "parallelism": "1 image per GPU per step, no collective",
                       "concurrency": C, "single_convert_latency_ms": round(latency_ms, 2)},
            "stages_ms": {k: round(v, 3) for k, v in stages.items()},
            "merge_stats": q.merge_stats(),
            "pass_mpixels_s": round(npx / (kernel_ms * 1e-3) / 1e6, 1) if kernel_ms > 0 else None,
            "roofline": {"bound": "hbm", "kernel": "gilbert_kernel<false,25> (per-pixel nearest/closest colour + error diffusion)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": measured_traffic(W, H),
                         "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * npx, "kernel_ms": round(kernel_ms, 3)},
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
