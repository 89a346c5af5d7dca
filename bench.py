#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: Mpixels/s, 4096x4096 RGBA -> 256-colour PnnLAB + dither.

One "step" = one batch (--batch, default 1536 = 96 GiB of input, 258 GiB with outputs and state: six merge loops per CU; shrunk to what
the device's free memory holds) of distinct 4096x4096 ARGB images, already resident in HBM, each through the
whole convert(256, dither=true): alpha pre-scan, histogram, find_nn, merge loop, palette fill, gilbert-curve error diffusion
(PARALLEL_TILED).  The merge loop of one image is a sequential chain on one CU, so images are handed over in batches
(nq_convert_batch_device): the merge loops of a batch run side by side, one workgroup each (six per CU at this batch size).
Device memory: batch x 160 MiB of pixel buffers (in, out, index) + 10 MiB of quantizer state per image.
Multi-GPU (driver: torch.distributed.run, one rank per GPU): every rank converts its own batches (independent units,
no data-path collective; RCCL only for the barrier / max-over-ranks of the time) -> "scaling": "weak".

Prints ONE JSON line on rank 0 (contract in the round prompt) with extra objects:
  roofline       -- dominant per-pixel kernel (gilbert_fast_kernel = nearest-colour + dither pass), algorithmic bytes
                    8 B/pixel (4 B ARGB read + 4 B ARGB write) / its average duration measured with HIP
                    events on the launch stream inside the timed region, against the 8 TB/s HBM peak;
  roofline_whole -- the same 8 B/pixel against the time of the WHOLE convert per image (all stages);
  roofline_lookup -- fast_lookup_pass1_kernel + fast_lookup_pass2_kernel (MODE_LOOKUP_ONLY: nearestColorIndex per pixel, the "dither off,
                    bit-exact indices" half of the north star) on the same image and palette, measured after the timed region: 4 B read +
                    2 B index + 4 B ARGB written = 10 B/pixel against the HIP-event time of the two passes (the list of the pixels the
                    float32 pass defers to the exact pass is not counted as algorithmic);
  batch_sweep, cfg3a_uniform, photo, host_batch, single_image_mpixels_s -- N = 1 only, after the timed region (--no-extras skips them):
                    smaller batches of the same images; BASELINE cfg 3 type (a) (uniform random colours, 65 536 bins: the worst case for
                    pnnquan) as single-image latency and a batch of 256; the reference's own sample photograph tiled to 4096^2 (a photographic
                    histogram: 2970 bins, sorted-by-yDiff queue, generic dither kernel); the PCIe-inclusive rate of
                    nq_convert_batch over page-locked host buffers (never the headline `value`);
  amortised_ms_per_image -- the three phases of a batch call (HIP events on the launch stream) divided by the batch size;
  cpu_baseline   -- the CPU oracle (C restatement of the reference's sequential Java path, 1 core) on the SAME 4096x4096 image
                    and seed as slot 0 of the batch (--cpu-sample shrinks it; the size is in its "sample" field), rank 0, N=1 only.

--config cfg3 (default) is the headline above.  The other two BASELINE configurations exercise the multi-GPU shapes:
  --config cfg4  batch of 64 x 1920x1080 frames, frame f on rank f mod N (independent units, no data-path collective; strong scaling);
  --config cfg5  one 16384x16384 image cut into N row bands: pre-scan all-reduce + all-gather of the 2.6 MB histogram partials
                 (RCCL), the palette built on every rank, every rank dithers its band (strong scaling); the collectives are inside
                 the timed region and their bytes / time are reported in "collectives".
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
    what = "the headline image itself (slot 0 of the batch, seed 3)" if sample == 4096 else "a bounded sample, NOT the 4096x4096 image of the headline"
    return {"value": round(sample * sample / dt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%dx%d %s, seed 3 (%s), ONE whole convert(256,true), sequential C restatement of the reference Java path, "
                      "%.1f s (pnnquan %.1f s, dither %.1f s)" % (sample, sample, workload, what, dt,
                                                            st["histogram"] + st["nn_init"] + st["merge"], st["gilbert"]),
            "nproc": os.cpu_count()}


LOOKUP_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "lookup_traffic.json")


def photo_image(W, H, slot=0):
    """The reference's sample photograph (tests/golden/sample_495x438.npz: decoded pixels of app/src/main/res/drawable/sample.jpg) tiled to
    W x H; slot k adds k % 7 to the red channel (clipped) so that the images of a batch differ (synth.tile_photo)."""
    from nquant.android_amd import synth
    return synth.tile_photo(np.load(os.path.join(ROOT, "tests", "golden", "sample_495x438.npz"))["rgb"], W, H, slot)


def run_extras(nq, synth, slots, W, H, latency_ms, tile):
    """Secondary measurements of the driver-run line (rank 0, N = 1, after the timed region); each is a few seconds."""
    npx = W * H
    out = {"single_image_mpixels_s": round(npx / (latency_ms * 1e-3) / 1e6, 2)}

    def batch_rate(sl, steps, ins=None, warm=1):
        qs = [s["q"] for s in sl]
        ins = ins if ins is not None else [s["in"].data_ptr() for s in sl]
        outs, idxs = [s["out"].data_ptr() for s in sl], [s["idx"].data_ptr() for s in sl]
        for _ in range(warm):
            nq.convert_batch_device(qs, ins, 256, True, outs, idxs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ph = {}
        for _ in range(steps):
            pals = nq.convert_batch_device(qs, ins, 256, True, outs, idxs)
            for k, v in qs[0].batch_phase_ms().items():
                ph[k] = ph.get(k, 0.0) + v / (steps * len(sl))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"batch": len(sl), "steps": steps, "mpixels_s": round(steps * len(sl) * npx / dt / 1e6, 1), "ms_per_image": round(dt / (steps * len(sl)) * 1e3, 3),
                "amortised_ms_per_image": {k: round(v, 4) for k, v in ph.items()}, "maxbins": int(qs[0].params.maxbins), "palette": int(len(pals[0]))}

    def single(q, d_in, s):
        q.convert_device(d_in.data_ptr(), 256, True, s["out"].data_ptr(), s["idx"].data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q.convert_device(d_in.data_ptr(), 256, True, s["out"].data_ptr(), s["idx"].data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    # 1. smaller batches of the headline images (the headline needs ~1500 merge loops side by side)
    out["batch_sweep"] = [batch_rate(slots[:n], 2) for n in (64, 256, 1024) if n < len(slots)]
    # 2. BASELINE cfg 3 type (a): uniform random colours, every one of the 65 536 histogram bins occupied
    nu = min(4, len(slots))
    uni = [torch.from_numpy(synth.uniform_rgb(W, H, 3 + k).reshape(-1)).cuda() for k in range(nu)]
    lat = single(slots[0]["q"], uni[0], slots[0])
    st = slots[0]["q"].stage_ms()
    nb = min(256, len(slots))
    r = batch_rate(slots[:nb], 1, ins=[uni[k % nu].data_ptr() for k in range(nb)])
    r.update({"workload": "%dx%d uniform random opaque colours (seeds 3..%d, cycled), PnnLABQuantizer.convert(256, true)" % (W, H, 2 + nu),
              "single_convert_latency_ms": round(lat, 1), "single_image_mpixels_s": round(npx / (lat * 1e-3) / 1e6, 2),
              "single_convert_stages_ms": {k: round(v, 3) for k, v in st.items()}})
    out["cfg3a_uniform"] = r
    del uni
    # 3. a photographic histogram: the reference's sample picture tiled to the headline size
    try:
        nph = min(64, len(slots))
        pho = [torch.from_numpy(photo_image(W, H, k).reshape(-1)).cuda() for k in range(min(7, nph))]
        for s in slots[:nph]:
            s["q"].set_tile(0, 0)            # the library's automatic rule
        lat = single(slots[0]["q"], pho[0], slots[0])
        st = slots[0]["q"].stage_ms()
        fast = slots[0]["q"].dither_path()[0]
        r = batch_rate(slots[:nph], 1, ins=[pho[k % len(pho)].data_ptr() for k in range(nph)])
        r.update({"workload": "the reference's sample.jpg (495x438 photograph) tiled to %dx%d, slot k: red + k %% 7; PnnLABQuantizer.convert(256, true)" % (W, H),
                  "single_convert_latency_ms": round(lat, 1), "single_image_mpixels_s": round(npx / (lat * 1e-3) / 1e6, 2),
                  "single_convert_stages_ms": {k: round(v, 3) for k, v in st.items()}, "specialised_dither_kernel": int(fast)})
        # ... and the call the reference's demo app actually makes on it: new PnnQuantizer(path).convert(256, true), the RGB kind
        # (app/src/main/java/nQuant/android/MainActivity.java:190-194)
        qr = nq.PnnQuantizer(np.zeros((1, 1), np.int32), mode=nq.MODE_PARALLEL_TILED, seed=3)
        qr.width, qr.height = W, H
        lat_rgb = single(qr, pho[0], slots[0])
        r["rgb_kind_single_convert_latency_ms"] = round(lat_rgb, 1)
        r["rgb_kind_single_convert_stages_ms"] = {k: round(v, 3) for k, v in qr.stage_ms().items()}
        del qr
        out["photo"] = r
        for s in slots[:nph]:
            s["q"].set_tile(tile, tile)
        del pho
    except FileNotFoundError:
        out["photo"] = None
    # 4. PCIe-inclusive: nq_convert_batch over page-locked host buffers (uploads / read-backs overlap the per-image stages)
    nh = min(128, len(slots))
    sl = slots[:nh]
    h_in = [s["in"].cpu().pin_memory() for s in sl]
    h_out = [torch.empty(npx, dtype=torch.int32).pin_memory() for _ in sl]
    h_idx = [torch.empty(npx, dtype=torch.int16).pin_memory() for _ in sl]
    best = None
    for it in range(2):
        t0 = time.perf_counter()
        pals = nq.convert_batch_host([s["q"] for s in sl], [t.data_ptr() for t in h_in], 256, True, [t.data_ptr() for t in h_out], [t.data_ptr() for t in h_idx])
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ok = bool((torch.from_numpy(pals[0])[(h_idx[0].to(torch.int64) & 0xFFFF)] == h_out[0]).all())
    out["host_batch"] = {"what": "nq_convert_batch: %d images of %dx%d in page-locked HOST memory, 10 B/pixel over PCIe (4 in, 4 + 2 out), copies overlapped "
                                 "with the per-image stages; PCIe-inclusive, NOT the headline" % (nh, W, H),
                         "batch": nh, "mpixels_s": round(nh * npx / best / 1e6, 1), "seconds": round(best, 3), "gb_moved": round(nh * npx * 10 / 1e9, 1),
                         "outputs_match_palette": ok}
    return out


def measured_traffic(w, h, path=None):
    """HBM bytes per gilbert_kernel launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    `bench.py --steps 1 --concurrency 1`, summaries under profiles/): 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B; confirmed here on prescan_kernel: 32 787 KB for a 64 MiB read).
    Only valid for the image size it was measured on."""
    try:
        t = json.load(open(path or TRAFFIC_FILE))
        if t.get("width") == w and t.get("height") == h:
            return int(2 * t["FETCH_SIZE_KB"] * 1024 + t["WRITE_SIZE_KB"] * 1024)
    except Exception:
        pass
    return None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries loaded later write there too (RCCL prints a version banner to stdout when its
    communicator starts), so file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved descriptor."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, "w")


_JSON_OUT = None


def emit(line):
    _JSON_OUT.write(json.dumps(line) + "\n")
    _JSON_OUT.flush()


def main():
    global _JSON_OUT
    _JSON_OUT = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="one step = one batch of --batch images through the whole hot path")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg4", "cfg5"],
                    help="BASELINE.json configuration: cfg3 = the headline (4096^2 batches), cfg4 = 64 x 1920x1080 frames sharded over the "
                         "ranks, cfg5 = one 16384^2 image in row bands with the RCCL histogram exchange")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--workload", default="gradient_noise", choices=["gradient_noise", "uniform", "photo"],
                    help="gradient_noise = BASELINE cfg 3 type (b), the headline; uniform = type (a), 65 536 bins; photo = the reference's sample.jpg tiled to --size")
    ap.add_argument("--tile", type=int, default=0, help="tile side of the PARALLEL_TILED decomposition (0 = automatic)")
    ap.add_argument("--cpu-sample", type=int, default=4096,
                    help="side of the CPU-baseline image (default: the headline's own 4096, ~80 s on one core; 0 = skip)")
    ap.add_argument("--no-dither", action="store_true", help="cfg5 only: convert(256, dither=false) -- the BlueNoise leg with the image-wide "
                                                             "distinct-colour count (presence-table all-reduce)")
    ap.add_argument("--batch", type=int, default=1536,
                    help="images per step (distinct synthetic images, all resident in HBM): the merge loop of one image is a "
                         "sequential chain on one CU, a batch runs its merge loops side by side (nq_convert_batch_device)")
    ap.add_argument("--concurrency", type=int, default=1,
                    help="host threads / HIP streams a step's batch is split over (measured: one call for the whole batch is best -- the "
                         "library then runs 128-thread merge workgroups, four per CU, and the per-pixel stages have the chip to themselves)")
    ap.add_argument("--no-extras", action="store_true", help="skip batch_sweep / cfg3a_uniform / photo / host_batch (N = 1 only, after the timed region)")
    ap.add_argument("--stagger", type=float, default=0.0,
                    help="with --concurrency T > 1: host thread t starts its first batch t * stagger seconds late, so that the merge loops of "
                         "one sub-batch (which leave most issue slots idle) run while another sub-batch is in its per-pixel stages")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU; a rehearsal with more ranks than GPUs (NQ_BENCH_BACKEND=gloo on a one-GPU box) wraps around
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("NQ_BENCH_BACKEND", "nccl")      # "nccl" = RCCL over xGMI; "gloo" only to rehearse the multi-rank path
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    elif args.config == "cfg5":
        import torch.distributed as dist       # the band pipeline always talks through a process group (world 1 here)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=0, world_size=1)

    import threading
    import nquant.android_amd as nq
    from nquant.android_amd import synth
    # the library ships prebuilt with the repo snapshot; should a rebuild be needed, exactly one rank does it
    if dist is None:
        nq.build_library()
    else:
        if rank == 0:
            nq.build_library()
        dist.barrier()

    if args.config == "cfg4":
        return bench_cfg4(args, nq, synth, dist, rank, local_rank, world)
    if args.config == "cfg5":
        return bench_cfg5(args, nq, synth, dist, rank, local_rank, world)

    W = H = args.size
    npx = W * H
    Bn = max(1, args.batch)
    # every image of the batch is resident: 10 B/pixel of buffers + ~10 MiB of quantizer state, plus the per-pixel scratch of the
    # batch (~20 B/pixel) and headroom; shrink the batch rather than run out of device memory
    free_b, _total_b = torch.cuda.mem_get_info()
    per_image = 10 * npx + (12 << 20)
    fit = int((free_b - 24 * npx - (6 << 30)) // per_image)
    if fit < Bn:
        Bn = max(1, fit)
    T = max(1, min(args.concurrency, Bn))
    tile = args.tile
    if tile <= 0:      # the library's automatic rule (nq_set_tile): 8x8 with >= 131072 tiles, else 4x4
        tile = 8 if ((W + 7) // 8) * ((H + 7) // 8) >= 131072 else 4

    # the batch: Bn distinct images (seed 3 + rank * Bn + slot), inputs and outputs resident in HBM
    slots = []
    uniform = None
    if args.workload == "uniform":
        uniform = [torch.from_numpy(synth.uniform_rgb(W, H, 3 + rank * 8 + k).reshape(-1)).cuda() for k in range(min(8, Bn))]
    elif args.workload == "photo":
        uniform = [torch.from_numpy(photo_image(W, H, k).reshape(-1)).cuda() for k in range(min(7, Bn))]
    for b in range(Bn):
        seed = 3 + rank * Bn + b
        d_in = synth.gradient_noise_torch(W, H, seed) if uniform is None else uniform[b % len(uniform)]
        q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), device=local_rank, mode=nq.MODE_PARALLEL_TILED, seed=seed,
                               tile=None if (args.workload == "photo" and args.tile <= 0) else (tile, tile))
        q.width, q.height = W, H
        slots.append({"q": q, "in": d_in, "out": torch.empty(npx, dtype=torch.int32, device="cuda"),
                      "idx": torch.empty(npx, dtype=torch.int16, device="cuda")})
    torch.cuda.synchronize()
    share = [Bn // T + (1 if t < Bn % T else 0) for t in range(T)]
    groups, o = [], 0
    for t in range(T):
        g = slots[o:o + share[t]]
        o += share[t]
        st = torch.cuda.Stream()
        g[0]["q"].set_stream(st.cuda_stream)          # a batch runs on its first handle's stream
        groups.append({"slots": g, "stream": st, "stages": {}, "phases": {}, "n": 0, "pals": None, "err": None, "index": t})

    def run_group(gr, nsteps, record):
        # one host thread per sub-batch: the C ABI blocks only on its own stream (ctypes releases the GIL)
        try:
            sl = gr["slots"]
            if args.stagger > 0 and gr["index"] > 0:
                time.sleep(args.stagger * gr["index"])
            for _ in range(nsteps):
                gr["pals"] = nq.convert_batch_device([s["q"] for s in sl], [s["in"].data_ptr() for s in sl], 256, True,
                                                     [s["out"].data_ptr() for s in sl], [s["idx"].data_ptr() for s in sl])
                if record:
                    for k, v in sl[0]["q"].batch_phase_ms().items():      # phases of the batch call (events on the launch stream)
                        gr["phases"][k] = gr["phases"].get(k, 0.0) + v
                    for s in sl:
                        for k, v in s["q"].stage_ms().items():      # HIP events recorded on the launch stream, per stage
                            gr["stages"][k] = gr["stages"].get(k, 0.0) + v
                        gr["n"] += 1
        except Exception as e:          # surfaced after the join
            gr["err"] = e

    def run_all(nsteps, record):
        th = [threading.Thread(target=run_group, args=(gr, nsteps, record)) for gr in groups]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for gr in groups:
            if gr["err"] is not None:
                raise gr["err"]

    # single-convert latency (one image, one stream, nothing else in flight), untimed
    q0 = slots[0]["q"]
    q0.convert_device(slots[0]["in"].data_ptr(), 256, True, slots[0]["out"].data_ptr(), slots[0]["idx"].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q0.convert_device(slots[0]["in"].data_ptr(), 256, True, slots[0]["out"].data_ptr(), slots[0]["idx"].data_ptr())
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t0) * 1e3
    single_stages = q0.stage_ms()
    # the same convert once more with the STAMPED build of the merge kernel (NQ_MERGE_STATS=1: phase ticks for "merge_stats"; ~4 % slower, untimed)
    os.environ["NQ_MERGE_STATS"] = "1"
    try:
        q0.convert_device(slots[0]["in"].data_ptr(), 256, True, slots[0]["out"].data_ptr(), slots[0]["idx"].data_ptr())
        torch.cuda.synchronize()
    finally:
        del os.environ["NQ_MERGE_STATS"]
    single_merge_stats = q0.merge_stats()
    run_all(args.warmup, False)

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
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stages = {}
    nrec = sum(gr["n"] for gr in groups)
    for gr in groups:
        for k, v in gr["stages"].items():
            stages[k] = stages.get(k, 0.0) + v
    stages = {k: v / max(nrec, 1) for k, v in stages.items()}
    phases = {}
    for gr in groups:
        for k, v in gr["phases"].items():
            phases[k] = phases.get(k, 0.0) + v
    phases = {k: v / max(nrec, 1) for k, v in phases.items()}        # per image: a group's phase span / its images, averaged over the steps
    pals = groups[0]["pals"]
    # sanity of the timed work itself: every output pixel is its palette entry, palettes are full and differ between images
    g0 = groups[0]
    for k in sorted({0, len(g0["slots"]) // 2, len(g0["slots"]) - 1}):
        sk = g0["slots"][k]
        palt = torch.from_numpy(pals[k]).cuda()
        if len(pals[k]) != 256 or not bool((palt[(sk["idx"].to(torch.int64) & 0xFFFF)] == sk["out"]).all()):
            raise SystemExit("bench: output pixels of image %d do not match palette[index]" % k)
    if len(g0["slots"]) > 1 and args.workload == "gradient_noise" and bool((pals[0] == pals[-1]).all()):
        raise SystemExit("bench: distinct images produced identical palettes")

    # LOOKUP_ONLY (nearestColorIndex per pixel, no diffusion) on slot 0 with its palette: the HBM-streaming kernel of the path
    lookup_ms = None
    if rank == 0:
        reps = 20
        acc = 0.0
        for r in range(reps + 2):
            q0.dither_device(slots[0]["in"].data_ptr(), pals[0], False, slots[0]["out"].data_ptr(), slots[0]["idx"].data_ptr(),
                             mode=nq.MODE_LOOKUP_ONLY)
            torch.cuda.synchronize()
            if r >= 2:
                acc += q0.stage_ms()["dither"]
        lookup_ms = acc / reps
    if rank == 0:
        p = q0.params
        kernel_ms = stages["dither"]
        achieved = BYTES_PER_PIXEL * npx / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        images = args.steps * Bn
        line = {
            "metric": "Mpixels/sec, 4096x4096 RGBA -> 256-colour PnnLAB + dither",
            "value": round(world * images * npx / dt / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d ARGB_8888 %s, PnnLABQuantizer.convert(256, dither=true), PARALLEL_TILED %dx%d tiles; "
                                   "one step = a batch of %d distinct images per rank (seeds 3 + rank*batch + slot), resident in HBM"
                                   % (W, H, args.workload, tile, tile, Bn),
                       "palette": int(len(pals[0])), "maxbins": int(p.maxbins),
                       "parallelism": "independent images per GPU, no collective",
                       "collective_backend": (backend if world > 1 else None),
                       "batch": Bn, "concurrency": T, "images_per_s": round(world * images / dt, 2),
                       "ms_per_image": round(dt / images * 1e3, 3),
                       "single_convert_latency_ms": round(latency_ms, 2)},
            "amortised_ms_per_image": {"prepare (pre-scan + histogram + initial find_nn pass)": round(phases.get("prepare", 0.0), 4),
                                       "merge (all merge loops of the batch in one launch + palette fill)": round(phases.get("merge", 0.0), 4),
                                       "finish (palette read-back + candidate lists + saliency + dither pass)": round(phases.get("finish", 0.0), 4),
                                       "of which the dither kernel": round(kernel_ms, 4),
                                       "sum": round(phases.get("total", 0.0), 4)},
            "single_convert_stages_ms": {k: round(v, 3) for k, v in single_stages.items()},
            "merge_stats": single_merge_stats,
            "pass_mpixels_s": round(npx / (kernel_ms * 1e-3) / 1e6, 1) if kernel_ms > 0 else None,
            "roofline": {"bound": "hbm", "kernel": "gilbert_fast_kernel (per-pixel nearest/closest colour + error diffusion, csrc/nq_dither_fast.hip)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": measured_traffic(W, H),
                         "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * npx, "kernel_ms": round(kernel_ms, 3)},
        }
        LOOKUP_BYTES = 10          # 4 B ARGB read + 2 B index + 4 B ARGB written
        if lookup_ms and lookup_ms > 0:
            la = LOOKUP_BYTES * npx / (lookup_ms * 1e-3) / 1e9
            line["roofline_lookup"] = {"bound": "hbm", "kernel": "fast_lookup_pass1_kernel + fast_lookup_pass2_kernel (MODE_LOOKUP_ONLY: per-pixel nearestColorIndex, "
                                                                 "csrc/nq_dither_fast.hip); algorithmic bytes exclude the deferred-pixel list between the passes; "
                                                                 "traffic = PMC sum over both kernels",
                                       "achieved": round(la, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(la / HBM_PEAK_GBPS, 5),
                                       "traffic": measured_traffic(W, H, LOOKUP_TRAFFIC_FILE), "algorithmic_bytes_per_launch": LOOKUP_BYTES * npx,
                                       "kernel_ms": round(lookup_ms, 4),
                                       "mpixels_s": round(npx / (lookup_ms * 1e-3) / 1e6, 1)}
        whole_ms = dt / images * 1e3 * world        # time one rank spends per image, all stages
        line["roofline_whole"] = {"bound": "hbm", "what": "8 B/pixel over the whole convert() of one image (all stages, batch amortised)",
                                  "achieved": round(BYTES_PER_PIXEL * npx / (whole_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(BYTES_PER_PIXEL * npx / (whole_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "ms_per_image": round(whole_ms, 3)}
        if world == 1 and not args.no_extras and T == 1:
            line.update(run_extras(nq, synth, slots, W, H, latency_ms, tile))
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample)
        else:
            line["cpu_baseline"] = None
        emit(line)
    if dist is not None:
        dist.destroy_process_group()


def _barrier(dist):
    torch.cuda.synchronize()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(dt, dist):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def bench_cfg4(args, nq, synth, dist, rank, local_rank, world):
    """BASELINE cfg 4: 64 frames of 1920x1080 (type (b), seeds 100 + f), LAB 256 colours + dither, frame f on rank f mod N.  One step =
    the whole batch once (every rank converts its frames in one nq_convert_batch_device call); no data-path collective."""
    from nquant.android_amd import parallel
    W, H, frames = 1920, 1080, 64
    npx = W * H
    mine = parallel.shard_frames(frames, rank, world)
    slots = []
    for f in mine:
        q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), device=local_rank, mode=nq.MODE_PARALLEL_TILED, seed=100 + f)
        q.width, q.height = W, H
        slots.append({"q": q, "in": synth.gradient_noise_torch(W, H, 100 + f), "out": torch.empty(npx, dtype=torch.int32, device="cuda"),
                      "idx": torch.empty(npx, dtype=torch.int16, device="cuda")})
    st = torch.cuda.Stream()
    slots[0]["q"].set_stream(st.cuda_stream)

    def step():
        return nq.convert_batch_device([s["q"] for s in slots], [s["in"].data_ptr() for s in slots], 256, True,
                                       [s["out"].data_ptr() for s in slots], [s["idx"].data_ptr() for s in slots])
    for _ in range(args.warmup):
        step()
    _barrier(dist)
    t0 = time.perf_counter()
    dither_ms = 0.0
    for _ in range(args.steps):
        pals = step()
        dither_ms += sum(s["q"].stage_ms()["dither"] for s in slots)
    _barrier(dist)
    dt = _max_over_ranks(time.perf_counter() - t0, dist)
    palt = torch.from_numpy(pals[0]).cuda()
    if not bool((palt[(slots[0]["idx"].to(torch.int64) & 0xFFFF)] == slots[0]["out"]).all()):
        raise SystemExit("bench cfg4: output pixels do not match palette[index]")
    if rank == 0:
        kernel_ms = dither_ms / (args.steps * len(slots))
        achieved = BYTES_PER_PIXEL * npx / (kernel_ms * 1e-3) / 1e9
        emit(({
            "metric": "Mpixels/sec, batch of 64 x 1920x1080 RGBA -> 256-colour PnnLAB + dither, frames sharded over the GPUs",
            "value": round(args.steps * frames * npx / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE cfg 4: 64 frames of 1920x1080 ARGB_8888 gradient_noise (seeds 100 + f), PnnLABQuantizer.convert(256, "
                                   "dither=true), PARALLEL_TILED automatic tiles (4x4); frame f on rank f mod N, one nq_convert_batch_device call per rank and step",
                       "frames": frames, "frames_per_rank": len(slots), "parallelism": "independent frames per GPU, no collective",
                       "collective_backend": (os.environ.get("NQ_BENCH_BACKEND", "nccl") if world > 1 else None),
                       "frames_per_s": round(args.steps * frames / dt, 2)},
            "roofline": {"bound": "hbm", "kernel": "gilbert_fast_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None, "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * npx,
                         "kernel_ms": round(kernel_ms, 4)},
            "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()


def bench_cfg5(args, nq, synth, dist, rank, local_rank, world):
    """BASELINE cfg 5: one 16384x16384 image (type (b), seed 5) in N row bands (cut at multiples of 64 rows): pre-scan all-reduce,
    all-gather of the f64 histogram partials (65536 x 5 x 8 B = 2.6 MB per rank), the palette built on every rank (the merge loop is
    a sequential chain: replicated, not sharded), every rank dithers its band.  One step = the whole image once."""
    from nquant.android_amd import parallel
    W = H = 16384
    y0, y1 = parallel.band_bounds(H, rank, world)
    rows = y1 - y0
    # the band of this rank, generated on the device from the same per-pixel stream as the whole image (offset = first pixel)
    d_band = synth.gradient_noise_torch(W, H, 5, row0=y0, rows=rows)
    d_out = torch.empty(max(rows * W, 1), dtype=torch.int32, device="cuda")
    d_idx = torch.empty(max(rows * W, 1), dtype=torch.int16, device="cuda")
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32), device=local_rank, mode=nq.MODE_PARALLEL_TILED, seed=5)
    st = torch.cuda.Stream()
    q.set_stream(st.cuda_stream)
    timings = {}

    def step():
        with torch.cuda.stream(st):
            return parallel.convert_banded(q, d_band, W, rows, y0, 256, not args.no_dither, d_out, d_idx, image_height=H, timings=timings)
    for _ in range(args.warmup):
        step()
    timings.clear()
    _barrier(dist)
    t0 = time.perf_counter()
    dither_ms = 0.0
    for _ in range(args.steps):
        pal = step()
        torch.cuda.synchronize()
        dither_ms += q.stage_ms()["dither"]
    _barrier(dist)
    dt = _max_over_ranks(time.perf_counter() - t0, dist)
    palt = torch.from_numpy(pal).cuda()
    if rows and not bool((palt[(d_idx.to(torch.int64) & 0xFFFF)] == d_out).all()):
        raise SystemExit("bench cfg5: output pixels do not match palette[index]")
    if rank == 0:
        kernel_ms = dither_ms / args.steps
        band_px = rows * W
        achieved = BYTES_PER_PIXEL * band_px / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        hist_bytes = 65536 * 5 * 8
        emit(({
            "metric": "Mpixels/sec, 16384x16384 RGBA tiled across the GPUs -> 256-colour PnnLAB%s (%s histogram exchange)"
                      % (", dither=false + BlueNoise post-pass" if args.no_dither else " + dither",
                         "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " [REHEARSAL backend, not RCCL]"),
            "value": round(args.steps * W * H / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE cfg 5: one 16384x16384 ARGB_8888 gradient_noise image (seed 5) in %d row bands of <= %d rows; per step: band "
                                   "pre-scan -> all-gather of 3 int64 -> band histogram -> all-gather of the 2.6 MB f64 partials -> palette on every rank "
                                   "(merge loop replicated) -> dither of the band with global tile indices" % (world, rows),
                       "parallelism": "row bands, one exchange step (all-gather), merge loop replicated", "collective_backend": dist.get_backend(),
                       "palette": int(len(pal))},
            "collectives": {"per_step_bytes_received_per_rank": world * (hist_bytes + 24), "histogram_partial_bytes": hist_bytes,
                            "seconds_per_step_in_collectives": round(timings.get("collectives", 0.0) / args.steps, 6),
                            "seconds_per_step_palette_build": round(timings.get("palette", 0.0) / args.steps, 6),
                            "seconds_per_step_band_dither": round(timings.get("dither", 0.0) / args.steps, 6)},
            "roofline": {"bound": "hbm", "kernel": "gilbert_fast_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None, "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * band_px,
                         "kernel_ms": round(kernel_ms, 4)},
            "cpu_baseline": None}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
