"""Randomised GPU-vs-oracle parity sweep (palette + scalars, tiled dither, lookups) over image kinds, sizes, K and flags.
python tests/fuzz_parity.py [seconds] [seed] [big|seq|fast]  -- prints every mismatch and a summary; exit code 1 on any mismatch.
fast: the specialised dither kernel (csrc/nq_dither_fast.hip) -- LAB, 33 <= K <= 256, opaque or alpha-0 images, `weight` injected into
both sides from (.0026, .0149) (the DITHER_MAX = 25 rung needs > 17 000 bins, i.e. large images, by itself), random tiles incl. odd ones,
dither on / off; output, lookups and the tiles handed back are compared with the oracle and the generic kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import nquant.android_amd as nq
from nquant.android_amd import synth

Ks = [3, 4, 5, 6, 7, 8, 12, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 1000]


def run(budget=300.0, seed=1, mode="std", max_cases=None, log=print):
    """One sweep: mode std | big | fast | seq, until `budget` seconds have passed or `max_cases` cases have run.  Returns (cases,
    mismatches).  tests/test_gpu_fuzz_slice.py runs fixed-seed slices of every mode under the driver."""
    BIG, FAST, SEQ = mode == "big", mode == "fast", mode == "seq"
    # big: 160..360 pixels a side (up to ~60k bins), palette + scalars only; seq: whole convert() in REFERENCE_SEQUENTIAL mode against the oracle's convert()
    rng = np.random.default_rng(int(seed))
    t_end = time.time() + budget
    n_cases = n_bad = 0

    def print(*a, **k):                    # every report line goes through `log`
        log(" ".join(str(x) for x in a))
    while time.time() < t_end and (max_cases is None or n_cases < max_cases):
        kind = int(rng.integers(0, 2))
        w, h = (int(rng.integers(160, 360)), int(rng.integers(160, 360))) if BIG else (int(rng.integers(17, 150)), int(rng.integers(17, 150)))
        if SEQ:
            w, h = int(rng.integers(9, 90)), int(rng.integers(9, 90))
        seed = int(rng.integers(1, 1 << 30))
        gen = int(rng.integers(0, 5))
        if gen == 0: img = synth.uniform_rgb(w, h, seed)
        elif gen == 1: img = synth.gradient_noise(w, h, seed, noise=int(rng.integers(0, 64)))
        elif gen == 2: img = synth.few_colors(w, h, seed, int(rng.integers(2, 600)))
        elif gen == 3: img = synth.with_alpha(synth.gradient_noise(w, h, seed), seed, p_transparent=float(rng.random() * 0.05), p_semi=float(rng.random() * 0.2))
        else: img = synth.with_alpha(synth.few_colors(w, h, seed, int(rng.integers(2, 300))), seed)
        K = int(Ks[rng.integers(0, len(Ks))])
        dither = bool(rng.integers(0, 2))
        tile = (int(rng.choice([4, 8, 16])),) * 2
        rseed = int(rng.integers(0, 1 << 20))
        tag = "kind %d %dx%d gen %d seed %d K %d dither %d tile %d" % (kind, w, h, gen, seed, K, dither, tile[0])
        try:
            if FAST:
                K = int(rng.integers(33, 257))
                w, h = int(rng.integers(24, 140)), int(rng.integers(24, 140))
                g2 = int(rng.integers(0, 4))
                img = synth.uniform_rgb(w, h, seed) if g2 == 0 else synth.gradient_noise(w, h, seed, noise=int(rng.integers(0, 64)))
                if g2 == 3:
                    img = synth.with_alpha(img, seed, p_transparent=float(rng.random() * 0.05), p_semi=0.0)
                tile = [(4, 4), (8, 8), (16, 16), (7, 5), (8, 4), (12, 12)][int(rng.integers(0, 6))]
                weight = float(rng.uniform(0.0026, 0.0149))
                tag = "FAST %dx%d gen %d seed %d K %d dither %d tile %s weight %.6f rseed %d" % (w, h, g2, seed, K, dither, tile, weight, rseed)
                oq = oracle_lib.OracleQuantizer(1, img, seed=rseed)
                oq.prescan(K)
                pal = oq.pnnquan(K)
                if len(pal) <= 32:
                    continue
                op = oq.params
                op.weight = weight; op.isNano = 1
                oq.set_params(op)
                gp = nq.Params()
                for f, _ in nq.Params._fields_:
                    setattr(gp, f, getattr(op, f))
                want_argb, want_idx = oq.dither(pal, dither, tile=tile)
                n_cases += 1
                for fast in (0, 1):          # (the lookups below then go through the specialised kernels)
                    gq = nq.PnnLABQuantizer(img, mode=nq.MODE_PARALLEL_TILED, seed=rseed, tile=tile)
                    gq.set_params(gp)
                    gq.set_option(2, fast)
                    got_argb, got_idx = gq.dither(pal, dither)
                    ran, back = gq.dither_path()
                    if ran != (fast if op.ratio >= 0 else 0):
                        n_bad += 1; print("PATH MISMATCH: fast=%d ran=%d" % (fast, ran), tag, flush=True)
                    if (got_idx.astype(np.int32) != want_idx).any() or (got_argb != want_argb).any():
                        n_bad += 1; print("FAST DITHER MISMATCH (fast=%d, handed back %d): %d px" % (fast, back, int((got_idx.astype(np.int32) != want_idx).sum())), tag, flush=True)
                cols = (synth.splitmix64(seed, 8192) & np.uint64(0xFFFFFF)).astype(np.uint32) | np.uint32(0xFF000000)
                cols = cols.view(np.int32)
                if (gq.nearestColorIndex(pal, cols) != oq.nearest_index(pal, cols)).any():
                    n_bad += 1; print("FAST NEAREST MISMATCH:", tag, flush=True)
                if (gq.closestTuple(pal, cols) != oq.closest_tuple(pal, cols)).any():
                    n_bad += 1; print("FAST CLOSEST MISMATCH:", tag, flush=True)
                if n_cases % 25 == 0:
                    print("... %d cases, %d mismatches" % (n_cases, n_bad), flush=True)
                continue
            if SEQ:
                Ks2 = K if rng.random() < 0.9 else int(rng.integers(1, 3))
                oq = oracle_lib.OracleQuantizer(kind, img, seed=rseed)
                try:
                    want_argb, want_idx, want_pal = oq.convert(Ks2, dither)
                except RuntimeError:
                    continue                                  # the Java code would throw (setAlphaComponent)
                gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=nq.MODE_REFERENCE_SEQUENTIAL, seed=rseed)
                out = gq.convert(Ks2, dither)
                n_cases += 1
                if len(out.palette) != len(want_pal) or (out.palette != want_pal).any() or (out.argb != want_argb).any() \
                        or (out.index.astype(np.int32) != want_idx).any():
                    n_bad += 1; print("SEQ CONVERT MISMATCH:", tag, "K used", Ks2, flush=True)
                if n_cases % 25 == 0:
                    print("... %d cases, %d mismatches" % (n_cases, n_bad), flush=True)
                continue
            oq = oracle_lib.OracleQuantizer(kind, img, seed=rseed)
            oq.prescan(K)
            want_pal = oq.pnnquan(K)
            gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=nq.MODE_PARALLEL_TILED, seed=rseed, tile=tile)
            got_pal = gq.pnnquan(K)
            n_cases += 1
            if len(got_pal) != len(want_pal) or (got_pal != want_pal).any():
                n_bad += 1; print("PALETTE MISMATCH:", tag, flush=True); continue
            po, pg = oq.params, gq.params
            for f in ("hasSemiTransparency", "transparentPixelIndex", "transparentColor", "isNano", "texicab", "quan_rt", "maxbins",
                      "PR", "PG", "PB", "PA", "ratio", "weight"):
                if getattr(po, f) != getattr(pg, f):
                    n_bad += 1; print("PARAM MISMATCH:", f, tag, flush=True)
            if BIG:
                print("ok", tag, "maxbins", po.maxbins, flush=True)
                continue
            oq.set_seed(rseed)
            want_argb, want_idx = oq.dither(want_pal, dither, tile=tile)
            if kind == 1 and not dither and len(want_pal) > 32:
                p = gq.params; p.distinctColors = oq.params.distinctColors; gq.set_params(p)
            got_argb, got_idx = gq.dither(got_pal, dither)
            if (got_idx.astype(np.int32) != want_idx).any() or (got_argb != want_argb).any():
                n_bad += 1; print("DITHER MISMATCH: %d px" % int((got_idx.astype(np.int32) != want_idx).sum()), tag, flush=True)
            cols = img.reshape(-1)[: 4096]
            if (gq.nearestColorIndex(got_pal, cols) != oq.nearest_index(want_pal, cols)).any():
                n_bad += 1; print("NEAREST MISMATCH:", tag, flush=True)
            if (gq.closestTuple(got_pal, cols) != oq.closest_tuple(want_pal, cols)).any():
                n_bad += 1; print("CLOSEST MISMATCH:", tag, flush=True)
        except nq.NqError as e:
            if e.status in (-3, -4):       # UNSUPPORTED / REFERENCE_THROWS are legitimate outcomes
                continue
            n_bad += 1; print("ERROR:", e, tag, flush=True)
        if n_cases % 25 == 0:
            print("... %d cases, %d mismatches" % (n_cases, n_bad), flush=True)
    log("fuzz: %d cases, %d mismatches" % (n_cases, n_bad))
    return n_cases, n_bad


if __name__ == "__main__":
    import builtins
    _mode = sys.argv[3] if len(sys.argv) > 3 else "std"
    _n, _bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, _mode,
                   log=lambda m: builtins.print(m, flush=True))
    sys.exit(1 if _bad else 0)
