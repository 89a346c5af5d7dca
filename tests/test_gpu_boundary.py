"""The drop-in boundary itself: the ditherers' static entry points (GilbertCurve.dither / BlueNoise.dither with
caller-supplied saliencies and signed weight) against the oracle's, a plain-C caller of include/nquant_abi.h (what a JNI / cgo
binding does, no Python in between), the host-buffer batch entry against the oracle, and the indexed-PNG writer on a GPU result."""
import os
import subprocess

import numpy as np
import pytest

from nquant.android_amd import synth

pytestmark = pytest.mark.gpu

SEQ, TILED = 0, 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _copy_params(src, dst_cls):
    p = dst_cls()
    for f, _ in dst_cls._fields_:
        setattr(p, f, getattr(src, f))
    return p


def _sal(shape, seed):
    z = synth.splitmix64(seed, shape[0] * shape[1])
    return (0.1 + 0.9 * (z & np.uint64(0xFFFF)).astype(np.float64) / 65535.0).astype(np.float32).reshape(shape)


STAGE_CASES = [  # kind, K, image, saliencies?, signed weight, dither, mode/tile, BlueNoise weight (None: no second stage)
    (1, 256, lambda: synth.gradient_noise(96, 64, 301), True, 0.0115, True, (8, 8), None),
    (1, 256, lambda: synth.gradient_noise(96, 64, 302), False, 0.0115, True, (8, 8), None),          # saliencies == null
    (1, 256, lambda: synth.gradient_noise(80, 64, 303), True, 0.0115, False, (8, 8), 0.95),           # indices out, then BlueNoise.dither
    (1, 64, lambda: synth.with_alpha(synth.gradient_noise(64, 64, 304), 304), True, -0.3, True, (16, 16), None),   # weight < 0: semi-transparency ladder
    (1, 16, lambda: synth.gradient_noise(64, 48, 305), True, 0.02, True, None, None),                 # sequential, K <= 32 branch
    (0, 64, lambda: synth.gradient_noise(64, 48, 306), False, 0.5, False, None, 1.0),                 # RGB sequential + BlueNoise with continued caches
    (0, 16, lambda: synth.uniform_rgb(48, 48, 307), False, 0.004, False, (8, 8), None),
    (1, 256, lambda: synth.uniform_rgb(64, 64, 308), True, 0.995, True, (4, 4), None),                # weight > .99 rung
]


@pytest.mark.parametrize("kind,K,mk,with_sal,weight,dither,tile,blue_w", STAGE_CASES)
def test_static_ditherer_entry_points_vs_oracle(nq, oracle, kind, K, mk, with_sal, weight, dither, tile, blue_w):
    img = mk()
    seed = 99
    oq = oracle.OracleQuantizer(kind, img, seed=seed)
    oq.prescan(K)
    pal = oq.pnnquan(K)
    params = _copy_params(oq.params, nq.Params)
    sal = _sal(img.shape, 1000 + K) if with_sal else None
    oq.set_seed(seed)
    want_q, want_idx = oq.gilbert_dither_stage(pal, sal, weight, dither, tile=tile)
    mode = SEQ if tile is None else TILED
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=mode, seed=seed, tile=tile)
    gq.set_params(params)
    got_q, got_idx = gq.gilbert_dither(pal, sal, weight, dither)
    assert (got_idx.astype(np.int32) == want_idx).all(), "indices: %d mismatches" % int((got_idx.astype(np.int32) != want_idx).sum())
    assert (got_q == want_q).all()
    if dither or len(pal) <= 32:
        assert (got_q == pal[got_idx]).all()             # :278-279 ARGB
    else:
        assert (got_q == got_idx).all()                  # indices
    if blue_w is not None and not dither and len(pal) > 32:
        want_argb, want_i2 = oq.bluenoise_dither_stage(pal, want_q, blue_w, tile is not None)
        got_argb, got_i2 = gq.bluenoise_dither(pal, got_q, blue_w)
        assert (got_i2.astype(np.int32) == want_i2).all()
        assert (got_argb == want_argb).all()


def test_plain_c_caller_of_the_abi(nq, oracle, tmp_path):
    """tests/c/c_abi_smoke.c, compiled with gcc against include/nquant_abi.h and libnquant_hip.so: nq_create / nq_set_tile /
    nq_convert / nq_get_params / nq_last_error / nq_destroy from C; its palette and index map are then checked against the oracle."""
    lib = nq.library_path()
    exe = str(tmp_path / "c_abi_smoke")
    torch_lib = ""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        torch_lib = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    except Exception:
        pass
    rpaths = [os.path.dirname(lib), "/opt/rocm/lib"]
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "c_abi_smoke.c"),
           "-o", exe, lib, "-Wl,--allow-shlib-undefined"] + ["-Wl,-rpath," + r for r in rpaths]
    subprocess.check_call(cmd)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = ":".join(["/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    out = str(tmp_path / "out.bin")
    r = subprocess.run([exe, out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "c_abi_smoke ok" in r.stdout
    raw = open(out, "rb").read()
    k = int(np.frombuffer(raw[:4], np.int32)[0])
    pal = np.frombuffer(raw[4:4 + 4 * k], np.int32)
    W, H = 96, 64
    idx = np.frombuffer(raw[4 + 4 * k:4 + 4 * k + 2 * W * H], np.uint16).reshape(H, W)
    img = np.frombuffer(raw[4 + 4 * k + 2 * W * H:], np.int32).reshape(H, W)
    oq = oracle.OracleQuantizer(1, img, seed=7)
    oq.prescan(256)
    want_pal = oq.pnnquan(256)
    oq.set_seed(7)
    _, want_idx = oq.dither(want_pal, True, tile=(8, 8))
    assert len(want_pal) == k and (want_pal == pal).all()
    assert (idx.astype(np.int32) == want_idx).all()


def test_convert_batch_host_buffers_vs_oracle(nq, oracle):
    """nq_convert_batch (host buffers in, host buffers out) against the oracle, image by image."""
    import torch
    imgs = [(1, synth.gradient_noise(96 + 8 * i, 80, 320 + i)) for i in range(4)] + [(0, synth.uniform_rgb(64, 72, 330))]
    qs = []
    for i, (kind, im) in enumerate(imgs):
        q = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(np.zeros((1, 1), np.int32), mode=TILED, seed=21 + i, tile=(8, 8))
        q.height, q.width = im.shape
        qs.append(q)
    h_in = [torch.from_numpy(np.ascontiguousarray(im).reshape(-1).copy()).pin_memory() for _, im in imgs]
    h_out = [torch.zeros(t.numel(), dtype=torch.int32).pin_memory() for t in h_in]
    h_idx = [torch.zeros(t.numel(), dtype=torch.int16).pin_memory() for t in h_in]
    pals = nq.convert_batch_host(qs, [t.data_ptr() for t in h_in], 256, True, [t.data_ptr() for t in h_out], [t.data_ptr() for t in h_idx])
    for i, (kind, im) in enumerate(imgs):
        oq = oracle.OracleQuantizer(kind, im, seed=21 + i)
        oq.prescan(256)
        want_pal = oq.pnnquan(256)
        want_argb, want_idx = oq.dither(want_pal, True, tile=(8, 8))
        assert (pals[i] == want_pal).all(), i
        assert (h_idx[i].numpy().view(np.uint16).reshape(im.shape).astype(np.int32) == want_idx).all(), i
        assert (h_out[i].numpy().reshape(im.shape) == want_argb).all(), i


def test_indexed_png_of_a_gpu_result_decodes_to_the_oracle_index_map(nq, oracle, tmp_path):
    """palette + u8 indices as an indexed PNG -- written from a GPU convert, decoded, compared with the ORACLE's
    index map and palette (incl. the transparent entry -> tRNS)."""
    import struct
    import zlib
    from nquant.android_amd.indexed_png import write_indexed_png
    img = synth.with_alpha(synth.gradient_noise(120, 72, 340), 340, p_transparent=0.03, p_semi=0.0)
    seed = 8
    oq = oracle.OracleQuantizer(1, img, seed=seed)
    oq.prescan(256)
    want_pal = oq.pnnquan(256)
    _, want_idx = oq.dither(want_pal, True, tile=(8, 8))
    gq = nq.PnnLABQuantizer(img, mode=TILED, seed=seed, tile=(8, 8))
    out = gq.convert(256, True)
    path = str(tmp_path / "q.png")
    write_indexed_png(path, out.index, out.palette)
    data = open(path, "rb").read()
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        chunks[tag] = chunks.get(tag, b"") + data[pos + 8:pos + 8 + n]
        pos += 12 + n
    w, h = struct.unpack(">II", chunks[b"IHDR"][:8])
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, w + 1)[:, 1:]
    assert (raw.astype(np.int32) == want_idx).all()
    plte = np.frombuffer(chunks[b"PLTE"], np.uint8).reshape(-1, 3)
    wp = want_pal.view(np.uint32)
    assert (plte == np.stack([(wp >> 16) & 0xFF, (wp >> 8) & 0xFF, wp & 0xFF], axis=1)).all()
    trns = np.frombuffer(chunks[b"tRNS"], np.uint8)
    assert len(trns) == len(wp) and (trns == ((wp >> 24) & 0xFF)).all() and (trns != 255).any()


@pytest.mark.parametrize("kind,K", [(1, 4096), (1, 8192), (0, 8192)])
def test_big_palettes_stage_through_lds_up_to_8192_entries(nq, oracle, kind, K):
    """Palettes of thousands of entries (far beyond the reference's use, but inside the documented limit of the dither pass): the
    LAB kind stages 16 bytes per entry + tables in LDS, above 64 KB from ~3700 entries on -- the launch must raise the kernel's
    dynamic-LDS limit.  Dither and pure lookups against the oracle; 8193 entries are refused with NQ_ERR_UNSUPPORTED."""
    img = synth.uniform_rgb(144, 128, 350 + K)
    seed = 3
    oq = oracle.OracleQuantizer(kind, img, seed=seed)
    oq.prescan(K)
    pal = oq.pnnquan(K)
    assert len(pal) == K
    params = _copy_params(oq.params, nq.Params)
    want_argb, want_idx = oq.dither(pal, True, tile=(8, 8))
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=TILED, seed=seed, tile=(8, 8))
    gq.set_params(params)
    got_argb, got_idx = gq.dither(pal, True)
    assert (got_idx.astype(np.int32) == want_idx).all() and (got_argb == want_argb).all()
    cols = (synth.splitmix64(K, 4000) & np.uint64(0xFFFFFF)).astype(np.uint32) | np.uint32(0xFF000000)
    assert (gq.nearestColorIndex(pal, cols.view(np.int32)) == oq.nearest_index(pal, cols.view(np.int32))).all()
    if K == 8192:
        big = np.concatenate([pal, pal[:1]])
        with pytest.raises(nq.NqError) as e:
            gq.dither(big, True)
        assert e.value.status == -3


def test_branch_free_ciede2000_equals_the_literal_functions(nq, oracle):
    """find_nn's exact phase evaluates deltaL', deltaC', deltaH', R_T in one branch-free pass (csrc/nq_device.h: ciede_terms_fast) and
    falls back to the literal functions (device-library pow / atan2 / sin / cos / exp) wherever a float narrowing or an angle
    comparison is too close to call.  Wherever the fast pass decides, its four floats must equal the literal ones bit for bit:
    3 million random pairs (uniform Lab, near-grey, near-equal hue, antipodal hue, equal colours, the Sharma table) -- and it must
    decide almost always, or it would not be worth having."""
    rng = np.random.default_rng(2024)
    n = 500_000
    def lab(n, chroma=128.0):
        return np.stack([rng.uniform(0, 100, n), rng.uniform(-chroma, chroma, n), rng.uniform(-chroma, chroma, n)], axis=1)
    sets = [np.concatenate([lab(n), lab(n)], axis=1),                                   # anything with anything
            np.concatenate([lab(n, 2.0), lab(n, 2.0)], axis=1),                         # near the grey axis
            None, None, None, None]
    a = lab(n)
    sets[2] = np.concatenate([a, a + rng.normal(0, 0.3, a.shape)], axis=1)             # close neighbours (what find_nn mostly sees)
    ang = rng.uniform(0, 2 * np.pi, n); r1 = rng.uniform(1, 120, n); r2 = rng.uniform(1, 120, n)
    sets[3] = np.stack([rng.uniform(0, 100, n), r1 * np.cos(ang), r1 * np.sin(ang),
                        rng.uniform(0, 100, n), -r2 * np.cos(ang), -r2 * np.sin(ang)], axis=1)     # antipodal hues (the 180 degree branch)
    d = rng.normal(0, 1e-6, n)
    sets[4] = np.stack([rng.uniform(0, 100, n), r1 * np.cos(ang), r1 * np.sin(ang),
                        rng.uniform(0, 100, n), r2 * np.cos(ang + d), r2 * np.sin(ang + d)], axis=1)  # nearly equal hues
    b = lab(n)
    b[:, 2] = 0.0                                                                        # B == 0 (atan2(+-0, x))
    sets[5] = np.concatenate([b, lab(n)], axis=1)
    q = nq.PnnLABQuantizer(np.zeros((1, 1), np.int32))
    decided_total = 0
    for k, s in enumerate(sets):
        fast, lit, ok = q.selftest_ciede(s.astype(np.float32))
        dec = ok == 1
        # the quad-parallel form of the pass (what the merge loop runs: csrc/nq_device.h ciede_terms_fast_quad) gives the same four
        # floats and the same decided flag as the one-lane form, pair by pair
        assert q.ciede_quad_identical.all(), "set %d: quad pass differs on %d pairs" % (k, int((q.ciede_quad_identical == 0).sum()))
        bad = (fast[dec] != lit[dec]).any(axis=1)
        assert not bad.any(), "set %d: %d of %d decided pairs differ, first %s" % (k, int(bad.sum()), int(dec.sum()), s[dec][bad][:1])
        # ... and against the ORACLE's four functions (oracle/nq_oracle.c L_prime / C_prime / H_prime / R_T with glibc's libm, restating
        # NQ/CIELABConvertor.java:91-194) on the same pairs, i.e. what find_nn really uses on the GPU (the fast floats where the pass
        # decided, the literal ones elsewhere) against the oracle's floats:
        #  * wherever the fast pass DECIDED the floats must be the oracle's bit for bit (its acceptance margins claim exactly that:
        #    any ulp-accurate libm narrows to the same float there);
        #  * where it declined, the device library's pow / atan2 / sin / cos / exp stand against glibc's.  Both are accurate to about an
        #    ulp of the double, but deltaH' = 2 sqrt(C1' C2') sin((h2' - h1') / 2) amplifies an ulp of atan2 when the two hues nearly
        #    cancel (the very reason the pass declines), so a float may move by a few ulps -- Java's own Math.atan2 is libm dependent
        #    in the same way (HotSpot intrinsic / fdlibm / bionic).  Measured on these 3 M pairs: one pair, 2 ulps in H' and R_T.
        want = oracle.ciede_terms(s.astype(np.float32)).view(np.uint32)
        used = np.where(dec[:, None], fast, lit)
        diff = used != want
        hard = diff.any(axis=1) & dec
        assert not hard.any(), "set %d: %d of %d DECIDED pairs differ from the oracle, first %s -> gpu %s oracle %s" % (
            k, int(hard.sum()), int(dec.sum()), s[hard][:1], used[hard][:1], want[hard][:1])
        soft = diff.any(axis=1) & ~dec
        if soft.any():
            # undecided pairs: H' (and R_T, which multiplies it) may differ in the last places of a SMALL number -- the error of the two
            # atan2 implementations (~1e-16 rad) over a hue difference of 1e-6 rad and less (set 4) is a relative 1e-10 of the double but
            # after the cancellation up to ~1e-3 of H' itself, on an H' of ~1e-4 that enters a sum of order 1 squared: absolute bound
            gu, wu = used[soft].view(np.float32).astype(np.float64), want[soft].view(np.float32).astype(np.float64)
            assert (np.abs(gu - wu) <= 1e-6 * np.maximum(1.0, np.abs(wu))).all(), "set %d: undecided pairs off by %g" % (k, float(np.abs(gu - wu).max()))
            assert not diff[soft][:, :2].any(), "L' and C' hold no cancelling difference: they must agree"
            if k != 4:
                assert soft.sum() <= 1e-4 * len(s), "set %d: %d undecided pairs differ from the oracle" % (k, int(soft.sum()))
        decided_total += int(dec.sum())
        if k != 4:                          # hues 1e-6 rad apart: the sine of the half difference loses all relative accuracy -> always literal
            assert dec.mean() > 0.95, (k, float(dec.mean()))
    assert decided_total > 0.8 * n * (len(sets) - 1)


def test_two_threads_two_handles_on_one_device(nq, oracle):
    """Distinct handles are independent (include/nquant_abi.h, "Threads and devices"): two host threads, each with its own handle on
    device 0 and its own stream, convert different images at the same time -- with 16x16 tiles, the configuration whose dither launch
    needs the > 64 KB dynamic-LDS attribute (set per launch on the current device, no process-wide flag) -- and both results equal the
    oracle's."""
    import threading
    import torch
    imgs = [synth.uniform_rgb(192, 160, 401), synth.uniform_rgb(176, 176, 402)]
    seeds = [17, 18]
    want = []
    for img, seed in zip(imgs, seeds):
        oq = oracle.OracleQuantizer(1, img, seed=seed)
        oq.prescan(256)
        pal = oq.pnnquan(256)
        argb, idx = oq.dither(pal, True, tile=(16, 16))
        want.append((pal, argb, idx))
    got = [None, None]
    errs = []

    def work(k):
        try:
            st = torch.cuda.Stream()
            for _ in range(3):
                q = nq.PnnLABQuantizer(imgs[k], mode=nq.MODE_PARALLEL_TILED, seed=seeds[k], tile=(16, 16))
                q.set_stream(st.cuda_stream)
                out = q.convert(256, True)
                fast = q.dither_path()[0]
                got[k] = (out, fast)
                q.close()
        except Exception as e:          # surfaced after the join
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        out, fast = got[k]
        pal, argb, idx = want[k]
        assert fast == 1, "the 16x16 case must run the specialised kernel"
        assert (out.palette == pal).all() and (out.index.astype(np.int32) == idx).all() and (out.argb == argb).all()
