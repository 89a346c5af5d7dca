"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
 * CPU: the oracle still reproduces them (guards the restatement against regressions);
 * GPU: the HIP path reproduces them through the C ABI."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden(name):
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.run_case(mg.CASES[name])
    for k in ("palette", "index", "argb", "scalars", "doubles"):
        assert (got[k] == want[k]).all(), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_gpu_reproduces_golden(nq, name):
    c = mg.CASES[name]
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    img = c["img"]()
    mode = nq.MODE_REFERENCE_SEQUENTIAL if c["tile"] is None else nq.MODE_PARALLEL_TILED
    q = (nq.PnnLABQuantizer if c["kind"] else nq.PnnQuantizer)(img, mode=mode, seed=c["seed"], tile=c["tile"])
    pal = q.pnnquan(c["K"])
    assert (pal == want["palette"]).all()
    p = q.params
    if int(want["distinct"]) and c["kind"] == 1 and not c["dither"]:
        p.distinctColors = int(want["distinct"])
        q.set_params(p)
    argb, idx = q.dither(pal, c["dither"])
    assert (idx == want["index"]).all()
    assert (argb == want["argb"]).all()


@pytest.mark.parametrize("name", sorted(mg.PALETTE_CASES))
def test_oracle_reproduces_golden_palette(name):
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.run_palette_case(mg.PALETTE_CASES[name])
    for k in ("palette", "scalars", "doubles"):
        assert (got[k] == want[k]).all(), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [512, 256, 128, 127])      # 127 = the dense 128-thread variant (six workgroups per CU)
@pytest.mark.parametrize("name", sorted(mg.PALETTE_CASES))
def test_gpu_reproduces_golden_palette_every_merge_variant(nq, name, threads, monkeypatch):
    """~64k bins: beyond the LDS mirrors of every merge-workgroup variant (csrc/nq_merge.inc), all 1005 position blocks in use."""
    monkeypatch.setenv("NQ_MERGE_THREADS", str(threads))
    c = mg.PALETTE_CASES[name]
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    q = (nq.PnnLABQuantizer if c["kind"] else nq.PnnQuantizer)(c["img"](), mode=nq.MODE_PARALLEL_TILED, seed=1)
    pal = q.pnnquan(c["K"])
    assert len(pal) == len(want["palette"]) and (pal == want["palette"]).all()
    p = q.params
    assert [p.maxbins, p.isNano, p.texicab, p.quan_rt] == list(want["scalars"])
    assert (np.array([p.ratio, p.weight]) == want["doubles"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("theta", ["1.0", "1.25", "64"])
@pytest.mark.parametrize("name", sorted(n for n in mg.PALETTE_CASES if mg.PALETTE_CASES[n]["kind"] == 0))
def test_gpu_rgb_palette_does_not_depend_on_the_assumed_error_cap(nq, name, theta, monkeypatch):
    """The RGB scans prune blocks against an ASSUMED cap of the running error (theta x the error after the seed blocks) and fall back
    to the unpruned scan when the assumption fails (csrc/nq_merge.inc find_nn_block_rgb_boxes): theta 1.0 fails whenever the error
    rises (NQ/PnnQuantizer.java:97-113 `break` then err = nerr), 64 prunes next to nothing; the palette must not move."""
    monkeypatch.setenv("NQ_RGB_THETA", theta)
    c = mg.PALETTE_CASES[name]
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    q = nq.PnnQuantizer(c["img"](), mode=nq.MODE_PARALLEL_TILED, seed=1)
    pal = q.pnnquan(c["K"])
    assert len(pal) == len(want["palette"]) and (pal == want["palette"]).all()
    st = q.merge_stats()
    assert st["chunks"] > 0                       # the pruned scan ran
    if theta == "1.0":
        assert st["overflows"] > 0                # ... and its fallback


@pytest.mark.parametrize("name", sorted(mg.LOOKUP_CASES))
def test_oracle_reproduces_golden_lookup(name):
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.run_lookup_case(mg.LOOKUP_CASES[name])
    for k in ("palette", "scalars", "doubles", "index_sha256", "index_sample", "index_histogram"):
        assert (got[k] == want[k]).all(), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.LOOKUP_CASES))
def test_gpu_cfg2_end_to_end_lookup_only(nq, name):
    """BASELINE cfg 2 end to end through nq_convert(..., LOOKUP_ONLY): the image's own 65 536-bin pnnquan (palette + scalars == golden) and
    the undithered index map (SHA-256, sample and per-entry counts == the oracle's per-pixel nearestColorIndex held by the fixture)."""
    import hashlib
    c = mg.LOOKUP_CASES[name]
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    img = c["img"]()
    q = (nq.PnnLABQuantizer if c["kind"] else nq.PnnQuantizer)(img, mode=nq.MODE_LOOKUP_ONLY, seed=1)
    out = q.convert(c["K"], False)
    assert len(out.palette) == len(want["palette"]) and (out.palette == want["palette"]).all()
    p = q.params
    assert [p.maxbins, p.isNano, p.texicab, p.quan_rt] == list(want["scalars"])
    assert (np.array([p.ratio, p.weight]) == want["doubles"]).all()
    idx = out.index.reshape(-1).astype(np.uint16)
    assert (idx[::c["stride"]] == want["index_sample"]).all(), "%d of the sampled indices differ" % int((idx[::c["stride"]] != want["index_sample"]).sum())
    assert (np.bincount(idx, minlength=len(out.palette)) == want["index_histogram"]).all()
    assert (np.frombuffer(hashlib.sha256(idx.tobytes()).digest(), np.uint8) == want["index_sha256"]).all()
    assert (out.argb.reshape(-1) == out.palette[idx]).all()


# ---- the reference's own input asset (a photograph): tests/golden/sample_*.npz ----

@pytest.mark.parametrize("name", sorted(mg.SAMPLE_CASES))
def test_oracle_reproduces_golden_sample(name):
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.run_sample_case(mg.SAMPLE_CASES[name])
    for k in want.files:
        assert (got[k] == want[k]).all(), (name, k)


def _sample_quantizer(nq, c, mode, tile=None):
    return (nq.PnnLABQuantizer if c["kind"] else nq.PnnQuantizer)(mg.sample_image(), mode=mode, seed=c["seed"], tile=tile)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.SAMPLE_CASES))
def test_gpu_sample_photo_whole_convert_sequential(nq, name):
    """The app's real call on the app's real picture -- new PnnQuantizer(path).convert(256, true) (MainActivity.java:190-194), the
    README's PnnLABQuantizer, their dither=false legs and a 16-colour LAB run -- as ONE convert() in REFERENCE_SEQUENTIAL mode (one
    curve over the image, colour-keyed first-come caches: weight 0.086 > .015, one Random(seed)): palette, scalars, every index and
    the ARGB bitmap equal the fixture."""
    c = mg.SAMPLE_CASES[name]
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    q = _sample_quantizer(nq, c, nq.MODE_REFERENCE_SEQUENTIAL)
    out = q.convert(c["K"], c["dither"])
    assert len(out.palette) == len(want["palette"]) and (out.palette == want["palette"]).all()
    p = q.params
    assert [p.hasSemiTransparency, p.transparentPixelIndex, p.transparentColor, p.isNano, p.texicab, p.quan_rt, p.maxbins,
            p.paletteLength] == list(want["scalars"])
    assert (np.array([p.PR, p.PG, p.PB, p.PA, p.ratio, p.weight]) == want["doubles"]).all()
    nbad = int((out.index != want["seq_index"]).sum())
    assert nbad == 0, "%d of %d indices differ" % (nbad, out.index.size)
    assert (mg._sha(out.argb) == want["seq_argb_sha256"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.SAMPLE_CASES))
def test_gpu_sample_photo_tiled(nq, oracle, name):
    """The same calls in the production mode (PARALLEL_TILED): 8x8 tiles, ragged at the right and bottom edges (495 x 438), against
    the oracle's tiled restatement held by the fixture (256 colours: the sorted-by-yDiff queue, whose tile chains start in the queue's
    steady state); and the AUTOMATIC tile rule (4x4 at this size, for every queue form) against the oracle run here."""
    c = mg.SAMPLE_CASES[name]
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    q = _sample_quantizer(nq, c, nq.MODE_PARALLEL_TILED, tile=mg.SAMPLE_TILE)
    out = q.convert(c["K"], c["dither"])
    assert (out.palette == want["palette"]).all()
    nbad = int((out.index != want["tiled_index"]).sum())
    assert nbad == 0, "%d of %d indices differ" % (nbad, out.index.size)
    assert (mg._sha(out.argb) == want["tiled_argb_sha256"]).all()
    img = mg.sample_image()
    oq = oracle.OracleQuantizer(c["kind"], img, seed=c["seed"])
    oq.prescan(c["K"])
    pal = oq.pnnquan(c["K"])
    oq.set_seed(c["seed"])
    want_argb, want_idx = oq.dither(pal, c["dither"], tile=(4, 4))
    q2 = _sample_quantizer(nq, c, nq.MODE_PARALLEL_TILED)
    out2 = q2.convert(c["K"], c["dither"])
    assert (out2.index.astype(np.int32) == want_idx).all() and (out2.argb == want_argb).all()
