"""Regression cases found by tests/fuzz_parity.py, kept as fixed tests."""
import numpy as np
import pytest

from nquant.android_amd import synth

pytestmark = pytest.mark.gpu


def test_rgb_pruned_scan_with_no_candidate_in_the_seed_blocks(nq, oracle, monkeypatch):
    """RGB merge loop, 128-thread variant, 365 bins -> 3 colours: late in the loop the positions behind the heap top are all deleted,
    the seed blocks yield no candidate and the running error is still 1e100 when the pruned blocks are walked.  The lanes without a
    block used a FINITE float sentinel (3e38 < 1e100), passed the test and indexed the block list with LDS garbage -> memory fault
    (only with dirty LDS: after a larger LAB image in the same process).  csrc/nq_merge.inc find_nn_block_rgb_boxes."""
    import oracle_lib
    monkeypatch.setenv("NQ_MERGE_THREADS", "128")
    cases = [(1, 349, 253, lambda: synth.gradient_noise(349, 253, 284438967, noise=57), 300),
             (0, 273, 324, lambda: synth.with_alpha(synth.few_colors(273, 324, 366450715, 26), 366450715), 3)]
    for kind, w, h, mk, K in cases:
        img = mk()
        oq = oracle_lib.OracleQuantizer(kind, img, seed=1)
        oq.prescan(K)
        want = oq.pnnquan(K)
        gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=nq.MODE_PARALLEL_TILED, seed=1)
        got = gq.pnnquan(K)
        assert len(got) == len(want) and (got == want).all()


@pytest.mark.parametrize("kind", [1, 0])
def test_merge_teams_with_helpers_that_never_become_resident(nq, oracle, monkeypatch, kind):
    """Merge teams (csrc/nq_merge.inc): 100 merge loops (LAB, then RGB) with 3 helper workgroups each = 400 workgroups of 512 threads for 256 CUs, so the
    helpers of the third role and part of the second are not resident while their masters run.  The masters must notice (bounded waits,
    then they stop asking that helper), finish with the oracle's palettes, and the late helpers must leave on `done`."""
    import torch
    monkeypatch.setenv("NQ_MERGE_HELPERS", "3")
    monkeypatch.setenv("NQ_MERGE_THREADS", "512")
    n = 100
    W, H = 96, 80
    imgs = [synth.gradient_noise(W, H, 700 + k) if k % 2 else synth.uniform_rgb(W, H, 700 + k) for k in range(n)]
    qs, ins, outs, idxs = [], [], [], []
    for k in range(n):
        q = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(np.zeros((1, 1), np.int32), mode=1, seed=5, tile=(8, 8))
        q.width, q.height = W, H
        qs.append(q); ins.append(torch.from_numpy(imgs[k].reshape(-1).copy()).cuda())
        outs.append(torch.zeros(W * H, dtype=torch.int32, device="cuda")); idxs.append(torch.zeros(W * H, dtype=torch.int16, device="cuda"))
    pals = nq.convert_batch_device(qs, [t.data_ptr() for t in ins], 256, True, [t.data_ptr() for t in outs], [t.data_ptr() for t in idxs])
    torch.cuda.synchronize()
    used = sum(q.team_stats()["used"] for q in qs)
    assert all(q.team_stats()["helpers"] == 3 for q in qs)
    assert used > 0, "no helper result was used at all"
    for k in (0, 1, 37, 64, 98, 99):
        oq = oracle.OracleQuantizer(kind, imgs[k], seed=5)
        oq.prescan(256)
        want = oq.pnnquan(256)
        assert len(pals[k]) == len(want) and (pals[k] == want).all(), "image %d" % k


def test_one_handle_many_palettes(nq):
    """csrc/nq_abi.cpp upload_palette: an upload is skipped when `d_palette` already holds exactly the entries asked for (in convert() the
    merge workgroup wrote them and the host read them back).  One handle must therefore keep giving, for every palette handed to
    dither(), what a fresh handle gives: same length / different entries, the same entries again, a shorter and a longer palette, and a
    convert() in between (which rewrites `d_palette` on the device)."""
    img = synth.gradient_noise(96, 72, 91)
    K = 48
    ref = nq.PnnLABQuantizer(img, mode=nq.MODE_PARALLEL_TILED, seed=4, tile=(8, 8))
    p1 = ref.pnnquan(K)
    p2 = p1.copy(); p2[5:20] = p1[5:20] ^ 0x00101010          # same length, other colours
    p3 = p1[:40].copy()
    p4 = np.concatenate([p1, p2[5:20]])

    params = ref.params

    def fresh(p, dither):
        q = nq.PnnLABQuantizer(img, mode=nq.MODE_PARALLEL_TILED, seed=4, tile=(8, 8))
        q.set_params(params)
        return q.dither(p, dither)[1]

    one = nq.PnnLABQuantizer(img, mode=nq.MODE_PARALLEL_TILED, seed=4, tile=(8, 8))
    one.set_params(params)
    for p, dither in [(p1, True), (p2, True), (p2, True), (p1, False), (p3, True), (p4, True), (p1, True)]:
        assert (one.dither(p, dither)[1] == fresh(p, dither)).all()
    got = one.convert(K, True)                                # rewrites d_palette on the device
    assert (got.palette == p1).all()
    for p in (p2, p1):
        one.set_params(params)
        assert (one.dither(p, True)[1] == fresh(p, True)).all()


@pytest.mark.parametrize("kind,alpha,throws", [(0, 0xFF, False), (1, 0x80, False), (1, 0xFF, True)])
def test_bin_count_saturates_at_two_to_the_24(nq, oracle, kind, alpha, throws):
    """Pnnbin.cnt is a float, `cnt++` (RGB NQ/PnnQuantizer.java:153) / `cnt += 1` (LAB NQ/PnnLABQuantizer.java:154)
    stops at 2^24 -- reachable from 4097^2 pixels up by any image with a flat region.  4200^2 = 17.64 M pixels, 17.61 M of them in one bin
    (csrc/nq_palette.inc compact_means_kernel clamps the exact count; hist_segments_kernel runs that bin's float32 chains 17.6 M steps
    long, the sums stalling where the float spacing exceeds the addend -- every rounding must fall as in the sequential sum).
     * RGB, opaque: sums are exact doubles, means = sum / 2^24 (too large by 5 %), (int) alpha mean 268 -> Color.argb wraps;
     * LAB, alpha 0x80 everywhere (4-4-4-4 keys): palette + scalars;
     * LAB, opaque: the alpha mean is 256 -> ColorUtils.setAlphaComponent throws in the reference; the oracle reports it and the
       GPU returns NQ_ERR_REFERENCE_THROWS."""
    img = synth.flat_with_patch(4200, alpha, 77)
    oq = oracle.OracleQuantizer(kind, img, seed=1)
    oq.prescan(256)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=nq.MODE_PARALLEL_TILED, seed=1)
    if throws:
        with pytest.raises(RuntimeError):
            oq.pnnquan(256)
        with pytest.raises(nq.NqError) as ei:
            gq.pnnquan(256)
        assert ei.value.status == -4                      # NQ_ERR_REFERENCE_THROWS
        return
    want = oq.pnnquan(256)
    got = gq.pnnquan(256)
    assert len(got) == len(want) and (got == want).all(), "%d palette entries differ" % int((got != want).sum())
    po, pg = oq.params, gq.params
    for f in ("hasSemiTransparency", "transparentPixelIndex", "isNano", "texicab", "quan_rt", "maxbins", "ratio", "weight"):
        assert getattr(po, f) == getattr(pg, f), f
    print("cnt saturation kind %d alpha %#x: maxbins %d, stage ms %s" % (kind, alpha, pg.maxbins, gq.stage_ms()))


def test_merge_time_limit_has_its_own_status_and_leaves_the_handle_usable(nq):
    """the wall-clock watchdog of the merge loop must not look like a broken heap.  65 536 bins (BASELINE cfg 2's image)
    take ~1.5 s of merge loop; with NQ_OPT_MERGE_WALL_SECONDS = 1 the call ends with NQ_ERR_TIME_LIMIT (-6), with the automatic limit the
    same handle then delivers the palette of the committed cfg 2 fixture."""
    import os
    img = synth.uniform_rgb(1024, 1024, 2)
    want = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_lab256_lookup_uniform_1024x1024.npz"))
    q = nq.PnnLABQuantizer(img, mode=nq.MODE_PARALLEL_TILED, seed=1)
    q.set_option(3, 1)                                  # NQ_OPT_MERGE_WALL_SECONDS
    with pytest.raises(nq.NqError) as ei:
        q.pnnquan(256)
    assert ei.value.status == -6 and "time limit" in str(ei.value)
    q.set_option(3, 0)                                  # automatic again
    pal = q.pnnquan(256)
    assert len(pal) == len(want["palette"]) and (pal == want["palette"]).all()
