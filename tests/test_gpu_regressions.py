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
