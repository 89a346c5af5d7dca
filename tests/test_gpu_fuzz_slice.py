"""Fixed-seed slices of the randomised GPU-vs-oracle sweep (tests/fuzz_parity.py) so that randomised coverage is part of the driver-run
suite: ~300 cases over the four modes -- std (palette + scalars + tiled dither + lookups, both kinds, 5 generators incl. alpha, K 3..1000),
fast (the specialised dither kernel and its lookups), seq (whole convert() in REFERENCE_SEQUENTIAL mode against the oracle's convert()),
big (palettes of 160..360-pixel images, up to ~60 000 bins, with the 128-thread merge workgroups forced)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,seed,cases,threads", [("std", 9101, 110, None), ("fast", 9102, 90, None), ("seq", 9103, 80, None),
                                                     ("big", 9104, 14, "128"), ("big", 9105, 8, "512")])
def test_fuzz_slice(nq, oracle, mode, seed, cases, threads, monkeypatch):
    import fuzz_parity
    if threads:
        monkeypatch.setenv("NQ_MERGE_THREADS", threads)
    lines = []
    n, bad = fuzz_parity.run(budget=600.0, seed=seed, mode=mode, max_cases=cases, log=lines.append)
    assert n == cases, "the slice stopped early: %d of %d cases" % (n, cases)
    assert bad == 0, "\n".join(l for l in lines if "MISMATCH" in l or "ERROR" in l)
