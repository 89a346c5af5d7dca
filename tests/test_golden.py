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
