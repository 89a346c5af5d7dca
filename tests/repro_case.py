"""Re-runs one fuzz case: python tests/repro_case.py kind w h gen seed K   (prints both palettes and the scalars)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import nquant.android_amd as nq
from nquant.android_amd import synth
kind, w, h, gen, seed, K = [int(a) for a in sys.argv[1:7]]
# the fuzz draws sub-parameters from its own stream: recover them by replaying the stream is not possible here, so the few_colors
# count is passed explicitly when known; default scan over counts
for ncol in ([int(sys.argv[7])] if len(sys.argv) > 7 else range(2, 600)):
    img = synth.few_colors(w, h, seed, ncol) if gen == 2 else None
    oq = oracle_lib.OracleQuantizer(kind, img, seed=1); oq.prescan(K); want = oq.pnnquan(K)
    gq = (nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img, mode=1, seed=1); got = gq.pnnquan(K)
    if len(got) != len(want) or (got != want).any():
        po, pg = oq.params, gq.params
        print("ncol", ncol, "maxbins", po.maxbins, pg.maxbins, "ratio", po.ratio, pg.ratio, "quan_rt", po.quan_rt, pg.quan_rt, "texicab", po.texicab, pg.texicab)
        print(" want", [hex(int(x) & 0xFFFFFFFF) for x in want]); print(" got ", [hex(int(x) & 0xFFFFFFFF) for x in got])
        print(" merge", gq.merge_stats())
        break
else:
    print("no mismatch found")
