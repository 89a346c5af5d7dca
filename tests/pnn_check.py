"""Palette parity (GPU vs oracle) on three medium images with the merge-loop counters: python tests/pnn_check.py  (needs a GPU)"""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle_lib as O, nquant.android_amd as nq
from nquant.android_amd import synth
for kind,K,img in ((1,256,synth.uniform_rgb(112,112,2)),(1,256,synth.gradient_noise(160,160,3)),(0,256,synth.gradient_noise(128,128,21))):
    oq=O.OracleQuantizer(kind,img); oq.prescan(K); want=oq.pnnquan(K)
    gq=(nq.PnnLABQuantizer if kind else nq.PnnQuantizer)(img)
    try:
        got=gq.pnnquan(K); print(kind,K,img.shape,"mismatch",(got!=want).sum(), gq.merge_stats(), flush=True)
    except Exception as e: print("ERR",e, gq.merge_stats(), flush=True)
