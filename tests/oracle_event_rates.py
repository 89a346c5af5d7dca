"""Event rates of the per-pixel pass (CPU oracle, tiled decomposition): how often closestColorIndex falls back to
nearestColorIndex, how often the error limiter fires, ... -- the numbers the dither kernel's fast paths are sized for.
Usage: python tests/oracle_event_rates.py [size] [tile]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import oracle_lib
from nquant.android_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# the bench image at reduced size keeps its per-pixel gradient only if it is a crop: take the top-left corner of the 4096^2 image
full = 4096
img = synth.gradient_noise(full, full, 3)[:size, :size].copy() if size < full else synth.gradient_noise(full, full, 3)
L = oracle_lib.lib()
L.nqo_debug_counters.argtypes = [C.c_void_p, C.c_int]
q = oracle_lib.OracleQuantizer(1, img, seed=3)
t = time.time(); q.prescan(256); pal = q.pnnquan(256); print("pnnquan %.1fs K=%d maxbins=%d weight=%.5f ratio=%.5f" % (time.time() - t, len(pal), q.params.maxbins, q.params.weight, q.params.ratio))
L.nqo_debug_counters(None, 1)
t = time.time(); q.dither(pal, True, tile=(tile, tile)); print("dither %.1fs" % (time.time() - t))
c = (C.c_int64 * 16)(); L.nqo_debug_counters(c, 0)
n = c[0]
names = {1: "closest calls", 2: "closest -> nearest fallback", 3: "limiter fires", 4: "ditherPixel 2nd stage", 5: "sal>.95 stage", 6: "direct (sal>.99)",
         7: "closest[2]==0", 8: "closest[2]>=K", 9: "limiter with tanh", 10: "maxErr raised",
         11: "lookup colour in the 5-6-5 cell of the undithered colour (ditherPixel calls)"}
for k, v in names.items():
    print("%-32s %10d  %.4f" % (v, c[k], c[k] / max(n, 1)))
