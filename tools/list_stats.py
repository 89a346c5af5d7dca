"""Pixel-weighted distribution of the candidate-list lengths (closest / nearest) on the bench image."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import nquant.android_amd as nq
from nquant.android_amd import synth
W = H = 2048
img = synth.gradient_noise(W, H, 3)
q = nq.PnnLABQuantizer(img, mode=1, seed=3)
out = q.convert(256, True)
cc, nc = q.list_counts()
px = img.reshape(-1).view(np.uint32)
cell = ((px >> 16) & 0xF8) << 8 | ((px >> 8) & 0xFC) << 3 | ((px & 0xFF) >> 3)
for name, cnt in (("closest", cc), ("nearest", nc)):
    c = np.asarray(cnt)[cell]
    h = np.bincount(c, minlength=34)
    print(name, "mean %.2f" % c.mean(), " ".join("%d:%.1f%%" % (i, 100.0 * h[i] / c.size) for i in range(len(h)) if h[i] > 0.002 * c.size))
