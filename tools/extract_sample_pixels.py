#!/usr/bin/env python3
"""Decode the reference's only input asset -- app/src/main/res/drawable/sample.jpg (495x438 RGB, the picture its demo
activity hands to `new PnnQuantizer(path).convert(256, true)`, app/src/main/java/nQuant/android/MainActivity.java:190-194)
-- and keep the PIXEL ARRAY as a test input: tests/golden/sample_495x438.npz {"rgb": uint8 [438][495][3]}.

DATA only (decoded pixels of an image asset, no source text).  The decoder is PIL 12.2 / libjpeg-turbo of the build
container; Android's BitmapFactory may round the IDCT / chroma upsampling differently by a unit here and there, which is
why the fixture pins the decoded array and not the file.  Run in the build container only (needs /root/reference); the
.npz is committed and travels to the GPU box."""
import pathlib
import sys

import numpy as np
from PIL import Image

src = pathlib.Path("/root/reference/app/src/main/res/drawable/sample.jpg")
out = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/sample_495x438.npz")
im = Image.open(src)
assert im.size == (495, 438) and im.mode == "RGB", (im.size, im.mode)
rgb = np.asarray(im.convert("RGB"), np.uint8)
np.savez_compressed(out, rgb=rgb)
print("wrote", out, rgb.shape, "distinct colours", len(np.unique(rgb.reshape(-1, 3), axis=0)), "bytes", out.stat().st_size)
