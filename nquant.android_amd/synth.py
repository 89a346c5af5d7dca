"""Deterministic synthetic ARGB_8888 inputs: splitmix64(seed + i) per pixel.  Used by bench.py and the
tests; pure numpy (inputs only -- no quantizer arithmetic lives here)."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, offset=0):
    """z_i = splitmix64 finaliser of (seed + offset + i + 1) * golden-gamma increments, i = 0..n-1 (uint64 array)."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(offset) + np.uint64(seed)) * np.uint64(1)
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _pack(a, r, g, b):
    v = (a.astype(np.uint32) << np.uint32(24)) | (r.astype(np.uint32) << np.uint32(16)) | \
        (g.astype(np.uint32) << np.uint32(8)) | b.astype(np.uint32)
    return v.view(np.int32)


def uniform_rgb(width, height, seed):
    """type (a): uniform random opaque RGB (worst case for pnnquan: every histogram bin occupied at >= 1024^2)."""
    z = splitmix64(seed, width * height)
    rgb = (z & np.uint64(0xFFFFFF)).astype(np.uint32)
    return (rgb | np.uint32(0xFF000000)).view(np.int32).reshape(height, width)


def gradient_noise(width, height, seed, noise=24, row0=0, rows=None):
    """type (b): smooth 2-D gradients + per-channel noise, opaque.  R follows x, G follows y, B follows a diagonal
    wave; `noise` is the peak-to-peak amplitude of the uniform per-channel noise (8-bit units).  With row0 / rows only the rows
    [row0, row0 + rows) of that image (large images are produced band by band)."""
    if rows is None:
        rows = height - row0
    n = width * rows
    z = splitmix64(seed, n, offset=row0 * width)
    y, x = np.divmod(np.arange(n, dtype=np.int64) + row0 * width, width)
    fx = x / max(width - 1, 1)
    fy = y / max(height - 1, 1)
    nr = ((z >> np.uint64(0)) & np.uint64(0xFF)).astype(np.float64) / 255.0 - 0.5
    ng = ((z >> np.uint64(8)) & np.uint64(0xFF)).astype(np.float64) / 255.0 - 0.5
    nb = ((z >> np.uint64(16)) & np.uint64(0xFF)).astype(np.float64) / 255.0 - 0.5
    r = 255.0 * fx + noise * nr
    g = 255.0 * fy + noise * ng
    b = 127.5 * (1.0 + np.sin(2.0 * np.pi * (0.75 * fx + 0.5 * fy))) + noise * nb
    r = np.clip(np.rint(r), 0, 255).astype(np.uint32)
    g = np.clip(np.rint(g), 0, 255).astype(np.uint32)
    b = np.clip(np.rint(b), 0, 255).astype(np.uint32)
    a = np.full(n, 255, np.uint32)
    return _pack(a, r, g, b).reshape(rows, width)


def gradient_noise_banded(width, height, seed, band_rows=1024):
    """gradient_noise() of a large image, produced band by band (the float64 temporaries of 2^28 pixels at once would need ~25 GB)."""
    out = np.empty((height, width), np.int32)
    for r0 in range(0, height, band_rows):
        rr = min(band_rows, height - r0)
        out[r0:r0 + rr] = gradient_noise(width, height, seed, row0=r0, rows=rr)
    return out


def with_alpha(img, seed, p_transparent=0.01, p_semi=0.05):
    """Variant exercising the transparency paths: ~1 % alpha == 0 pixels and ~5 % alpha in [16, 223]."""
    flat = img.reshape(-1).view(np.uint32).copy()
    z = splitmix64(seed ^ 0xA1FA, flat.size)
    u = (z & np.uint64(0xFFFF)).astype(np.float64) / 65536.0
    a = np.full(flat.size, 255, np.uint32)
    semi = u < (p_transparent + p_semi)
    a[semi] = (16 + ((z[semi] >> np.uint64(16)) % np.uint64(208))).astype(np.uint32)
    a[u < p_transparent] = 0
    out = (flat & np.uint32(0x00FFFFFF)) | (a << np.uint32(24))
    return out.view(np.int32).reshape(img.shape)


def few_colors(width, height, seed, ncolors):
    """Image drawn from `ncolors` distinct opaque colours (exercises the few-bins branches)."""
    z = splitmix64(seed, width * height)
    pal = (splitmix64(seed ^ 0x5EED, ncolors) & np.uint64(0xFFFFFF)).astype(np.uint32) | np.uint32(0xFF000000)
    return pal[(z % np.uint64(ncolors)).astype(np.int64)].view(np.int32).reshape(height, width)


def flat_with_patch(side, alpha, seed, patch=160, rgb=0x336698):
    """A flat region of more than 2^24 pixels (for side >= 4100) whose colours differ only in the low three blue bits -- ONE histogram
    bin under every key form, eight distinct colours -- plus a patch x patch gradient+noise square in the middle rows so that the image
    has more bins than colours asked for.  Row T3 of SURVEY 8a: `cnt` is a float and `cnt++` stops at 16 777 216."""
    z = splitmix64(seed, side * side)
    img = (np.uint32((alpha << 24) | rgb) | (z & np.uint64(7)).astype(np.uint32)).view(np.int32).reshape(side, side).copy()
    y0 = side // 2
    img[y0:y0 + patch, 100:100 + patch] = gradient_noise(patch, patch, seed + 1)
    return img


def tile_photo(rgb, width, height, slot=0):
    """A decoded photograph (uint8 [h][w][3], e.g. tests/golden/sample_495x438.npz) tiled to width x height opaque ARGB pixels; slot k adds
    k % 7 to the red channel (clipped) so that the images of a batch differ.  bench.py's `photo` workload and its parity test."""
    rgb = np.asarray(rgb).astype(np.uint32)
    r = np.minimum(rgb[..., 0] + np.uint32(slot % 7), np.uint32(255))
    a = ((np.uint32(255) << np.uint32(24)) | (r << np.uint32(16)) | (rgb[..., 1] << np.uint32(8)) | rgb[..., 2]).view(np.int32)
    return np.ascontiguousarray(np.tile(a, (height // a.shape[0] + 1, width // a.shape[1] + 1))[:height, :width])


def gradient_noise_torch(width, height, seed, device="cuda", noise=24, row0=0, rows=None):
    """gradient_noise() generated on the device with torch (bench.py fills a whole batch of distinct images this way): the same
    integer stream and float64 formulae; a device sin() that differs from numpy's in the last place can move a blue value that
    sits exactly on a rounding boundary, nothing else.  Returns a flat int32 tensor of width*height ARGB pixels; with row0 / rows only
    the rows [row0, row0 + rows) of that image (a rank's band of a tiled image)."""
    import torch

    def s64(v):
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(t, k):
        return (t >> k) & ((1 << (64 - k)) - 1)

    if rows is None:
        rows = height - row0
    n = width * rows
    i = torch.arange(n, dtype=torch.int64, device=device) + row0 * width
    z = i + int(seed) + s64(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    y = torch.div(i, width, rounding_mode="floor")
    x = i - y * width
    fx = x.double() / max(width - 1, 1)
    fy = y.double() / max(height - 1, 1)
    nr = (z & 0xFF).double() / 255.0 - 0.5
    ng = ((z >> 8) & 0xFF).double() / 255.0 - 0.5
    nb = ((z >> 16) & 0xFF).double() / 255.0 - 0.5
    r = torch.clamp(torch.round(255.0 * fx + noise * nr), 0, 255).long()
    g = torch.clamp(torch.round(255.0 * fy + noise * ng), 0, 255).long()
    b = torch.clamp(torch.round(127.5 * (1.0 + torch.sin(2.0 * np.pi * (0.75 * fx + 0.5 * fy))) + noise * nb), 0, 255).long()
    v = (255 << 24) | (r << 16) | (g << 8) | b
    return (v - ((v >> 31) << 32)).to(torch.int32)          # reinterpret the low 32 bits as a signed int32
