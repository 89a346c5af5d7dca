"""Host-side mirror of the reference's quantizer objects over the C ABI (include/nquant_abi.h).

Reference interface mirrored (NQ/ = nQuant.master/src/main/java/com/android/nQuant/):
  new PnnQuantizer(fname) / new PnnLABQuantizer(fname)   NQ/PnnQuantizer.java:35, NQ/PnnLABQuantizer.java:24
  Bitmap convert(int nMaxColors, boolean dither)          NQ/PnnQuantizer.java:409
  boolean hasAlpha()                                       NQ/PnnQuantizer.java:458
  protected pnnquan / nearestColorIndex / closestColorIndex / dither hooks
The file name argument is replaced by the decoded ARGB_8888 pixels (BitmapFactory is Android platform I/O and out
of scope); a non-zero status raises NqError just as the reference's convert() `throws Exception`."""
import ctypes as C
import os

import numpy as np

NQ_KIND_RGB, NQ_KIND_LAB = 0, 1
MODE_REFERENCE_SEQUENTIAL, MODE_PARALLEL_TILED, MODE_LOOKUP_ONLY = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABI_SYMBOLS = [
    "nq_abi_version", "nq_create", "nq_destroy", "nq_last_error", "nq_set_stream", "nq_set_tile", "nq_set_option", "nq_get_list_counts", "nq_get_params",
    "nq_set_params", "nq_convert", "nq_convert_device", "nq_convert_batch_device", "nq_convert_batch", "nq_pnnquan", "nq_pnnquan_device", "nq_dither",
    "nq_dither_device", "nq_nearest_index", "nq_closest_tuple", "nq_band_scan_device", "nq_set_scan",
    "nq_band_histogram_device", "nq_palette_from_histograms_device", "nq_band_distinct_device", "nq_set_distinct", "nq_get_stage_ms", "nq_get_merge_stats",
    "nq_get_dither_path", "nq_get_batch_phase_ms", "nq_get_team_stats", "nq_set_band", "nq_band_color_presence_device", "nq_gilbert_dither", "nq_bluenoise_dither", "nq_selftest_ciede",
]
OPT_CELL_LISTS, OPT_FAST_DITHER, OPT_MERGE_WALL_SECONDS = 1, 2, 3


def abi_symbols():
    return list(ABI_SYMBOLS)


class NqError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("nquant status %d: %s" % (status, message))
        self.status = status


class Params(C.Structure):
    """nq_params (include/nquant_abi.h)."""
    _fields_ = [("kind", C.c_int32), ("nMaxColors", C.c_int32), ("hasSemiTransparency", C.c_int32),
                ("transparentPixelIndex", C.c_int32), ("transparentColor", C.c_int32), ("isNano", C.c_int32),
                ("texicab", C.c_int32), ("quan_rt", C.c_int32), ("maxbins", C.c_int32), ("paletteLength", C.c_int32),
                ("PR", C.c_double), ("PG", C.c_double), ("PB", C.c_double), ("PA", C.c_double),
                ("ratio", C.c_double), ("weight", C.c_double), ("distinctColors", C.c_int64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def library_path():
    """libnquant_hip.so next to this file; NQ_LIB=<path> loads another build of it (kernel experiments: build.py NQ_BUILD_TAG)."""
    return os.environ.get("NQ_LIB") or os.path.join(_HERE, "libnquant_hip.so")


def _preload_hip_runtime():
    """One HIP runtime per process.  libnquant_hip.so needs `libamdhip64.so.7`; a PyTorch-ROCm wheel bundles its own copy
    (same SONAME) and always loads it by file name, so if the system runtime got in first the process would hold two
    runtimes and the second one sees no GPU.  When torch is installed, load ITS runtime first (cheap: no torch import);
    the dynamic loader then resolves our NEEDED entry and torch's to that one copy, and torch tensors / streams are
    valid in our launches.  NQ_HIP_RUNTIME=<path> overrides, NQ_HIP_RUNTIME=system skips."""
    choice = os.environ.get("NQ_HIP_RUNTIME", "")
    if choice == "system":
        return
    cand = choice
    if not cand:
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec is not None and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        except Exception:
            cand = ""
    if cand and os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Loads libnquant_hip.so (built in-tree by build.py).  Raises if it is missing: there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    _preload_hip_runtime()
    L = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.nq_abi_version.restype = i32
    L.nq_create.argtypes = [i32, i32, C.POINTER(vp)]
    L.nq_destroy.argtypes = [vp]
    L.nq_destroy.restype = None
    L.nq_last_error.argtypes = [vp]
    L.nq_last_error.restype = C.c_char_p
    L.nq_set_stream.argtypes = [vp, vp]
    L.nq_set_tile.argtypes = [vp, i32, i32]
    L.nq_set_option.argtypes = [vp, i32, i32]
    L.nq_get_list_counts.argtypes = [vp, vp, vp]
    L.nq_get_params.argtypes = [vp, C.POINTER(Params)]
    L.nq_set_params.argtypes = [vp, C.POINTER(Params)]
    L.nq_convert.argtypes = [vp, vp, i32, i32, i32, i32, i64, i32, vp, vp, vp, C.POINTER(C.c_int32)]
    L.nq_convert_device.argtypes = [vp, vp, i32, i32, i32, i32, i64, i32, vp, vp, vp, C.POINTER(C.c_int32)]
    L.nq_convert_batch_device.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, i32, vp]
    L.nq_convert_batch.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, i32, vp]
    L.nq_pnnquan.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(C.c_int32)]
    L.nq_pnnquan_device.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(C.c_int32)]
    L.nq_dither.argtypes = [vp, vp, i32, i32, vp, i32, i32, i64, i32, vp, vp]
    L.nq_dither_device.argtypes = [vp, vp, i32, i32, vp, i32, i32, i64, i32, vp, vp]
    L.nq_nearest_index.argtypes = [vp, vp, i32, vp, i64, vp]
    L.nq_closest_tuple.argtypes = [vp, vp, i32, vp, i64, vp]
    L.nq_band_scan_device.argtypes = [vp, vp, i64, i64, i32, vp]
    L.nq_set_scan.argtypes = [vp, i32, i64, C.c_uint32, i64]
    L.nq_band_histogram_device.argtypes = [vp, vp, i64, vp]
    L.nq_palette_from_histograms_device.argtypes = [vp, vp, i32, i32, vp, C.POINTER(C.c_int32)]
    L.nq_band_distinct_device.argtypes = [vp, vp, i64, i32, vp, vp]
    L.nq_set_distinct.argtypes = [vp, i64, vp]
    L.nq_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.nq_get_merge_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.nq_get_batch_phase_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.nq_get_team_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.nq_get_dither_path.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.nq_set_band.argtypes = [vp, i32, i32]
    L.nq_selftest_ciede.argtypes = [vp, vp, i64, vp]
    L.nq_gilbert_dither.argtypes = [vp, i32, i32, vp, vp, i32, vp, C.c_double, i32, i64, i32, vp, vp]
    L.nq_bluenoise_dither.argtypes = [vp, i32, i32, vp, vp, i32, vp, C.c_float, i64, i32, vp]
    L.nq_band_color_presence_device.argtypes = [vp, vp, i64, vp, i32, C.POINTER(C.c_int64), vp]
    _LIB = L
    return L


def _as_i32(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    if a.dtype != np.int32:
        raise TypeError("pixels must be int32/uint32 ARGB_8888, got %s" % a.dtype)
    return a


class QuantizedImage:
    """What convert() returns: the ARGB pixels of the reference's Bitmap plus the index map and palette."""

    def __init__(self, argb, index, palette):
        self.argb, self.index, self.palette = argb, index, palette


STAGES = ["prescan", "histogram", "nn_init", "merge", "palette_fill", "dither", "bluenoise", "total"]


class PnnQuantizer:
    """RGB PNN quantizer (NQ/PnnQuantizer.java) on one MI355X."""
    KIND = NQ_KIND_RGB

    def __init__(self, pixels, width=None, height=None, device=0, mode=MODE_PARALLEL_TILED, seed=0, tile=None):
        pixels = _as_i32(pixels)
        if width is None:
            height, width = pixels.shape
        self.width, self.height = int(width), int(height)
        self.pixels = pixels.reshape(-1)
        if self.pixels.size != self.width * self.height:
            raise ValueError("pixel count does not match width*height")
        self.mode, self.seed = mode, seed
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.nq_create(self.KIND, device, C.byref(h))
        if rc != 0:
            raise NqError(rc, (self._L.nq_last_error(None) or b"").decode())
        self._h = h
        if tile is not None:
            self._L.nq_set_tile(self._h, int(tile[0]), int(tile[1]))

    # -- plumbing --
    def close(self):
        if getattr(self, "_h", None):
            self._L.nq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise NqError(rc, (self._L.nq_last_error(self._h) or b"").decode())

    def set_stream(self, hip_stream):
        self._check(self._L.nq_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_tile(self, tile_w, tile_h):
        self._check(self._L.nq_set_tile(self._h, tile_w, tile_h))

    def set_band(self, y0, image_height):
        """The following dither calls treat their buffer as rows [y0, ...) of an image of image_height rows (0, 0 = whole image)."""
        self._check(self._L.nq_set_band(self._h, int(y0), int(image_height)))

    def list_counts(self):
        c = np.zeros(65536, np.uint8)
        n = np.zeros(65536, np.uint8)
        self._check(self._L.nq_get_list_counts(self._h, c.ctypes.data, n.ctypes.data))
        return c, n

    def set_option(self, option, value):
        self._check(self._L.nq_set_option(self._h, int(option), int(value)))

    def dither_path(self):
        """(ran the specialised dither kernel?, tiles it handed back to the generic kernel) of the last dither pass."""
        fast, failed = C.c_int32(0), C.c_int32(0)
        self._check(self._L.nq_get_dither_path(self._h, C.byref(fast), C.byref(failed)))
        return fast.value, failed.value

    @property
    def params(self):
        p = Params()
        self._check(self._L.nq_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, p):
        self._check(self._L.nq_set_params(self._h, C.byref(p)))

    def stage_ms(self):
        a = (C.c_float * 8)()
        self._check(self._L.nq_get_stage_ms(self._h, a))
        return dict(zip(STAGES, list(a)))

    def team_stats(self):
        """Counters of the last merge loop's team of helper workgroups (nq_get_team_stats)."""
        a = (C.c_int64 * 16)()
        self._check(self._L.nq_get_team_stats(self._h, a))
        return dict(zip(["published", "used", "timeouts", "wait_ticks_100MHz", "helpers", "speculating_at_end", "cache_hits_top", "virtual_merges_used",
                         "sift_ticks", "merge_ticks", "top_fetch_ticks", "pops", "epilogue_ticks", "select_ticks", "declined", "gave_up"], list(a)))

    def batch_phase_ms(self):
        """Phases of the last batch call this quantizer was the FIRST handle of: {prepare, merge, finish, total} in ms (nq_get_batch_phase_ms)."""
        a = (C.c_float * 4)()
        self._check(self._L.nq_get_batch_phase_ms(self._h, a))
        return dict(zip(["prepare", "merge", "finish", "total"], list(a)))

    def merge_stats(self):
        a = (C.c_int64 * 16)()
        self._check(self._L.nq_get_merge_stats(self._h, a))
        return dict(zip(["find_nn_calls", "merges", "find_ticks_100MHz", "ctrl_ticks_100MHz", "rebuilds", "overflows", "exact_evals",
                         "bound_ticks", "exact_ticks", "replay_ticks", "chunks", "chunks_l1", "chunks_l2", "chunks_listed", "aborted", "seed_round_ticks"], list(a)[:16]))

    # -- the reference interface --
    def hasAlpha(self):
        """NQ/PnnQuantizer.java:458-460"""
        return self.params.transparentPixelIndex > -1

    def convert(self, nMaxColors, dither, mode=None, seed=None):
        """Bitmap convert(int nMaxColors, boolean dither) (NQ/PnnQuantizer.java:409-456)."""
        n = self.width * self.height
        out = np.empty(n, np.int32)
        idx = np.empty(n, np.uint16)
        pal = np.zeros(max(int(nMaxColors), 2), np.int32)
        K = C.c_int32(0)
        self._check(self._L.nq_convert(self._h, self.pixels.ctypes.data, self.width, self.height, int(nMaxColors), int(bool(dither)),
                                       int(self.seed if seed is None else seed), int(self.mode if mode is None else mode),
                                       out.ctypes.data, idx.ctypes.data, pal.ctypes.data, C.byref(K)))
        return QuantizedImage(out.reshape(self.height, self.width), idx.reshape(self.height, self.width), pal[:K.value].copy())

    def pnnquan(self, nMaxColors):
        """Integer[] pnnquan(int[] pixels, int nMaxColors) preceded by convert()'s alpha pre-scan."""
        pal = np.zeros(max(int(nMaxColors), 2), np.int32)
        K = C.c_int32(0)
        self._check(self._L.nq_pnnquan(self._h, self.pixels.ctypes.data, self.width, self.height, int(nMaxColors),
                                       pal.ctypes.data, C.byref(K)))
        return pal[:K.value].copy()

    def dither(self, palette, dither, mode=None, seed=None):
        """int[] dither(cPixels, palette, width, height, dither) (RGB :393-407, LAB NQ/PnnLABQuantizer.java:493-522)."""
        palette = _as_i32(palette)
        n = self.width * self.height
        out = np.empty(n, np.int32)
        idx = np.empty(n, np.uint16)
        self._check(self._L.nq_dither(self._h, self.pixels.ctypes.data, self.width, self.height, palette.ctypes.data, len(palette),
                                      int(bool(dither)), int(self.seed if seed is None else seed),
                                      int(self.mode if mode is None else mode), out.ctypes.data, idx.ctypes.data))
        return out.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def gilbert_dither(self, palette, saliencies, weight, dither, mode=None, seed=None):
        """static int[] GilbertCurve.dither(width, height, pixels, palette, this, saliencies, weight, dither) (NQ/GilbertCurve.java:367):
        returns (qPixels as the reference leaves them, palette indices)."""
        palette = _as_i32(palette)
        n = self.width * self.height
        sal = None if saliencies is None else np.ascontiguousarray(saliencies, np.float32).reshape(-1)
        out = np.empty(n, np.int32)
        idx = np.empty(n, np.uint16)
        self._check(self._L.nq_gilbert_dither(self._h, self.width, self.height, self.pixels.ctypes.data, palette.ctypes.data, len(palette),
                                              None if sal is None else sal.ctypes.data, float(weight), int(bool(dither)),
                                              int(self.seed if seed is None else seed), int(self.mode if mode is None else mode),
                                              out.ctypes.data, idx.ctypes.data))
        return out.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def bluenoise_dither(self, palette, qpixels, weight, mode=None, seed=None):
        """static int[] BlueNoise.dither(width, height, pixels, palette, this, qPixels, weight) (NQ/BlueNoise.java:207): qPixels =
        palette indices in, returns (ARGB, final indices)."""
        palette = _as_i32(palette)
        n = self.width * self.height
        io = np.ascontiguousarray(qpixels, np.int32).reshape(-1).copy()
        idx = np.empty(n, np.uint16)
        self._check(self._L.nq_bluenoise_dither(self._h, self.width, self.height, self.pixels.ctypes.data, palette.ctypes.data, len(palette),
                                                io.ctypes.data, float(weight), int(self.seed if seed is None else seed),
                                                int(self.mode if mode is None else mode), idx.ctypes.data))
        return io.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def selftest_ciede(self, lab_pairs):
        """(n, 6) float32 {L1,A1,B1,L2,A2,B2} -> (n, 4) fast floats, (n, 4) literal floats (as uint32 bit patterns), (n,) decided flags."""
        a = np.ascontiguousarray(lab_pairs, np.float32).reshape(-1, 6)
        out = np.zeros((a.shape[0], 9), np.uint32)
        self._check(self._L.nq_selftest_ciede(self._h, a.ctypes.data, a.shape[0], out.ctypes.data))
        self.ciede_quad_identical = (out[:, 8] >> 1) & 1     # the quad-parallel pass (merge loop) gave the same floats and flag
        return out[:, 0:4], out[:, 4:8], out[:, 8] & 1

    def nearestColorIndex(self, palette, colors):
        """short nearestColorIndex(palette, c, pos) on a cache miss, vectorised over `colors`."""
        palette, colors = _as_i32(palette), _as_i32(colors).reshape(-1)
        out = np.empty(colors.size, np.int16)
        self._check(self._L.nq_nearest_index(self._h, palette.ctypes.data, len(palette), colors.ctypes.data, colors.size, out.ctypes.data))
        return out

    def closestTuple(self, palette, colors):
        """The closest[4] tuple closestColorIndex builds for every colour."""
        palette, colors = _as_i32(palette), _as_i32(colors).reshape(-1)
        out = np.empty((colors.size, 4), np.int32)
        self._check(self._L.nq_closest_tuple(self._h, palette.ctypes.data, len(palette), colors.ctypes.data, colors.size, out.ctypes.data))
        return out

    # -- device-pointer entry points (bench / resident pipelines): ints are HIP device addresses --
    def convert_device(self, d_pixels, nMaxColors, dither, d_out_argb, d_out_index=0, mode=None, seed=None):
        pal = np.zeros(max(int(nMaxColors), 2), np.int32)
        K = C.c_int32(0)
        self._check(self._L.nq_convert_device(self._h, C.c_void_p(d_pixels), self.width, self.height, int(nMaxColors), int(bool(dither)),
                                              int(self.seed if seed is None else seed), int(self.mode if mode is None else mode),
                                              C.c_void_p(d_out_argb), C.c_void_p(d_out_index or None), pal.ctypes.data, C.byref(K)))
        return pal[:K.value].copy()

    def pnnquan_device(self, d_pixels, nMaxColors):
        pal = np.zeros(max(int(nMaxColors), 2), np.int32)
        K = C.c_int32(0)
        self._check(self._L.nq_pnnquan_device(self._h, C.c_void_p(d_pixels), self.width, self.height, int(nMaxColors),
                                              pal.ctypes.data, C.byref(K)))
        return pal[:K.value].copy()

    def dither_device(self, d_pixels, palette, dither, d_out_argb, d_out_index=0, mode=None, seed=None):
        palette = _as_i32(palette)
        self._check(self._L.nq_dither_device(self._h, C.c_void_p(d_pixels), self.width, self.height, palette.ctypes.data, len(palette),
                                             int(bool(dither)), int(self.seed if seed is None else seed),
                                             int(self.mode if mode is None else mode), C.c_void_p(d_out_argb),
                                             C.c_void_p(d_out_index or None)))


def convert_batch_host(quantizers, pixels, nMaxColors, dither, out_argb, out_index=None, mode=None, seeds=None):
    """nq_convert_batch: like convert_batch_device with HOST addresses (ints) of the pixel / output buffers -- uploads and
    read-backs overlap the per-image stages (fully when the buffers are page-locked)."""
    return convert_batch_device(quantizers, pixels, nMaxColors, dither, out_argb, out_index, mode, seeds, _entry="nq_convert_batch")


def convert_batch_device(quantizers, d_pixels, nMaxColors, dither, d_out_argb, d_out_index=None, mode=None, seeds=None,
                         _entry="nq_convert_batch_device"):
    """nq_convert_batch_device: convert() of several quantizer objects in one call -- their merge loops run side by side in one
    launch.  `quantizers[i].width/height` describe image i, `d_pixels[i]`, `d_out_argb[i]`, `d_out_index[i]` are HIP device
    addresses.  Returns the list of palettes; results equal len(quantizers) separate convert_device calls."""
    n = len(quantizers)
    if n == 0:
        return []
    q0 = quantizers[0]
    hs = (C.c_void_p * n)(*[q._h for q in quantizers])
    src = (C.c_void_p * n)(*[int(a) for a in d_pixels])
    dst = (C.c_void_p * n)(*[int(a) for a in d_out_argb])
    idx = (C.c_void_p * n)(*[int(a) for a in d_out_index]) if d_out_index is not None else None
    widths = np.array([q.width for q in quantizers], np.int32)
    heights = np.array([q.height for q in quantizers], np.int32)
    sd = np.array([q.seed for q in quantizers] if seeds is None else list(seeds), np.int64)
    stride = max(int(nMaxColors), 2)
    pal = np.zeros((n, stride), np.int32)
    K = np.zeros(n, np.int32)
    q0._check(getattr(q0._L, _entry)(hs, n, src, widths.ctypes.data, heights.ctypes.data, int(nMaxColors), int(bool(dither)),
                                            sd.ctypes.data, int(q0.mode if mode is None else mode), dst, idx, pal.ctypes.data,
                                            stride, K.ctypes.data))
    return [pal[i, :K[i]].copy() for i in range(n)]


class PnnLABQuantizer(PnnQuantizer):
    """CIELAB PNN quantizer (NQ/PnnLABQuantizer.java) on one MI355X."""
    KIND = NQ_KIND_LAB
