"""ctypes binding of the CPU oracle (oracle/libnq_oracle.so).  Test infrastructure only: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libnq_oracle.so")


class Params(C.Structure):
    """nqo_params / nq_params (identical layout)."""
    _fields_ = [("kind", C.c_int32), ("nMaxColors", C.c_int32), ("hasSemiTransparency", C.c_int32),
                ("transparentPixelIndex", C.c_int32), ("transparentColor", C.c_int32), ("isNano", C.c_int32),
                ("texicab", C.c_int32), ("quan_rt", C.c_int32), ("maxbins", C.c_int32), ("paletteLength", C.c_int32),
                ("PR", C.c_double), ("PG", C.c_double), ("PB", C.c_double), ("PA", C.c_double),
                ("ratio", C.c_double), ("weight", C.c_double), ("distinctColors", C.c_int64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def build_oracle():
    src = os.path.join(ORACLE_DIR, "nq_oracle.c")
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(LIB_PATH)
    p32 = C.POINTER(C.c_int32)
    L.nqo_create.restype = C.c_void_p
    L.nqo_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.nqo_destroy.argtypes = [C.c_void_p]
    L.nqo_set_seed.argtypes = [C.c_void_p, C.c_int64]
    L.nqo_gilbert_dither_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.nqo_bluenoise_dither_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    L.nqo_set_bands.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.nqo_get_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.nqo_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.nqo_convert.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, p32]
    L.nqo_prescan.argtypes = [C.c_void_p, C.c_int]
    L.nqo_pnnquan.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.nqo_dither.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.nqo_dither_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.nqo_dither_tile_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.nqo_nearest_index.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    L.nqo_closest_tuple.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    L.nqo_get_color_index.restype = C.c_int32
    L.nqo_get_color_index.argtypes = [C.c_int32, C.c_int, C.c_int]
    L.nqo_rgb2lab.argtypes = [C.c_int32, C.POINTER(C.c_float)]
    L.nqo_lab2rgb.restype = C.c_int32
    L.nqo_lab2rgb.argtypes = [C.c_float] * 4
    L.nqo_ciede2000.restype = C.c_float
    L.nqo_ciede2000.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.nqo_ciede_terms.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.nqo_ciede_terms.restype = None
    L.nqo_debug_virtual_merge.argtypes = [C.c_int, C.c_void_p]
    L.nqo_debug_virtual_merge.restype = None
    L.nqo_y_diff.restype = C.c_double
    L.nqo_y_diff.argtypes = [C.c_int32, C.c_int32]
    L.nqo_u_diff.restype = C.c_double
    L.nqo_u_diff.argtypes = [C.c_int32, C.c_int32]
    L.nqo_blue_diffuse.restype = C.c_int32
    L.nqo_blue_diffuse.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int, C.c_int]
    L.nqo_blue_noise.restype = C.c_int8
    L.nqo_blue_noise.argtypes = [C.c_int]
    L.nqo_gilbert_path.restype = C.c_int64
    L.nqo_gilbert_path.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.nqo_gilbert_params.restype = C.c_float
    L.nqo_gilbert_params.argtypes = [C.c_int, C.c_double, C.c_int, p32]
    L.nqo_jrandom_seed.argtypes = [C.POINTER(C.c_int64), C.c_int64]
    L.nqo_jrandom_next_int.restype = C.c_int32
    L.nqo_jrandom_next_int.argtypes = [C.POINTER(C.c_int64)]
    L.nqo_jrandom_next_int_bound.restype = C.c_int32
    L.nqo_jrandom_next_int_bound.argtypes = [C.POINTER(C.c_int64), C.c_int32]
    L.nqo_get_stage_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.nqo_get_find_nn_calls.restype = C.c_int64
    L.nqo_get_find_nn_calls.argtypes = [C.c_void_p]
    _lib = L
    return L


def _i32(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    assert a.dtype == np.int32, a.dtype
    return a


class OracleQuantizer:
    """Mirrors `new PnnQuantizer(fname)` / `new PnnLABQuantizer(fname)` with the pixels handed in decoded."""
    RGB, LAB = 0, 1

    def __init__(self, kind, argb, width=None, height=None, seed=0):
        argb = _i32(argb)
        if width is None:
            height, width = argb.shape
        self.width, self.height, self.kind = int(width), int(height), int(kind)
        self._L = lib()
        self._h = self._L.nqo_create(self.kind, argb.ctypes.data, self.width, self.height)
        self._L.nqo_set_seed(self._h, seed)

    def close(self):
        if self._h:
            self._L.nqo_destroy(self._h)
            self._h = None

    __del__ = close

    def set_seed(self, seed):
        self._L.nqo_set_seed(self._h, seed)

    def set_bands(self, row_starts):
        """Banded restatement of the LAB histogram: float sums restart at these rows, partials added in band order."""
        a = np.asarray(list(row_starts), np.int32)
        self._L.nqo_set_bands(self._h, len(a), a.ctypes.data)

    @property
    def params(self):
        p = Params()
        self._L.nqo_get_params(self._h, C.byref(p))
        return p

    def set_params(self, p):
        self._L.nqo_set_params(self._h, C.byref(p))

    def convert(self, nMaxColors, dither):
        n = self.width * self.height
        out = np.zeros(n, np.int32)
        idx = np.zeros(n, np.int32)
        pal = np.zeros(max(nMaxColors, 2), np.int32)
        K = C.c_int32(0)
        rc = self._L.nqo_convert(self._h, nMaxColors, int(dither), out.ctypes.data, idx.ctypes.data, pal.ctypes.data, C.byref(K))
        if rc != 0:
            raise RuntimeError("oracle convert failed (the Java code would throw)")
        return out.reshape(self.height, self.width), idx.reshape(self.height, self.width), pal[:K.value].copy()

    def prescan(self, nMaxColors):
        self._L.nqo_prescan(self._h, nMaxColors)

    def pnnquan(self, nMaxColors):
        pal = np.zeros(max(nMaxColors, 2), np.int32)
        k = self._L.nqo_pnnquan(self._h, nMaxColors, pal.ctypes.data)
        if k < 0:
            raise RuntimeError("oracle pnnquan failed")
        return pal[:k].copy()

    def dither(self, palette, dither, tile=None):
        palette = _i32(palette)
        n = self.width * self.height
        out = np.zeros(n, np.int32)
        idx = np.zeros(n, np.int32)
        if tile is None:
            self._L.nqo_dither(self._h, palette.ctypes.data, len(palette), int(dither), out.ctypes.data, idx.ctypes.data)
        else:
            self._L.nqo_dither_tiled(self._h, palette.ctypes.data, len(palette), int(dither), int(tile[0]), int(tile[1]),
                                     out.ctypes.data, idx.ctypes.data)
        return out.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def dither_tile_rows(self, palette, dither, tile, row_first, row_count):
        """Tiled dither of the tile rows [row_first, row_first + row_count) only (tiles are independent chains)."""
        palette = _i32(palette)
        n = self.width * self.height
        out = np.zeros(n, np.int32)
        idx = np.zeros(n, np.int32)
        self._L.nqo_dither_tile_rows(self._h, palette.ctypes.data, len(palette), int(dither), int(tile[0]), int(tile[1]),
                                     int(row_first), int(row_count), out.ctypes.data, idx.ctypes.data)
        return out.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def gilbert_dither_stage(self, palette, saliencies, weight, dither, tile=None):
        palette = _i32(palette)
        n = self.width * self.height
        sal = None if saliencies is None else np.ascontiguousarray(saliencies, np.float32).reshape(-1)
        out = np.zeros(n, np.int32); idx = np.zeros(n, np.int32)
        tw, th = (0, 0) if tile is None else tile
        self._L.nqo_gilbert_dither_stage(self._h, palette.ctypes.data, len(palette), None if sal is None else sal.ctypes.data, float(weight),
                                         int(dither), int(tw), int(th), out.ctypes.data, idx.ctypes.data)
        return out.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def bluenoise_dither_stage(self, palette, qpixels, weight, tiled):
        palette = _i32(palette)
        io = np.ascontiguousarray(qpixels, np.int32).reshape(-1).copy()
        idx = np.zeros(io.size, np.int32)
        self._L.nqo_bluenoise_dither_stage(self._h, palette.ctypes.data, len(palette), io.ctypes.data, float(weight), int(tiled), idx.ctypes.data)
        return io.reshape(self.height, self.width), idx.reshape(self.height, self.width)

    def nearest_index(self, palette, colors):
        palette, colors = _i32(palette), _i32(colors).ravel()
        out = np.zeros(colors.size, np.int16)
        self._L.nqo_nearest_index(self._h, palette.ctypes.data, len(palette), colors.ctypes.data, colors.size, out.ctypes.data)
        return out

    def closest_tuple(self, palette, colors):
        palette, colors = _i32(palette), _i32(colors).ravel()
        out = np.zeros((colors.size, 4), np.int32)
        self._L.nqo_closest_tuple(self._h, palette.ctypes.data, len(palette), colors.ctypes.data, colors.size, out.ctypes.data)
        return out

    def stage_seconds(self):
        a = (C.c_double * 6)()
        self._L.nqo_get_stage_seconds(self._h, a)
        return dict(zip(["prescan", "histogram", "nn_init", "merge", "gilbert", "bluenoise"], list(a)))

    def find_nn_calls(self):
        return self._L.nqo_get_find_nn_calls(self._h)


def gilbert_path(w, h):
    xy = np.zeros((w * h, 2), np.int32)
    n = lib().nqo_gilbert_path(w, h, xy.ctypes.data)
    assert n == w * h
    return xy


def rgb2lab(c):
    o = (C.c_float * 4)()
    lib().nqo_rgb2lab(np.int32(np.uint32(c).view(np.int32) if isinstance(c, np.uint32) else _wrap(c)), o)
    return tuple(o)  # alpha, L, A, B


def _wrap(c):
    c = int(c) & 0xFFFFFFFF
    return c - (1 << 32) if c >= (1 << 31) else c


def ciede_terms(pairs):
    """(n, 6) float32 {L1,A1,B1,L2,A2,B2} -> (n, 4) float32 {L', C', H', R_T} from the oracle's four CIEDE2000 functions."""
    a = np.ascontiguousarray(pairs, np.float32).reshape(-1, 6)
    out = np.zeros((a.shape[0], 4), np.float32)
    lib().nqo_ciede_terms(a.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def ciede2000_sq(lab1, lab2):
    a = (C.c_float * 3)(*lab1)
    b = (C.c_float * 3)(*lab2)
    return lib().nqo_ciede2000(a, b)
