"""Pins the CPU oracle against external known answers (the reference ships no tests):
Sharma-Wu-Dalal CIEDE2000 table, sRGB<->Lab identities of ColorUtils, java.util.Random, gilbert-curve bijection,
GilbertCurve constructor ladder, BitmapUtilities.getColorIndex."""
import ctypes as C
import math

import numpy as np
import pytest

SHARMA = [
    (50.0000, 2.6772, -79.7751, 50.0000, 0.0000, -82.7485, 2.0425),
    (50.0000, 3.1571, -77.2803, 50.0000, 0.0000, -82.7485, 2.8615),
    (50.0000, 2.8361, -74.0200, 50.0000, 0.0000, -82.7485, 3.4412),
    (50.0000, -1.3802, -84.2814, 50.0000, 0.0000, -82.7485, 1.0000),
    (50.0000, -1.1848, -84.8006, 50.0000, 0.0000, -82.7485, 1.0000),
    (50.0000, -0.9009, -85.5211, 50.0000, 0.0000, -82.7485, 1.0000),
    (50.0000, 0.0000, 0.0000, 50.0000, -1.0000, 2.0000, 2.3669),
    (50.0000, -1.0000, 2.0000, 50.0000, 0.0000, 0.0000, 2.3669),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0009, 7.1792),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0010, 7.1792),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0011, 7.2195),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0012, 7.2195),
    (50.0000, -0.0010, 2.4900, 50.0000, 0.0009, -2.4900, 4.8045),
    (50.0000, -0.0010, 2.4900, 50.0000, 0.0010, -2.4900, 4.8045),
    (50.0000, -0.0010, 2.4900, 50.0000, 0.0011, -2.4900, 4.7461),
    (50.0000, 2.5000, 0.0000, 50.0000, 0.0000, -2.5000, 4.3065),
    (50.0000, 2.5000, 0.0000, 73.0000, 25.0000, -18.0000, 27.1492),
    (50.0000, 2.5000, 0.0000, 61.0000, -5.0000, 29.0000, 22.8977),
    (50.0000, 2.5000, 0.0000, 56.0000, -27.0000, -3.0000, 31.9030),
    (50.0000, 2.5000, 0.0000, 58.0000, 24.0000, 15.0000, 19.4535),
    (50.0000, 2.5000, 0.0000, 50.0000, 3.1736, 0.5854, 1.0000),
    (50.0000, 2.5000, 0.0000, 50.0000, 3.2972, 0.0000, 1.0000),
    (50.0000, 2.5000, 0.0000, 50.0000, 1.8634, 0.5757, 1.0000),
    (50.0000, 2.5000, 0.0000, 50.0000, 3.2592, 0.3350, 1.0000),
    (60.2574, -34.0099, 36.2677, 60.4626, -34.1751, 39.4387, 1.2644),
    (63.0109, -31.0961, -5.8663, 62.8187, -29.7946, -4.0864, 1.2630),
    (61.2901, 3.7196, -5.3901, 61.4292, 2.2480, -4.9620, 1.8731),
    (35.0831, -44.1164, 3.7933, 35.0232, -40.0716, 1.5901, 1.8645),
    (22.7233, 20.0904, -46.6940, 23.0331, 14.9730, -42.5619, 2.0373),
    (36.4612, 47.8580, 18.3852, 36.2715, 50.5065, 21.2231, 1.4146),
    (90.8027, -2.0831, 1.4410, 91.1528, -1.6435, 0.0447, 1.4441),
    (90.9257, -0.5406, -0.9208, 88.6381, -0.8985, -0.7239, 1.5381),
    (6.7747, -0.2908, -2.4247, 5.8714, -0.0985, -2.2286, 0.6377),
    (2.0776, 0.0795, -1.1350, 0.9033, -0.0636, -0.5514, 0.9082),
]
# Rows 10 and 14 sit exactly on the 180-degree hue discontinuity.  The reference compares against FLOAT constants
# deg180InRad/deg360InRad (NQ/CIELABConvertor.java:123-124,168), float(pi) > pi, so for exactly antipodal hues it takes the
# other branch than the published table; a literal restatement must reproduce that (value of the neighbouring row).
DISCONTINUITY = {10: 7.2195, 14: 4.7461}


def test_ciede2000_sharma_table(oracle):
    for i, row in enumerate(SHARMA, start=1):
        want = DISCONTINUITY.get(i, row[6])
        for a, b in ((row[0:3], row[3:6]), (row[3:6], row[0:3])):
            got = math.sqrt(max(0.0, oracle.ciede2000_sq(a, b)))
            assert abs(got - want) < 2e-4, (i, got, want)


def test_srgb_lab_identities(oracle):
    L = oracle.lib()
    known = {0xFFFFFFFF: (100.0, 0.00526, -0.01040), 0xFF000000: (0.0, 0.0, 0.0),
             0xFFFF0000: (53.2329, 80.1093, 67.2201), 0xFF00FF00: (87.7370, -86.1846, 83.1812),
             0xFF0000FF: (32.3026, 79.1967, -107.8637)}
    for c, (l, a, b) in known.items():
        alpha, gl, ga, gb = oracle.rgb2lab(c)
        assert alpha == 255.0
        assert abs(gl - l) < 2e-3 and abs(ga - a) < 2e-3 and abs(gb - b) < 2e-3, (hex(c), gl, ga, gb)
    # round trip over a colour cube: LABToColor(colorToLAB(c)) == c
    for r in range(0, 256, 51):
        for g in range(0, 256, 51):
            for b in range(0, 256, 51):
                c = (0xFF << 24) | (r << 16) | (g << 8) | b
                al, l_, a_, b_ = oracle.rgb2lab(c)
                assert (L.nqo_lab2rgb(al, l_, a_, b_) & 0xFFFFFFFF) == c


def test_srgb_lab_published_secondaries_and_gray(oracle):
    """More anchors for the restated ColorUtils.colorToLAB (D65, white 95.047 / 100 / 108.883): the published CIELAB values of
    the sRGB secondaries and mid grey (Lindbloom / EasyRGB tables; androidx.core's ColorUtilsTest lists the same triples for
    BLACK, WHITE, RED, GREEN, BLUE and CYAN to three decimals)."""
    known = {0xFFFFFF00: (97.138, -21.556, 94.482),     # yellow
             0xFF00FFFF: (91.117, -48.080, -14.138),    # cyan
             0xFFFF00FF: (60.320, 98.254, -60.843),     # magenta
             0xFF808080: (53.585, 0.003, -0.006)}       # mid grey: a*, b* are the white-point rounding residue, not 0
    for c, (l, a, b) in known.items():
        _, gl, ga, gb = oracle.rgb2lab(c)
        assert abs(gl - l) < 1.5e-3 and abs(ga - a) < 1.5e-3 and abs(gb - b) < 1.5e-3, (hex(c), gl, ga, gb)
    # L* is monotone in grey level and a*, b* stay at the residue
    prev = -1.0
    for v in range(256):
        _, gl, ga, gb = oracle.rgb2lab(0xFF000000 | v << 16 | v << 8 | v)
        assert gl > prev or v == 0
        assert abs(ga) < 0.01 and abs(gb) < 0.02
        prev = gl


def test_java_random_published_sequences(oracle):
    """java.util.Random is specified bit for bit (48-bit LCG, JLS / javadoc): the first outputs of new Random(0), new Random(42)
    and new Random(1) as printed by any JVM."""
    L = oracle.lib()
    st = C.c_int64(0)
    want = {0: [-1155484576, -723955400, 1033096058, -1690734402, -1557280266, 1327362106, -1930858313, 502539523, -1728529858, -938301587],
            42: [-1170105035, 234785527, -1360544799, 205897768],
            1: [-1155869325, 431529176, 1761283695, 1749940626]}
    for seed, seq in want.items():
        L.nqo_jrandom_seed(C.byref(st), seed)
        assert [L.nqo_jrandom_next_int(C.byref(st)) for _ in seq] == seq, seed
    # nextInt(bound) for a bound that is not a power of two = next(31) % bound with the rejection rule; bound 32767 is the one
    # closestColorIndex draws (NQ/PnnLABQuantizer.java:467): consistent with nextInt() >>> 1 of the same stream
    L.nqo_jrandom_seed(C.byref(st), 0)
    a = [L.nqo_jrandom_next_int_bound(C.byref(st), 32767) for _ in range(6)]
    assert a == [((v & 0xFFFFFFFF) >> 1) % 32767 for v in want[0][:6]]


def test_java_random_known_answers(oracle):
    L = oracle.lib()
    st = C.c_int64(0)
    L.nqo_jrandom_seed(C.byref(st), 0)
    assert L.nqo_jrandom_next_int(C.byref(st)) == -1155484576      # new Random(0).nextInt()
    L.nqo_jrandom_seed(C.byref(st), 42)
    assert L.nqo_jrandom_next_int(C.byref(st)) == -1170105035      # new Random(42).nextInt()
    L.nqo_jrandom_seed(C.byref(st), 42)
    assert [L.nqo_jrandom_next_int_bound(C.byref(st), 10) for _ in range(10)] == [0, 3, 8, 4, 0, 5, 5, 8, 9, 3]
    L.nqo_jrandom_seed(C.byref(st), 1)
    v = [L.nqo_jrandom_next_int_bound(C.byref(st), 32767) for _ in range(1000)]
    assert min(v) >= 0 and max(v) < 32767 and len(set(v)) > 900


@pytest.mark.parametrize("w,h", [(1, 1), (5, 1), (1, 7), (2, 2), (8, 8), (16, 16), (13, 7), (7, 13), (64, 64), (100, 37), (33, 96)])
def test_gilbert_curve_is_a_unit_step_bijection(oracle, w, h):
    xy = oracle.gilbert_path(w, h)
    assert len(set(map(tuple, xy.tolist()))) == w * h
    assert xy[:, 0].min() == 0 and xy[:, 0].max() == w - 1 and xy[:, 1].min() == 0 and xy[:, 1].max() == h - 1
    assert tuple(xy[0]) == (0, 0)
    if w * h > 1:
        step = np.abs(np.diff(xy, axis=0)).sum(1)
        # the generalised curve allows a single diagonal step only for odd-sized awkward rectangles
        assert step.max() <= 2 and (step != 1).sum() <= 1


def test_gilbert_constructor_ladder(oracle):
    L = oracle.lib()
    o = (C.c_int32 * 5)()
    # K=256, weight=256/65536, opaque
    beta = L.nqo_gilbert_params(256, 256 / 65536.0, 1, o)
    assert list(o) == [8, 0, 25, 23, -112] and abs(beta - 0.18) < 1e-7
    # config 1: K=16, weight ~ .0042
    beta = L.nqo_gilbert_params(16, 16 / 3863.0, 0, o)
    assert list(o) == [6, 0, 25, 23, -112] and abs(beta - 0.25) < 1e-7
    # sorted mode: K > 128 and weight >= .02
    beta = L.nqo_gilbert_params(256, 0.05, 1, o)
    assert o[1] == 1 and o[2] == 9 and o[4] == -64


def test_get_color_index(oracle):
    L = oracle.lib()
    c = np.int32(np.uint32(0x80FF8001).view(np.int32))
    assert L.nqo_get_color_index(int(c), 1, 0) == (0x80 << 8 | 0xF0 << 4 | 0x80 | 0)
    assert L.nqo_get_color_index(int(c), 0, 1) == (0x80 << 8 | 0xF8 << 7 | 0x80 << 2 | 0)
    assert L.nqo_get_color_index(int(c), 0, 0) == (0xF8 << 8 | 0x80 << 3 | 0)
    assert L.nqo_blue_noise(0) == 18 and L.nqo_blue_noise(4096) == 18
    assert sum(L.nqo_blue_noise(i) for i in range(4096)) == -2048       # each of -128..127 sixteen times


def test_integer_cube_root_resolution(oracle):
    """(int) Math.cbrt(cnt) is libm dependent on perfect cubes (glibc: cbrt(3375.0) = 14.999999999999998); oracle and GPU
    both use the exact integer cube root.  Here: libm agrees with it everywhere except on (some) perfect cubes."""
    lm = C.CDLL("libm.so.6")
    lm.cbrt.restype = C.c_double
    lm.cbrt.argtypes = [C.c_double]
    cubes = {n ** 3 for n in range(0, 257)}
    for v in list(range(0, 5000)) + [n ** 3 + d for n in range(2, 257) for d in (-1, 0, 1)]:
        exact = round(v ** (1.0 / 3.0))
        exact = exact if exact ** 3 <= v else exact - 1
        assert exact ** 3 <= v < (exact + 1) ** 3
        if v not in cubes:
            assert int(lm.cbrt(float(v))) == exact, v


@pytest.mark.parametrize("kind", [1, 0])
@pytest.mark.parametrize("mk,K", [("gradient_noise", 256), ("uniform_rgb", 64), ("with_alpha", 256), ("few_colors", 16)])
def test_virtual_merge_premise_holds_in_the_oracle(oracle, mk, K, kind):
    """The GPU's merge teams compute the find_nn that FOLLOWS a merge before the merge happens (csrc/nq_merge.inc, "virtual merge"): a scan
    over the unchanged lists with the merged count and means (`d = 1f / (n1 + n2)`, `d * (n1 x1 + n2 x2)`, NQ/PnnLABQuantizer.java:297-306)
    and with the neighbour passed over.  The oracle checks that premise at every merge of its own LAB loop: both scans, err (as float) and
    nn compared bit for bit -- texicab and non-texicab ladders, semi-transparency (the alpha term), few-hundred-bin images.  kind 0: the same
    premise for the RGB loop (`d * Math.round(n1 x1 + n2 x2)`, NQ/PnnQuantizer.java:240-248), which the GPU does not use yet."""
    from nquant.android_amd import synth
    img = {"gradient_noise": lambda: synth.gradient_noise(96, 80, 5), "uniform_rgb": lambda: synth.uniform_rgb(72, 72, 6),
           "with_alpha": lambda: synth.with_alpha(synth.gradient_noise(96, 96, 7), 7), "few_colors": lambda: synth.few_colors(96, 96, 8, 700)}[mk]()
    L = oracle.lib()
    oq = oracle.OracleQuantizer(kind, img, seed=1)
    oq.prescan(K)
    plain = oq.pnnquan(K)
    L.nqo_debug_virtual_merge(1, None)
    try:
        oq2 = oracle.OracleQuantizer(kind, img, seed=1)
        oq2.prescan(K)
        checked = oq2.pnnquan(K)
    finally:
        out = (C.c_int64 * 2)()
        L.nqo_debug_virtual_merge(0, out)
    assert (checked == plain).all()                      # (the check does not touch the loop's state)
    assert out[0] == max(0, oq2.params.maxbins - K) and out[0] > 0, (out[0], oq2.params.maxbins)
    assert out[1] == 0, "%d of %d merges: the scan before the merge differs from the scan behind it" % (out[1], out[0])
