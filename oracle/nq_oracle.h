/*
 * nq_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A sequential, single-threaded plain-C restatement of the reference's Java quantizer path
 * (mcychan/nQuant.android, NQ/ = nQuant.master/src/main/java/com/android/nQuant/):
 *   NQ/BitmapUtilities.java:6-20, NQ/CIELABConvertor.java (whole), NQ/BlueNoise.java:13-197,207-222,
 *   NQ/PnnQuantizer.java (whole), NQ/PnnLABQuantizer.java (whole), NQ/GilbertCurve.java (whole),
 *   NQ/Ditherable.java.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product (libnquant_hip.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" at two boundaries:
 *   - the reference ships no tests / golden vectors and cannot run here (no JVM);
 *   - androidx.core.graphics.ColorUtils (colorToLAB / LABToColor) is a third-party dependency absent from
 *     /root/reference; its published algorithm is restated in nqo_rgb2lab / nqo_lab2rgb.
 * What IS pinned: CIEDE2000 against the Sharma-Wu-Dalal table, sRGB<->Lab identities, java.util.Random
 * known answers, gilbert-curve bijection (tests/test_oracle_*.py).
 */
#ifndef NQ_ORACLE_H
#define NQ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nqo_quantizer nqo_quantizer;

/* Same layout as nq_params in include/nquant_abi.h (kept textually separate: the oracle is independent). */
typedef struct nqo_params {
    int32_t kind;                 /* 0 = PnnQuantizer (RGB), 1 = PnnLABQuantizer */
    int32_t nMaxColors;
    int32_t hasSemiTransparency;  /* NQ/PnnQuantizer.java:431 */
    int32_t transparentPixelIndex;/* m_transparentPixelIndex, -1 if none (:420) */
    int32_t transparentColor;     /* m_transparentColor (:22,:422) */
    int32_t isNano;               /* NQ/PnnLABQuantizer.java:180 (LAB only) */
    int32_t texicab;              /* :219 (LAB only) */
    int32_t quan_rt;              /* final value of quan_rt in pnnquan */
    int32_t maxbins;              /* non-empty histogram bins */
    int32_t paletteLength;
    double PR, PG, PB, PA;        /* NQ/PnnQuantizer.java:24,432-436 (+ :176-180 reset) */
    double ratio;                 /* value in force after pnnquan (LAB: after the :259-264 retune) */
    double weight;                /* signed: negated by dither() for semi-transparent images (:396-397) */
    int64_t distinctColors;       /* LAB: pixelMap.size() right after the histogram */
} nqo_params;

/* ---- object mirroring new PnnQuantizer(fname) / new PnnLABQuantizer(fname) (pixels handed in decoded) ---- */
nqo_quantizer* nqo_create(int kind, const int32_t* argb, int width, int height);
void nqo_destroy(nqo_quantizer* q);
/* Banded restatement of the LAB histogram for the multi-GPU split: the float32 sums of a bin restart at the first row of
 * every band (row_start[0] = 0 < row_start[1] < ... ) and the band partials are added in band order.  n_bands <= 1: reference order. */
void nqo_set_bands(nqo_quantizer* q, int n_bands, const int32_t* row_start);
void nqo_set_seed(nqo_quantizer* q, int64_t seed);      /* replaces the unseeded static Random (LAB :22) */
void nqo_get_params(const nqo_quantizer* q, nqo_params* out);
void nqo_set_params(nqo_quantizer* q, const nqo_params* in);  /* for function-level tests */

/* convert(nMaxColors, dither): NQ/PnnQuantizer.java:409-456.  out_argb = the Bitmap pixels (w*h).
 * out_index (nullable) = palette index finally chosen per pixel. out_palette has room for nMaxColors
 * (at least 2) entries. Returns 0, or <0 on a condition where the Java code would throw. */
int nqo_convert(nqo_quantizer* q, int nMaxColors, int dither,
                int32_t* out_argb, int32_t* out_index, int32_t* out_palette, int32_t* out_K);

/* Stage entry points (same object state as convert uses). */
void nqo_prescan(nqo_quantizer* q, int nMaxColors);                       /* :410-436 */
int  nqo_pnnquan(nqo_quantizer* q, int nMaxColors, int32_t* out_palette); /* returns palette length */
/* dither(): RGB NQ/PnnQuantizer.java:393-407, LAB NQ/PnnLABQuantizer.java:493-522 */
int  nqo_dither(nqo_quantizer* q, const int32_t* palette, int K, int dither,
                int32_t* out_argb, int32_t* out_index);

/* Tiled decomposition used by the GPU PARALLEL_TILED mode, restated on the CPU so that the GPU output can be
 * checked bit for bit: independent gilbert curve + error queue + Random(seed + tileIndex) per tile_w x tile_h
 * tile (tiles in row-major order), nearest/closest lookups with cache-miss semantics (no bin-keyed memo),
 * BlueNoise weight from the source image's distinct-colour count.  tile_w/tile_h <= 0 -> whole image. */
int  nqo_dither_tiled(nqo_quantizer* q, const int32_t* palette, int K, int dither, int tile_w, int tile_h,
                      int32_t* out_argb, int32_t* out_index);

/* The same, restricted to the tile rows [row_first, row_first + row_count): tiles are independent, so the result equals those
 * rows of nqo_dither_tiled; pixels outside come back as index 0 / palette[0].  For full-size images checked on a sample of rows. */
int  nqo_dither_tile_rows(nqo_quantizer* q, const int32_t* palette, int K, int dither, int tile_w, int tile_h,
                          int row_first, int row_count, int32_t* out_argb, int32_t* out_index);
/* diagnostics: event counters of the per-pixel pass (tests/oracle_event_rates.py) */
void nqo_debug_counters(int64_t* out16, int reset);

/* static GilbertCurve.dither / BlueNoise.dither with caller-supplied saliencies (nullable) and weight; tile <= 0 = sequential */
int  nqo_gilbert_dither_stage(nqo_quantizer* q, const int32_t* palette, int K, const float* saliencies, double weight, int dither,
                              int tile_w, int tile_h, int32_t* out_qpixels, int32_t* out_index);
int  nqo_bluenoise_dither_stage(nqo_quantizer* q, const int32_t* palette, int K, int32_t* io_qpixels, float weight, int tiled,
                                int32_t* out_index);

/* Pure lookups with cache-miss semantics (memo cleared before every colour). */
void nqo_nearest_index(nqo_quantizer* q, const int32_t* palette, int K, const int32_t* colors, int64_t M,
                       int16_t* out_index);
void nqo_closest_tuple(nqo_quantizer* q, const int32_t* palette, int K, const int32_t* colors, int64_t M,
                       int32_t* out_closest4);

/* ---- stateless pieces ---- */
int32_t nqo_get_color_index(int32_t c, int hasSemiTransparency, int hasTransparency); /* NQ/BitmapUtilities.java:8-15 */
void    nqo_rgb2lab(int32_t c, float* out_alpha_L_A_B);          /* NQ/CIELABConvertor.java:58-69 + ColorUtils.colorToLAB */
int32_t nqo_lab2rgb(float alpha, float L, float A, float B);     /* :77-80 + ColorUtils.LABToColor */
float   nqo_ciede2000(const float* lab1_L_A_B, const float* lab2_L_A_B); /* :201-213 (squared deltaE00) */
void    nqo_ciede_terms(const float* pairs6, int64_t n, float* out4);   /* L', C', H', R_T per pair, as find_nn obtains them (:93-110 of PnnLABQuantizer) */
double  nqo_y_diff(int32_t c1, int32_t c2);                      /* :215-227 */
double  nqo_u_diff(int32_t c1, int32_t c2);                      /* :229-238 */
int32_t nqo_blue_diffuse(int32_t pixel, int32_t qPixel, float weight, float strength, int x, int y); /* NQ/BlueNoise.java:180-197 */
int8_t  nqo_blue_noise(int i);                                   /* TELL_BLUE_NOISE[i & 4095] */
/* NQ/GilbertCurve.java:282-334,356-365: visiting order, out_xy = 2*w*h ints (x,y). Returns count. */
int64_t nqo_gilbert_path(int width, int height, int32_t* out_xy);
/* GilbertCurve constructor-derived scalars (:50-112), for tests of the magic-number ladders.
 * out: [0]=margin [1]=sortedByYDiff [2]=DITHER_MAX [3]=ditherMax [4]=thresold ; beta returned. */
float   nqo_gilbert_params(int K, double weight, int hasSaliencies, int32_t* out5);
/* java.util.Random */
void    nqo_jrandom_seed(int64_t* state, int64_t seed);
int32_t nqo_jrandom_next_int(int64_t* state);
int32_t nqo_jrandom_next_int_bound(int64_t* state, int32_t bound);

/* counters for the per-stage CPU baseline split (seconds, filled by nqo_convert) */
void nqo_get_stage_seconds(const nqo_quantizer* q, double* out6 /* prescan, hist, nn_init, merge, dither, bluenoise */);
int64_t nqo_get_find_nn_calls(const nqo_quantizer* q);

#ifdef __cplusplus
}
#endif
#endif
