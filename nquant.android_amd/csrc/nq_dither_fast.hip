// nq_dither_fast.hip -- second device translation unit of libnquant_hip.so (gfx950 only): the specialised dither kernel.
// Build flags as nq_kernels.hip (-ffp-contract=off -fno-fast-math).  Own copy of the constant tables (upload_tables_fast).
#include "nq_device.h"
#include "nq_kernels.h"
#include <cstring>
#include <cmath>
#include <cstdlib>

// nq_dither_fast.inc -- the production form of the per-pixel nearest-colour + dither pass for
// the configuration family the headline benchmark and BASELINE cfg 3-5 use:
//     PnnLABQuantizer, 32 < K <= 256, no semi-transparency, DITHER_MAX = 25, not sorted-by-yDiff, PARALLEL_TILED.
// Every other configuration (and every tile this kernel gives up on, see `failed`) runs the generic gilbert_kernel of
// nq_dither.inc.  Results are bit for bit those of the generic kernel / of the oracle's tiled restatement; what changes is
// HOW the same values are obtained:
//
//  * one compile-time specialisation: no RGB / CIEDE2000 / sorted-queue / alpha-weight code, so the chain state (25 x 4 error
//    queue) and the working set fit 2-3 wavefronts per SIMD instead of one with spills (the generic kernel: 449 registers);
//  * the error queue is a register window of 25 + 4 boxes: five consecutive steps read boxes [u, u + 25) and append box 25 + u,
//    the window moves down by five boxes once per five steps (20 moves per step instead of 100); the weights of
//    initWeights(25) are literal operands (no scalar registers), verified against the host's table before every launch;
//  * closestColorIndex (NQ/PnnLABQuantizer.java:415-461): only floor(err_k) of every candidate matters (all comparisons are
//    against ints).  err_k is a positive quadratic form in (dr, dg, db); it is evaluated in float32 with a proven error bound
//    and only a candidate whose float32 value lies within that bound of an integer is re-evaluated with the reference's own
//    f64 statement sequence (closest_err_exact);
//  * nearestColorIndex (NQ/PnnLABQuantizer.java:369-375, K > 32): the argmin is first sought with a float32 Lab of the colour;
//    if the runner-up is not clearly (2 eps) behind, the exact f64 path (RGB2LAB_fast + reference arithmetic) decides;
//  * (float) Math.tanh((double) x) (NQ/GilbertCurve.java:246): 1 - 2t/(1 + t), t = exp(-2|x|) in f64 with a 4e-16 error, accepted
//    when v +- 1e-14 round to the same float, else the library tanh decides;
//  * Y_Diff threshold tests: float32 first, f64 when within 1e-3 of the threshold;
//  * pixel + saliency of the next step are fetched while the current step computes; indices are staged per lane in LDS
//    (conflict-free stride) and leave as whole rows (16-byte stores).
//
// Exactness of the float32 filters is argued next to each of them; tests/test_gpu_parity.py checks the kernel against the
// oracle (tiled cases, the 2^24 colour cube for the lookups) and against the generic kernel (NQ_OPT_FAST_DITHER off).

namespace nq {

struct FastArgs {
    float qa, qb, qc;        // closestColorIndex err as a quadratic form: qa dr^2 + qb dg^2 + qc db^2 (float32 of the f64 weights)
    float limiterDiv;        // (float) (1 + Math.sqrt(ditherMax))  (NQ/GilbertCurve.java:251)
    int strideBytes;         // per-lane stride of the staged index rows (multiple of 4, odd number of dwords)
    int failedCap;
    int vecOut;              // 1: rows leave as 16-byte (ARGB) / 8-byte (index) stores (tile width, image width and pointers allow it)
    int* failed;             // [0] = number of tiles handed to the generic kernel, [1..] = their indices
    const uint4* packed;     // [65536][2]: per colour cell {closest list, nearest list} in one 32-byte record (pack_lists_kernel)
    const uint4* cont;       // [2][65536]: candidates 15..30 of the closest / nearest lists
    int debug;               // timing experiments only (builds with -DNQ_FAST_KNOCKOUT): bit mask of stages to leave out
};
// instruction budget by block (tools/fast_budget.py compiles this file with -DNQ_FAST_MARKS and counts between the markers)
#ifdef NQ_FAST_MARKS
#define NQ_MARK(name) asm volatile("; MARK " name)
#else
#define NQ_MARK(name)
#endif
#ifdef NQ_FAST_KNOCKOUT
#define NQ_KO(bit) (F.debug & (bit))
#else
#define NQ_KO(bit) false
#endif

// initWeights(25) (NQ/GilbertCurve.java:336-354) as bit patterns; the host compares them with its own table
#define NQ_FAST_W(t) __uint_as_float(k_fast_w25[t])
static constexpr unsigned k_fast_w25[25] = {
    0x3a24fc7du, 0x3a5271d1u, 0x3a8636c6u, 0x3aab3193u, 0x3ada5cbdu, 0x3b0b437au, 0x3b31a273u, 0x3b6293edu, 0x3b9080cfu, 0x3bb8515cu,
    0x3beb1a3eu, 0x3c15f09cu, 0x3c3f40a4u, 0x3c73f2a9u, 0x3c9b94c7u, 0x3cc672b7u, 0x3cfd2047u, 0x3d216f47u, 0x3d4dea18u, 0x3d835325u,
    0x3da78228u, 0x3dd5a962u, 0x3e084405u, 0x3e2dcf89u, 0x3e5db34bu};
bool fast_weights_match(const float* w25) {
    for (int i = 0; i < 25; ++i) { unsigned u; std::memcpy(&u, &w25[i], 4); if (u != k_fast_w25[i]) return false; }
    return true;
}

// One 32-byte record per 5-6-5 colour cell: bytes 0..14 the first closest candidates, byte 15 their number, bytes 16..30 the
// first nearest candidates, byte 31 their number (255 = the cell needs a full scan or has more than 31 candidates: the tile goes to
// the generic kernel).  Both lists of a lookup arrive with ONE cache line instead of four (two lists + two counts in separate
// arrays).  Candidates 15..30 of the few longer lists live in two continuation arrays (16 bytes per cell each).
__global__ void __launch_bounds__(256) pack_lists_kernel(CellLists L, uint4* __restrict__ out, uint4* __restrict__ cont) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= 65536) return;
    __align__(16) unsigned char rec[32];
    __align__(16) unsigned char more[32];
    for (int half = 0; half < 2; ++half) {
        const unsigned char* src = (half ? L.nearest : L.closest) + (size_t) cell * NQ_LIST_CAP;
        const int n = (half ? L.nearestCount : L.closestCount)[cell];
        const bool ok = n <= 31;
        for (int i = 0; i < 15; ++i) rec[half * 16 + i] = ok && i < n ? src[i] : 0;
        rec[half * 16 + 15] = ok ? (unsigned char) n : 255;
        for (int i = 0; i < 16; ++i) more[half * 16 + i] = ok && 15 + i < n ? src[15 + i] : 0;
    }
    const uint4* r = reinterpret_cast<const uint4*>(rec);
    const uint4* m = reinterpret_cast<const uint4*>(more);
    out[2 * cell] = r[0]; out[2 * cell + 1] = r[1];
    cont[cell] = m[0]; cont[65536 + cell] = m[1];
}

#define NQ_FQ 29           // boxes in the register window: 25 live + 4 appended before the window moves

struct FastLds {           // views into the dynamic LDS block (layout: fast_lds_bytes)
    const int* argb;       // [256]
    const float4* lab;     // [256] {L, A, B, -}
    const double* gamma;   // [256]
    const float* gamma32;  // [256]
    const signed char* blue;   // [4096]
    const unsigned* path;  // [4][tilepx]
    int4* tileinfo;        // [256] {x0, y0, w, h} of every lane's tile
    unsigned char* stage;  // [4 waves][64 lanes][strideBytes]
};
__host__ __device__ __forceinline__ size_t fast_lds_bytes(int tilepx, int strideBytes) {
    return 256 * 4 + 256 * 16 + 256 * 8 + 256 * 4 + 4096 + (size_t) 4 * tilepx * 4 + 256 * 16 + (size_t) 4 * 64 * strideBytes;
}
__host__ __device__ __forceinline__ int fast_stride_bytes(int tilepx) {
    int dw = (tilepx + 3) / 4 + 1;
    if ((dw & 1) == 0) ++dw;
    return dw * 4;
}

// ---- (float) Math.tanh((double) x) --------------------------------------------------------------------------------------------
// v = 1 - 2t/(1 + t), t = exp(-2|x|): Cody-Waite reduction, degree-12 Taylor polynomial on |f| <= ln2/2 (truncation 1.7e-16),
// two Newton steps on v_rcp_f64; |v - tanh| < 1e-15.  The float nearest to v is the float nearest to tanh unless a rounding
// boundary lies within that error: then (v - 1e-14) and (v + 1e-14) round differently and the library decides (3e-7 of all calls).
__device__ __attribute__((noinline)) float tanh_library(double ax) { return (float) tanh(ax); }      // cold: 3e-7 of the calls
__device__ __attribute__((noinline)) float normal_distribution_cold(float x, float peak) { return normalDistribution(x, peak); }
__device__ __forceinline__ float tanh_to_float(float xf) {
    const double x = (double) xf;
    const double ax = fabs(x);
    float r;
    if (ax >= 9.2) r = 1.0f;                     // 1 - tanh < 2 exp(-18.4) = 2.1e-8 < 2^-25: rounds to 1.0f
    else if (ax < 0.5) r = tanh_library(ax);     // not reached by the limiter (|e| >= ditherMax); kept for completeness
    else {
        const double z = -2.0 * ax;
        const double n = rint(z * 1.4426950408889634);
        double f = fma(n, -0.6931471803691238, z);            // ln2 high part (32 significant bits)
        f = fma(n, -1.9082149292705877e-10, f);               // ln2 low part
        double p = 2.08767569878681e-09;                      // 1/12!
        p = fma(p, f, 2.505210838544172e-08);
        p = fma(p, f, 2.755731922398589e-07);
        p = fma(p, f, 2.7557319223985893e-06);
        p = fma(p, f, 2.48015873015873e-05);
        p = fma(p, f, 0.0001984126984126984);
        p = fma(p, f, 0.001388888888888889);
        p = fma(p, f, 0.008333333333333333);
        p = fma(p, f, 0.041666666666666664);
        p = fma(p, f, 0.16666666666666666);
        p = fma(p, f, 0.5);
        p = fma(p, f, 1.0);
        p = fma(p, f, 1.0);
        const double t = ldexp(p, (int) n);
        const double d = 1.0 + t;
        double rc = __builtin_amdgcn_rcp(d);
        rc = fma(fma(-d, rc, 1.0), rc, rc);
        rc = fma(fma(-d, rc, 1.0), rc, rc);
        const double v = 1.0 - 2.0 * t * rc;
        const float lo = (float) (v - 1e-14), hi = (float) (v + 1e-14);
        r = lo;
        if (lo != hi) r = tanh_library(ax);
    }
    return x < 0 ? -r : r;
}

// ---- closestColorIndex, one candidate, the reference's statement sequence (NQ/PnnLABQuantizer.java:421-445, no semi-transparency,
// ratio >= 0: every term is >= 0 and the gates only leave early a candidate that neither branch takes) -> err as f64
__device__ __attribute__((noinline)) double closest_err_exact(int c2, int cr, int cg, int cb, double wr, double wg, double wb, double ratio) {
    const int dr = c_red(c2) - cr, dg = c_green(c2) - cg, db = c_blue(c2) - cb;
    double err = wr * sqr((double) dr);
    err += wg * sqr((double) dg);
    err += wb * sqr((double) db);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        err += ratio * sqr((double) (k_coeffs[i][0] * dr));
        err += ratio * sqr((double) (k_coeffs[i][1] * dg));
        err += ratio * sqr((double) (k_coeffs[i][2] * db));
    }
    return err;
}

struct FastClosest { int c0, c1, e0, e1; };      // closest[0..3]

// one candidate k (palette word c2) of the closest scan.  e32 = float32 value of the quadratic form: relative error <= 6 * 2^-24
// (weights rounded once, three squares exact -- |d| <= 255 --, three multiplications, two additions, all terms >= 0) against the
// real-number form, which the reference's f64 sum follows to 2^-22 relative (it rounds coeff * d to FLOAT before squaring:
// NQ/PnnLABQuantizer.java:434-440, 2 * 2^-24 per YUV term).  |e32 - err| <= e32 * 2^-20 + 1e-6 =: delta is a safe bound; the
// integer part is taken from e32 unless e32 lies within delta of an integer AND the candidate could enter the tuple.
__device__ __forceinline__ void fast_closest_step(FastClosest& t, int k, int c2, float crf, float cgf, float cbf, int cr, int cg, int cb,
                                                  float qa, float qb, float qc, double wr, double wg, double wb, double ratio) {
    const float dr = (float) ((c2 >> 16) & 0xFF) - crf, dg = (float) ((c2 >> 8) & 0xFF) - cgf, db = (float) (c2 & 0xFF) - cbf;
    float e = qa * (dr * dr);
    e = __builtin_fmaf(qb, dg * dg, e);
    e = __builtin_fmaf(qc, db * db, e);
    int F = (int) e;                                             // e >= 0, < 2^20
    const float frac = e - (float) F;
    const float delta = __builtin_fmaf(e, 9.5367431640625e-07f, 1e-6f);
    if ((frac < delta || frac > 1.0f - delta) && F <= t.e1) F = (int) closest_err_exact(c2, cr, cg, cb, wr, wg, wb, ratio);
    // `if (err < closest[2]) ... else if (err < closest[3])` with int thresholds == the same tests on floor(err)
    const bool lt0 = F < t.e0, lt1 = F < t.e1;
    t.c1 = lt0 ? t.c0 : (lt1 ? k : t.c1);
    t.e1 = lt0 ? t.e0 : (lt1 ? F : t.e1);
    t.c0 = lt0 ? k : t.c0;
    t.e0 = lt0 ? F : t.e0;
}

// ---- float32 Lab of a colour for the nearest pre-selection.  Error against RGB2LAB_fast's float result: cube root by
// exp2(log2(x)/3) (v_log_f32 / v_exp_f32, 1 ulp each) is within 1e-6 relative; with the matrix in float32 (3e-7) the L, A, B
// errors stay below 1.5e-4, 1.2e-3 and 5e-4 (factors 116, 500, 200).  NQ_FAST_NEAR_EPS bounds the resulting error of
// |dL| + sqrt(dA^2 + dB^2) (1.5e-4 + 1.3e-3 + float32 evaluation 1e-4) with a factor 2 in hand.
#define NQ_FAST_NEAR_EPS 3.0e-3f
__device__ __forceinline__ float cbrt32_pivot(float c) {
    const float r = __builtin_amdgcn_exp2f(0.333333343f * __builtin_amdgcn_logf(c));
    return c > 0.008856f ? r : __builtin_fmaf(c, 7.787068966f, 0.137931034f);     // (903.3 c + 16) / 116
}
__device__ __forceinline__ void lab32_of(int c, const float* __restrict__ g32, float& L, float& A, float& B) {
    const float sr = g32[c_red(c)], sg = g32[c_green(c)], sb = g32[c_blue(c)];
    const float X = __builtin_fmaf(sb, 0.189906045f, __builtin_fmaf(sg, 0.376235967f, sr * 0.433891654f));   // 100 m / 95.047
    const float Y = __builtin_fmaf(sb, 0.0722f, __builtin_fmaf(sg, 0.7152f, sr * 0.2126f));
    const float Z = __builtin_fmaf(sb, 0.872955374f, __builtin_fmaf(sg, 0.109475308f, sr * 0.017725448f));   // 100 m / 108.883
    const float x = cbrt32_pivot(X), y = cbrt32_pivot(Y), z = cbrt32_pivot(Z);
    L = fmaxf(0.0f, __builtin_fmaf(116.0f, y, -16.0f));
    A = 500.0f * (x - y);
    B = 200.0f * (y - z);
}

// exact nearestColorIndex over a candidate list (the list branch of nearest_lab, nq_device.h): f64 reference arithmetic
__device__ __attribute__((noinline)) int fast_nearest_exact(const FastLds& S, int c, uint4 list, const uint4* cont, int n, int kfirst) {
    const Lab lab1 = RGB2LAB_fast(c, S.gamma);
    double mindist = 2147483647.0;
    int k = kfirst;
    unsigned w0 = list.x, w1 = list.y, w2 = list.z, w3 = list.w;
#pragma unroll 1
    for (int t = 0; t < n; ++t) {
        if (t == 15) { const uint4 m = *cont; w0 = m.x; w1 = m.y; w2 = m.z; w3 = m.w; }
        const int i = (int) (w0 & 0xFF);
        w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
        w2 = __builtin_amdgcn_alignbyte(w3, w2, 1); w3 >>= 8;
        const float4 l2 = S.lab[i];
        double curdist = (double) fabsf(l2.x - lab1.L);
        if (curdist > mindist) continue;
        curdist += sqrt(sqr((double) (l2.y - lab1.A)) + sqr((double) (l2.z - lab1.B)));
        if (curdist > mindist) continue;
        mindist = curdist;
        k = i;
    }
    return k;
}

// uniform context of the two lookups
struct FastLookup {
    const uint4* packed;     // FastArgs::packed
    const uint4* cont;       // FastArgs::cont
    float qa, qb, qc;        // closest quadratic form (float32)
    double wr, wg, wb, ratio;   // the reference's own weights PR (1 - ratio) ... for the exact evaluation
    int kfirst;              // first index of the nearest scan for a visible colour: 1 when palette[0] is the transparent colour
};

// closest[] of NQ/PnnLABQuantizer.java:415-461 for the colour c (alpha > 0xF) over the n1 listed candidates of its cell.
// The list is a 16-byte shift register: entry i is the low byte after i shifts; the palette word of entry i + 1 is requested from
// LDS before entry i is evaluated.
__device__ __forceinline__ FastClosest fast_closest_tuple(const FastLds& S, const FastLookup& X, int c, uint4 la, int n1, int cell) {
    const int cr = c_red(c), cg = c_green(c), cb = c_blue(c);
    const float crf = (float) cr, cgf = (float) cg, cbf = (float) cb;
    FastClosest t; t.c0 = t.c1 = 0; t.e0 = t.e1 = 2147483647;
    // the first eight entries straight-line (the longest list of a wavefront is ~7): their palette words are requested together,
    // no shift register, no loop control; the rare longer lists continue in the rolled loop
    {
        int kk[8], cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { kk[j] = (int) (((j < 4 ? la.x : la.y) >> ((j & 3) * 8)) & 0xFF); cc[j] = S.argb[kk[j]]; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < n1) fast_closest_step(t, kk[j], cc[j], crf, cgf, cbf, cr, cg, cb, X.qa, X.qb, X.qc, X.wr, X.wg, X.wb, X.ratio);
    }
    if (n1 > 8) {
        unsigned w0 = la.z, w1 = la.w, w2 = 0u, w3 = 0u;
        int k_next = (int) (w0 & 0xFF);
        int c2_next = S.argb[k_next];
#pragma unroll 1
        for (int i = 8; i < n1; ++i) {
            const int k = k_next, c2k = c2_next;
            w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
            w2 = __builtin_amdgcn_alignbyte(w3, w2, 1); w3 >>= 8;
            if (i == 14 && n1 > 15) { const uint4 m = X.cont[cell]; w0 = m.x; w1 = m.y; w2 = m.z; w3 = m.w; }
            k_next = (int) (w0 & 0xFF);
            c2_next = S.argb[k_next];
            fast_closest_step(t, k, c2k, crf, cgf, cbf, cr, cg, cb, X.qa, X.qb, X.qc, X.wr, X.wg, X.wb, X.ratio);
        }
    }
    if (t.e1 == 2147483647) t.c1 = t.c0;
    return t;
}

// nearestColorIndex (cache-miss semantics) of a visible colour for K > 32 (NQ/PnnLABQuantizer.java:369-375): argmin of
// |dL| + sqrt(dA^2 + dB^2) over the n2 listed candidates, ties to the higher index.  float32 first (lab32_of); the exact f64 scan
// decides when the runner-up is within 2 NQ_FAST_NEAR_EPS.
// float32 part: the argmin and whether it is safe (runner-up more than 2 NQ_FAST_NEAR_EPS behind)
__device__ __forceinline__ int fast_nearest32(const FastLds& S, const FastLookup& X, int c, uint4 na, int n2, int cell, bool& safe) {
    float L1, A1, B1;
    lab32_of(c, S.gamma32, L1, A1, B1);
    float d1 = 3.0e38f, d2 = 3.0e38f;
    int k1 = X.kfirst;
    {
        int kk[8]; float4 ll[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { kk[j] = (int) (((j < 4 ? na.x : na.y) >> ((j & 3) * 8)) & 0xFF); ll[j] = S.lab[kk[j]]; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < n2) {
                const float dA = ll[j].y - A1, dB = ll[j].z - B1;
                const float d = fabsf(ll[j].x - L1) + __builtin_amdgcn_sqrtf(__builtin_fmaf(dA, dA, dB * dB));
                const bool lt = d < d1;
                d2 = lt ? d1 : fminf(d2, d);
                k1 = lt ? kk[j] : k1;
                d1 = fminf(d1, d);
            }
    }
    if (n2 > 8) {
        unsigned w0 = na.z, w1 = na.w, w2 = 0u, w3 = 0u;
        int k_next = (int) (w0 & 0xFF);
        float4 l_next = S.lab[k_next];
#pragma unroll 1
        for (int i = 8; i < n2; ++i) {
            const int k = k_next;
            const float4 l2 = l_next;
            w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
            w2 = __builtin_amdgcn_alignbyte(w3, w2, 1); w3 >>= 8;
            if (i == 14 && n2 > 15) { const uint4 m = X.cont[65536 + cell]; w0 = m.x; w1 = m.y; w2 = m.z; w3 = m.w; }
            k_next = (int) (w0 & 0xFF);
            l_next = S.lab[k_next];
            const float dA = l2.y - A1, dB = l2.z - B1;
            const float d = fabsf(l2.x - L1) + __builtin_amdgcn_sqrtf(__builtin_fmaf(dA, dA, dB * dB));
            const bool lt = d < d1;
            d2 = lt ? d1 : fminf(d2, d);
            k1 = lt ? k : k1;
            d1 = fminf(d1, d);
        }
    }
    safe = d2 - d1 > 2.0f * NQ_FAST_NEAR_EPS;
    return k1;
}
__device__ __forceinline__ int fast_nearest(const FastLds& S, const FastLookup& X, int c, uint4 na, int n2, int cell) {
    bool safe;
    int k1 = fast_nearest32(S, X, c, na, n2, cell, safe);
    if (!safe) k1 = fast_nearest_exact(S, c, na, X.cont + 65536 + cell, n2, X.kfirst);
    return k1;
}

// RGB nearestColorIndex (NQ/PnnQuantizer.java:276-310) of an OPAQUE colour over the listed candidates of its cell: the reference's own f64
// statement sequence in index order (four multiply-adds per candidate: no float32 filter needed), ties to the higher index.
__device__ __forceinline__ int fast_nearest_rgb(const FastLds& S, double pa, double pr, double pg, double pb, int c, uint4 na, int n2,
                                                const uint4* cont_cell) {
    const int cr = c_red(c), cg = c_green(c), cb = c_blue(c);
    double mindist = 2147483647.0;
    int k = 0;
    unsigned w0 = na.x, w1 = na.y, w2 = na.z, w3 = na.w;
#pragma unroll 1
    for (int t = 0; t < n2; ++t) {
        if (t == 15) { const uint4 m = *cont_cell; w0 = m.x; w1 = m.y; w2 = m.z; w3 = m.w; }
        const int i = (int) (w0 & 0xFF);
        w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
        w2 = __builtin_amdgcn_alignbyte(w3, w2, 1); w3 >>= 8;
        const int c2 = S.argb[i];
        double curdist = pa * sqr((double) (c_alpha(c2) - 255));
        curdist += pr * sqr((double) (c_red(c2) - cr));
        curdist += pg * sqr((double) (c_green(c2) - cg));
        curdist += pb * sqr((double) (c_blue(c2) - cb));
        if (curdist > mindist) continue;          // (the partial-sum gates of the reference only skip what this one skips)
        mindist = curdist;
        k = i;
    }
    return k;
}

// which of the two closest candidates (NQ/PnnLABQuantizer.java:465-468): 0 when closest[2] == 0 or
// random.nextInt(32767) % (closest[3] + closest[2]) <= closest[3], else 1; draws from `rng` exactly when the reference does
__device__ __forceinline__ int fast_closest_pick(const FastClosest& t, long long& rng) {
    if (t.e0 == 0) return 0;
    // random.nextInt(32767): next(31), then the rejection loop of the non-power-of-two bound (taken with probability 1e-9)
    int r;
    for (;;) {
        const int uu = jr_next(rng, 31);
        int v = (uu >> 15) + (uu & 32767);                  // 32768 == 1 (mod 32767)
        v = (v >> 15) + (v & 32767);
        r = v >= 32767 ? v - 32767 : v;
        if ((int) ((unsigned) (uu - r) + 32766u) >= 0) break;
    }
    const int dsum = (int) ((unsigned) t.e1 + (unsigned) t.e0);
    int rem = r;                                            // dsum < 0 (wrapped) or dsum > r: r % dsum == r
    if (dsum > 0 && dsum <= r) {
        // r < 2^15: the float quotient is within 0.01 of the true one
        int qq = (int) ((float) r * __builtin_amdgcn_rcpf((float) dsum));
        rem = r - qq * dsum;
        if (rem < 0) rem += dsum; else if (rem >= dsum) rem -= dsum;
    }
    return rem <= t.e1 ? 0 : 1;
}

// Y_Diff(c1, c2) > thr (gt) or < thr (!gt), NQ/CIELABConvertor.java:215-227.  float32 first: each luminance is within 4e-7 of
// its f64 value (table entries 6e-8 relative, three products, two sums, Y <= 1), the difference times 100 within 1e-4.
__device__ __forceinline__ bool fast_ydiff_cmp(const FastLds& S, int c1, int c2, double thr, bool gt) {
    const float* g = S.gamma32;
    const float y1 = __builtin_fmaf(g[c_blue(c1)], 0.0722f, __builtin_fmaf(g[c_green(c1)], 0.7152f, g[c_red(c1)] * 0.2126f));
    const float y2 = __builtin_fmaf(g[c_blue(c2)], 0.0722f, __builtin_fmaf(g[c_green(c2)], 0.7152f, g[c_red(c2)] * 0.2126f));
    const float yd = fabsf(y2 - y1) * 100.0f;
    const float thr32 = (float) thr;
    if (fabsf(yd - thr32) > 1e-3f) return gt ? yd > thr32 : yd < thr32;
    const double d = Y_Diff_y(color2Y_t(c1, S.gamma), color2Y_t(c2, S.gamma));
    return gt ? d > thr : d < thr;
}

// ditherPixel (NQ/GilbertCurve.java:125-185) for K > 32, up to its final lookup: returns the colour handed to it.
// qcur = palette[qPixels[bidx]] = palette[0] at this point (the array is fresh).
__device__ __forceinline__ int fast_dither_color(const FastLds& S, const GilbertConsts& G, int K, int x, int y, int c2, int pixel, float sal) {
    const float beta = G.beta;
    const int qcur = S.argb[0];
    const double weight = G.weightAbs;
    const float strength = 1 / 3.0f;
    const int acceptedDiff = max(2, K - G.margin);
    const int c_in = c2;
    if (2 * acceptedDiff > 100 || fast_ydiff_cmp(S, pixel, c2, (double) (2 * acceptedDiff), false)) {       // Y_Diff <= 100 always
        float kappa;
        if (K > 64) kappa = sal < .6f ? beta * .15f / sal : beta * .4f / sal;
        else if (weight < .005) kappa = beta * normal_distribution_cold(sal, .5f) + beta;
        else kappa = beta * .5f / sal;
        c2 = blue_diffuse_t(pixel, qcur, kappa, strength, x, y, S.blue);
    }
    const double gamma = beta;                  // K > 32
    if (gamma * acceptedDiff < 100.5 && fast_ydiff_cmp(S, pixel, c2, gamma * acceptedDiff, true)) {
        if (G.margin > 6) {
            // :150-168 for K > 32: kappa = beta * normalDistribution(sal, peak) or one of two closed forms
            float kappa = sal < .4f ? beta * .4f * sal : beta * .4f / sal;
            int c1 = c_in;
            float peak = 0.f;                   // > 0: kappa = beta * normalDistribution(sal, peak)
            if (sal < .9) peak = 2.0f;
            else {
                if (weight >= .0015 && sal < .6) c1 = pixel;
                if (weight >= .005 && sal < .6) peak = weight < .0008 ? 2.5f : 1.75f;
                else {                          // K >= 32
                    const double ub = 1 - K / 320.0;
                    if (sal > .15 && sal < ub) kappa = beta * (weight < .0025 ? .55f : .5f) / sal;
                    else peak = weight < .0025 ? 1.82f : 2.0f;
                }
            }
            if (peak > 0.f) kappa = beta * normal_distribution_cold(sal, peak);
            c2 = blue_diffuse_t(c1, qcur, kappa, strength, x, y, S.blue);
        }
        else c2 = c_in;
    }
    if (sal > .95) {
        const float kappa = beta * fmaxf(.05f, .75f - K / 128.0f) * sal;
        c2 = blue_diffuse_t(pixel, qcur, kappa, strength, x, y, S.blue);
    }
    return c2;
}

// error accumulation over the 25 boxes [U, U + 25) of the window, oldest first (NQ/GilbertCurve.java:194-204)
template <int U>
__device__ __forceinline__ void fast_accumulate(const float (&q)[NQ_FQ][4], float (&e)[4], float& maxErr) {
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const float w = NQ_FAST_W(t);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = e[j] + q[U + t][j] * w;
        maxErr = fmaxf(fmaxf(maxErr, e[0]), e[1]);
        maxErr = fmaxf(fmaxf(maxErr, e[2]), e[3]);
    }
    // keeps the five instantiations apart: without it the optimiser sinks them into ONE body fed by 100 register copies per case
    asm volatile("; window offset %0" :: "n"(U));
}

// KIND 1: PnnLABQuantizer (the family described at the top).  KIND 0: PnnQuantizer with dither = true, 32 < K <= 256, an image without
// transparency: GilbertCurve has no saliencies there, so a step is accumulate -> nearestColorIndex(c2) -> limiter -> queue
// (NQ/GilbertCurve.java:212-229 takes the plain lookup; NQ/PnnQuantizer.java:377-391 getDitherFn(true) = nearestColorIndex).
template <int MINWAVES, int KIND = 1>
__global__ void __launch_bounds__(256, MINWAVES) gilbert_fast_kernel(DevParams P, GilbertConsts G, TileGeom T, CellLists lists, FastArgs F,
                                                                      const int* __restrict__ pixels, const float* __restrict__ saliency,
                                                                      const int* __restrict__ g_palette, long long seed,
                                                                      unsigned short* __restrict__ out_index, int* __restrict__ out_argb) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tilepx = T.tile_w * T.tile_h;
    FastLds S;
    {
        unsigned char* p = smem;
        S.argb = (const int*) p; p += 256 * 4;
        S.lab = (const float4*) p; p += 256 * 16;
        S.gamma = (const double*) p; p += 256 * 8;
        S.gamma32 = (const float*) p; p += 256 * 4;
        S.blue = (const signed char*) p; p += 4096;
        S.path = (const unsigned*) p; p += (size_t) 4 * tilepx * 4;
        S.tileinfo = (int4*) p; p += 256 * 16;
        S.stage = p;
    }
    const int K = P.K;
    const int tid = threadIdx.x;
    {
        int* a = (int*) S.argb; float4* l = (float4*) S.lab; double* g = (double*) S.gamma; float* g32 = (float*) S.gamma32;
        int c2 = tid < K ? g_palette[tid] : 0;
        a[tid] = c2;
        if (KIND == 1) {
            const Lab l2 = RGB2LAB(c2);                // getLab(palette[i]) (NQ/PnnLABQuantizer.java:352)
            l[tid] = make_float4(l2.L, l2.A, l2.B, 0.f);
        }
        g[tid] = g_tab.gamma[tid];
        g32[tid] = (float) g_tab.gamma[tid];
        for (int i = tid; i < 1024; i += 256) ((int*) S.blue)[i] = ((const int*) g_tab.blue)[i];
        unsigned* pp = (unsigned*) S.path;
        for (int s = 0; s < 4; ++s)
            for (int i = tid; i < tilepx; i += 256) pp[s * tilepx + i] = i < T.path_len[s] ? T.path[s][i] : 0u;
    }

    const int ntiles = T.tiles_x * T.tiles_y;
    const int tile = blockIdx.x * 256 + tid;
    const bool live = tile < ntiles;
    const int ty = live ? tile / T.tiles_x : 0, tx = live ? tile - ty * T.tiles_x : 0;
    const int shape = (tx == T.tiles_x - 1 ? 1 : 0) | (ty == T.tiles_y - 1 ? 2 : 0);
    const int steps = live ? T.path_len[shape] : 0;
    const int x0 = tx * T.tile_w, y0 = ty * T.tile_h;
    S.tileinfo[tid] = make_int4(x0, y0, live ? T.shape_w[shape] : 0, live ? T.shape_h[shape] : 0);
    __syncthreads();

    const unsigned* __restrict__ path = S.path + shape * tilepx;
    unsigned char* stage = S.stage + (size_t) tid * F.strideBytes;        // (wave w, lane l) -> row 64 w + l
    const int width = T.width;
    long long rng = jr_seed((long long) mix64((unsigned long long) seed + (unsigned long long) (T.tile_base + tile)));
    const int gofs = T.y_origin * width;        // band-local pixel index -> pixel index in the whole image (TileGeom::y_origin)
    const bool branchA = KIND == 1 && G.hasSaliencies && G.dither;         // (!hasAlphaW: no semi-transparency here)
    const bool hasSal = KIND == 1 && G.hasSaliencies != 0;
    // RGB nearestColorIndex weights (NQ/PnnQuantizer.java:281-285; K > 2 here)
    const double rpa = P.PA, rpr = P.PR, rpg = P.PG, rpb = P.PB;
    FastLookup X;
    X.packed = F.packed; X.cont = F.cont; X.qa = F.qa; X.qb = F.qb; X.qc = F.qc;
    X.ratio = P.ratio; X.wr = P.PR * (1 - P.ratio); X.wg = P.PG * (1 - P.ratio); X.wb = P.PB * (1 - P.ratio);
    X.kfirst = P.hasAlpha ? 1 : 0;
    const float ditherMaxF = (float) G.ditherMax, ditherMax1 = (float) (G.ditherMax - 1);
    const bool illusionAll = S.blue[0] > G.thresold;                      // yDiff == 1: TELL_BLUE_NOISE[(int) 4096.0 & 4095]
    bool failed = false;

    float q[NQ_FQ][4];
#pragma unroll
    for (int t = 0; t < NQ_FQ; ++t) { q[t][0] = q[t][1] = q[t][2] = q[t][3] = 0.f; }     // run(): initWeights(25) zero boxes

    // software pipeline: the pixel (and saliency) of step s + 1 is requested during step s
    unsigned d_next = 0;
    int bidx_next = 0, pixel_next = 0;
    float sal_next = 0.f;
    if (steps > 0) {
        d_next = path[0];
        bidx_next = (x0 + (int) (d_next & 0xFFFFu)) + (y0 + (int) (d_next >> 16)) * width;
        pixel_next = pixels[bidx_next];
        if (hasSal) sal_next = saliency[bidx_next];
    }
    const int maxsteps = T.path_len[0];
    for (int s = 0; s < maxsteps; ++s) {
        const int u = s % 5;                    // uniform: position of this step inside the window
        if (s < steps) {
            const int bidx = bidx_next;
            const int pixel = pixel_next;
            const float sal = sal_next;
            const int dxx = (int) (d_next & 0xFFFFu), dyy = (int) (d_next >> 16);
            const int xx = x0 + dxx, yy = y0 + dyy + T.y_origin, pos = dyy * T.tile_w + dxx;      // (xx, yy): position in the whole image
            if (s + 1 < steps) {
                d_next = path[s + 1];
                bidx_next = (x0 + (int) (d_next & 0xFFFFu)) + (y0 + (int) (d_next >> 16)) * width;
                pixel_next = pixels[bidx_next];
                if (hasSal) sal_next = saliency[bidx_next];
            }

            NQ_MARK("step_begin");
            float e[4] = {(float) c_red(pixel), (float) c_green(pixel), (float) c_blue(pixel), (float) c_alpha(pixel)};
            float maxErr = 24.0f;               // DITHER_MAX - 1
            if (!NQ_KO(32))
            switch (u) {
                case 0: fast_accumulate<0>(q, e, maxErr); break;
                case 1: fast_accumulate<1>(q, e, maxErr); break;
                case 2: fast_accumulate<2>(q, e, maxErr); break;
                case 3: fast_accumulate<3>(q, e, maxErr); break;
                default: fast_accumulate<4>(q, e, maxErr); break;
            }
            const int r_pix = (int) fminf(255.0f, fmaxf(e[0], 0.0f));
            const int g_pix = (int) fminf(255.0f, fmaxf(e[1], 0.0f));
            const int b_pix = (int) fminf(255.0f, fmaxf(e[2], 0.0f));
            const int a_pix = (int) fminf(255.0f, fmaxf(e[3], 0.0f));
            const int c2 = c_argb(a_pix, r_pix, g_pix, b_pix);
            NQ_MARK("accumulated");

            // :212-229 (K > 32: never branch B)
            int c = c2;
            if (branchA && !(K >= 256 && sal > .99f) && !NQ_KO(8)) c = fast_dither_color(S, G, K, xx, yy, c2, pixel, sal);
            NQ_MARK("dither_color");

            // ---- Ditherable.nearestColorIndex(palette, c, bidx) = closestColorIndex (K > 4), NQ/PnnLABQuantizer.java:407-474
            int qidx = 0;
            if (KIND == 0) {
                // the candidate lists hold for opaque colours only (the alpha term of an entry is then the same for the whole cell)
                if (c_alpha(c) != 255) failed = true;
                else {
                    const int cell = cell_of(c);
                    const uint4 na = X.packed[2 * cell + 1];
                    const int nnr = (int) (na.w >> 24);
                    if (nnr == 255) failed = true;
                    else qidx = fast_nearest_rgb(S, rpa, rpr, rpg, rpb, c, na, nnr, X.cont + 65536 + cell);
                }
            }
            else if (c_alpha(c) <= 0xF) failed = true;      // transparent colour: the generic kernel redoes this tile
            else {
                const int cell = NQ_KO(1) ? 0 : cell_of(c);
                const uint4 la = X.packed[2 * cell], na = X.packed[2 * cell + 1];
                const int ncl = (int) (la.w >> 24), nnr = (int) (na.w >> 24);
                if (ncl == 255 || nnr == 255) failed = true;
                const int n1 = (ncl == 255 || NQ_KO(16)) ? 0 : ncl, n2 = nnr == 255 ? 0 : nnr;
                NQ_MARK("lists_fetched");
                const FastClosest t = fast_closest_tuple(S, X, c, la, n1, cell);
                NQ_MARK("closest_done");
                const int idx = fast_closest_pick(t, rng);              // :465-468
                const int ci = idx ? t.c1 : t.c0, ei = idx ? t.e1 : t.e0;
                qidx = ci;
                NQ_MARK("picked");
                if ((ei >= K || ci == 0 || c_alpha(S.argb[ci]) < c_alpha(c)) && !NQ_KO(2)) {
                    qidx = fast_nearest(S, X, c, na, n2, cell);
                }
                NQ_MARK("nearest_done");
            }

            // :236-264
            const int cq = S.argb[qidx];
            e[0] = (float) (r_pix - c_red(cq));
            e[1] = (float) (g_pix - c_green(cq));
            e[2] = (float) (b_pix - c_blue(cq));
            e[3] = (float) (a_pix - c_alpha(cq));
            const bool diffuse = S.blue[(bidx + gofs) & 4095] > G.thresold;
            NQ_MARK("error_formed");
            {
                float ea = e[0], eb = e[1], ec = e[2];
#pragma unroll 1
                for (int j = 0; j < 3; ++j) {
                    if (fabsf(ea) >= ditherMaxF && !NQ_KO(4)) {
                        if (diffuse) ea = tanh_to_float(ea / maxErr * 20) * ditherMax1;
                        else if (illusionAll) ea = (ea / maxErr) * ditherMax1;      // (float) ((double) (e / maxErr) * 1.0) * (ditherMax - 1)
                        else ea /= F.limiterDiv;
                    }
                    const float tmp = ea; ea = eb; eb = ec; ec = tmp;              // one copy of the limiter code: the channels rotate through it
                }
                e[0] = ea; e[1] = eb; e[2] = ec;
            }
            NQ_MARK("limited");
            // errorq.poll() + errorq.add(error): box 25 + u of the window; after the fifth step the window moves down by five
            switch (u) {
                case 0: q[25][0] = e[0]; q[25][1] = e[1]; q[25][2] = e[2]; q[25][3] = e[3]; asm volatile("; push 0"); break;
                case 1: q[26][0] = e[0]; q[26][1] = e[1]; q[26][2] = e[2]; q[26][3] = e[3]; asm volatile("; push 1"); break;
                case 2: q[27][0] = e[0]; q[27][1] = e[1]; q[27][2] = e[2]; q[27][3] = e[3]; asm volatile("; push 2"); break;
                case 3: q[28][0] = e[0]; q[28][1] = e[1]; q[28][2] = e[2]; q[28][3] = e[3]; asm volatile("; push 3"); break;
                default:
#pragma unroll
                    for (int t = 0; t < 20; ++t) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[t][j] = q[t + 5][j];
                    }
#pragma unroll
                    for (int t = 20; t < 24; ++t) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[t][j] = q[t + 5][j];
                    }
                    q[24][0] = e[0]; q[24][1] = e[1]; q[24][2] = e[2]; q[24][3] = e[3];
                    break;
            }
            stage[pos] = (unsigned char) qidx;
            NQ_MARK("step_end");
        }
    }

    if (failed && live) {
        const int slot = atomicAdd(F.failed, 1);
        if (slot < F.failedCap) F.failed[1 + slot] = tile;
    }
    __syncthreads();

    // write-out: the wave's 64 tiles leave row by row; lane l handles group it * 64 + l of every tile row
    {
        const int wave = tid >> 6, lane = tid & 63;
        const unsigned char* wstage = S.stage + (size_t) (wave * 64) * F.strideBytes;
        const int4* winfo = S.tileinfo + wave * 64;
        const int tw = T.tile_w, th = T.tile_h;
        if (F.vecOut) {
            const int gpr = tw >> 2;                    // 4-pixel groups per tile row
            for (int r = 0; r < th; ++r)
                for (int it = 0; it < gpr; ++it) {
                    const int g = it * 64 + lane;
                    const int t = g / gpr, cgrp = (g - t * gpr) * 4;
                    const int4 ti = winfo[t];
                    if (r < ti.w && cgrp < ti.z) {          // ti.z = tile width (multiple of 4 here), ti.w = tile height
                        const unsigned kk = *reinterpret_cast<const unsigned*>(wstage + (size_t) t * F.strideBytes + r * tw + cgrp);
                        const long long bidx = (long long) (ti.y + r) * width + ti.x + cgrp;
                        const unsigned k0 = kk & 0xFF, k1 = (kk >> 8) & 0xFF, k2 = (kk >> 16) & 0xFF, k3 = kk >> 24;
                        uint2 iv; iv.x = k0 | (k1 << 16); iv.y = k2 | (k3 << 16);
                        *reinterpret_cast<uint2*>(out_index + bidx) = iv;
                        if (out_argb) *reinterpret_cast<int4*>(out_argb + bidx) = make_int4(S.argb[k0], S.argb[k1], S.argb[k2], S.argb[k3]);
                    }
                }
        } else {
            for (int r = 0; r < th; ++r)
                for (int it = 0; it < tw; ++it) {
                    const int g = it * 64 + lane;
                    const int t = g / tw, cc = g - t * tw;
                    const int4 ti = winfo[t];
                    if (r < ti.w && cc < ti.z) {
                        const int k = wstage[(size_t) t * F.strideBytes + r * tw + cc];
                        const long long bidx = (long long) (ti.y + r) * width + ti.x + cc;
                        out_index[bidx] = (unsigned short) k;
                        if (out_argb) out_argb[bidx] = S.argb[k];
                    }
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The same two lookups as stand-alone kernels behind nq_nearest_index / nq_closest_tuple / LOOKUP_ONLY (BASELINE cfg 2): the
// pure functions of SURVEY 8b, so that the float32 filters of the dither kernel are checked against the oracle colour by colour
// (tests: the whole 2^24 colour cube).  Colours the filters do not cover (alpha <= 0xF, cells without a packed list) take the
// generic functions of nq_device.h.
// ------------------------------------------------------------------------------------------------
struct FastLookupLds { FastLds S; PalView pal; };
__device__ __forceinline__ FastLookupLds fast_stage_lookup(const DevParams& P, const int* __restrict__ g_palette) {
    __shared__ __align__(16) int s_argb[256];
    __shared__ __align__(16) float4 s_lab[256];
    __shared__ __align__(16) double s_gamma[256];
    __shared__ float s_gamma32[256];
    __shared__ float s_L[256], s_A[256], s_B[256];
    const int tid = threadIdx.x;
    const int c2 = tid < P.K ? g_palette[tid] : 0;
    s_argb[tid] = c2;
    const Lab l2 = RGB2LAB(c2);
    s_lab[tid] = make_float4(l2.L, l2.A, l2.B, 0.f);
    s_L[tid] = l2.L; s_A[tid] = l2.A; s_B[tid] = l2.B;
    s_gamma[tid] = g_tab.gamma[tid];
    s_gamma32[tid] = (float) g_tab.gamma[tid];
    __syncthreads();
    FastLookupLds r;
    r.S.argb = s_argb; r.S.lab = s_lab; r.S.gamma = s_gamma; r.S.gamma32 = s_gamma32; r.S.blue = g_tab.blue; r.S.path = nullptr;
    r.S.tileinfo = nullptr; r.S.stage = nullptr;
    r.pal.argb = s_argb; r.pal.L = s_L; r.pal.A = s_A; r.pal.B = s_B; r.pal.gamma = s_gamma; r.pal.blue = g_tab.blue;
    return r;
}
__device__ __forceinline__ FastLookup fast_lookup_ctx(const DevParams& P, const FastArgs& F) {
    FastLookup X;
    X.packed = F.packed; X.cont = F.cont; X.qa = F.qa; X.qb = F.qb; X.qc = F.qc;
    X.ratio = P.ratio; X.wr = P.PR * (1 - P.ratio); X.wg = P.PG * (1 - P.ratio); X.wb = P.PB * (1 - P.ratio);
    X.kfirst = P.hasAlpha ? 1 : 0;
    return X;
}
__device__ __forceinline__ int fast_nearest_any(const DevParams& P, const FastLookupLds& T, const FastLookup& X, const CellLists& lists, int c) {
    if (c_alpha(c) > 0xF) {
        const int cell = cell_of(c);
        const uint4 na = X.packed[2 * cell + 1];
        const int n = (int) (na.w >> 24);
        if (n != 255) return fast_nearest(T.S, X, c, na, n, cell);
    }
    return nearest_lab(P, T.pal, c, &lists);
}
__global__ void __launch_bounds__(256) fast_nearest_index_kernel(DevParams P, CellLists lists, FastArgs F, const int* __restrict__ g_palette,
                                                                 const int* __restrict__ colors, long long M, short* __restrict__ out) {
    const FastLookupLds T = fast_stage_lookup(P, g_palette);
    const FastLookup X = fast_lookup_ctx(P, F);
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long) gridDim.x * blockDim.x)
        out[i] = (short) fast_nearest_any(P, T, X, lists, colors[i]);
}
// LOOKUP_ONLY in two passes.  Pass 1 is the float32 pre-selection alone -- no f64 code, no generic fallback in the kernel, so it
// keeps a small register file (the one-pass form held 199 VGPRs = 2 wavefronts per SIMD and spent 58 % of its wavefront cycles
// waiting) -- and appends every pixel it cannot settle (runner-up within the error bound, alpha <= 0xF, a cell without a packed list)
// to a list; pass 2 runs the exact functions over that list.  Same results as the one-pass form.
__global__ void __launch_bounds__(256) fast_lookup_pass1_kernel(DevParams P, FastArgs F, const int* __restrict__ g_palette,
                                                                const int* __restrict__ pixels, long long N,
                                                                unsigned short* __restrict__ out_index, int* __restrict__ out_argb,
                                                                unsigned* __restrict__ todo /* [0] = count, [1..] = pixel indices */) {
    __shared__ __align__(16) int s_argb[256];
    __shared__ __align__(16) float4 s_lab[256];
    __shared__ float s_gamma32[256];
    const int tid = threadIdx.x;
    {
        const int c2 = tid < P.K ? g_palette[tid] : 0;
        s_argb[tid] = c2;
        const Lab l2 = RGB2LAB(c2);
        s_lab[tid] = make_float4(l2.L, l2.A, l2.B, 0.f);
        s_gamma32[tid] = (float) g_tab.gamma[tid];
    }
    __syncthreads();
    FastLds S;
    S.argb = s_argb; S.lab = s_lab; S.gamma = nullptr; S.gamma32 = s_gamma32; S.blue = nullptr; S.path = nullptr; S.tileinfo = nullptr; S.stage = nullptr;
    FastLookup X;
    X.packed = F.packed; X.cont = F.cont; X.qa = X.qb = X.qc = 0.f; X.wr = X.wg = X.wb = X.ratio = 0.0;
    X.kfirst = P.hasAlpha ? 1 : 0;
    const int lane = tid & 63;
    const long long stride = (long long) gridDim.x * blockDim.x;
    const long long n_round = (N + stride - 1) / stride * stride;              // whole wavefronts stay in the loop (ballot below)
    for (long long i = (long long) blockIdx.x * blockDim.x + tid; i < n_round; i += stride) {
        const bool in = i < N;
        const int c = in ? pixels[i] : (int) 0xFF000000;
        bool safe = false;
        int k = 0;
        if (c_alpha(c) > 0xF) {
            const int cell = cell_of(c);
            const uint4 na = X.packed[2 * cell + 1];
            const int n = (int) (na.w >> 24);
            if (n != 255) k = fast_nearest32(S, X, c, na, n, cell, safe);
        }
        const bool defer = in && !safe;
        const unsigned long long dm = __ballot(defer);
        if (dm) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&todo[0], (unsigned) __popcll(dm));
            base = (unsigned) __builtin_amdgcn_readfirstlane((int) base);
            if (defer) todo[1 + base + __popcll(dm & ((1ULL << lane) - 1ULL))] = (unsigned) i;
        }
        if (in && safe) {
            if (out_index) out_index[i] = (unsigned short) k;
            if (out_argb) out_argb[i] = s_argb[k];
        }
    }
}
__global__ void __launch_bounds__(256) fast_lookup_pass2_kernel(DevParams P, CellLists lists, FastArgs F, const int* __restrict__ g_palette,
                                                                const int* __restrict__ pixels, unsigned short* __restrict__ out_index,
                                                                int* __restrict__ out_argb, const unsigned* __restrict__ todo) {
    const FastLookupLds T = fast_stage_lookup(P, g_palette);
    const FastLookup X = fast_lookup_ctx(P, F);
    const unsigned count = todo[0];
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        const long long i = (long long) todo[1 + t];
        const int k = fast_nearest_any(P, T, X, lists, pixels[i]);          // (nMaxColors > 32 here: no alpha-0 rewrite)
        if (out_index) out_index[i] = (unsigned short) k;
        if (out_argb) out_argb[i] = T.S.argb[k];
    }
}
// BlueNoise.dither (NQ/BlueNoise.java:207-222), parallel form (one Random(mix64((seed ^ tag) + i)) per pixel, as bluenoise_kernel of
// nq_dither.inc and the oracle's tiled restatement): the lookup is closestColorIndex through the float32-filtered functions above
__global__ void __launch_bounds__(256) fast_bluenoise_kernel(DevParams P, CellLists lists, FastArgs F, const int* __restrict__ g_palette,
                                                             const int* __restrict__ pixels, int width, int height, int y_origin, float weight,
                                                             long long seed, unsigned short* __restrict__ io_index, int* __restrict__ out_argb) {
    __shared__ __align__(16) signed char s_blue[4096];
    for (int i = threadIdx.x; i < 1024; i += 256) ((int*) s_blue)[i] = ((const int*) g_tab.blue)[i];
    FastLookupLds T = fast_stage_lookup(P, g_palette);          // ends with a barrier
    T.S.blue = s_blue;
    const FastLookup X = fast_lookup_ctx(P, F);
    const int K = P.K;
    const long long N = (long long) width * height;
    const float strength = 1 / 3.0f;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long) gridDim.x * blockDim.x) {
        const int y = (int) (i / width), x = (int) (i - (long long) y * width);
        const long long gi = i + (long long) y_origin * width;
        const int pixel = pixels[i];
        const int c = blue_diffuse_t(pixel, T.S.argb[io_index[i]], weight, strength, x, y + y_origin, s_blue);
        long long rng = jr_seed((long long) mix64(((unsigned long long) seed ^ 0xB10E5EEDULL) + (unsigned long long) gi));
        int k;
        if (c_alpha(c) <= 0xF) k = nearest_lab(P, T.pal, c, &lists);                  // closestColorIndex :408-409
        else {
            const int cell = cell_of(c);
            const uint4 la = X.packed[2 * cell], na = X.packed[2 * cell + 1];
            const int ncl = (int) (la.w >> 24), nnr = (int) (na.w >> 24);
            FastClosest t;
            if (ncl != 255) t = fast_closest_tuple(T.S, X, c, la, ncl, cell);
            else { int cl[4]; closest_tuple_lab(P, T.pal, c, cl, &lists); t.c0 = cl[0]; t.c1 = cl[1]; t.e0 = cl[2]; t.e1 = cl[3]; }
            const int idx = fast_closest_pick(t, rng);
            const int ci = idx ? t.c1 : t.c0, ei = idx ? t.e1 : t.e0;
            k = ci;
            if (ei >= K || ci == 0 || c_alpha(T.S.argb[ci]) < c_alpha(c))
                k = nnr != 255 ? fast_nearest(T.S, X, c, na, nnr, cell) : nearest_lab(P, T.pal, c, &lists);
        }
        io_index[i] = (unsigned short) k;
        out_argb[i] = T.S.argb[k];
    }
}
__global__ void __launch_bounds__(256) fast_closest_tuple_kernel(DevParams P, CellLists lists, FastArgs F, const int* __restrict__ g_palette,
                                                                 const int* __restrict__ colors, long long M, int* __restrict__ out4) {
    const FastLookupLds T = fast_stage_lookup(P, g_palette);
    const FastLookup X = fast_lookup_ctx(P, F);
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long) gridDim.x * blockDim.x) {
        const int c = colors[i];
        int closest[4];
        if (c_alpha(c) <= 0xF) closest[0] = closest[1] = closest[2] = closest[3] = -1;
        else {
            const int cell = cell_of(c);
            const uint4 la = X.packed[2 * cell];
            const int n = (int) (la.w >> 24);
            if (n != 255) {
                const FastClosest t = fast_closest_tuple(T.S, X, c, la, n, cell);
                closest[0] = t.c0; closest[1] = t.c1; closest[2] = t.e0; closest[3] = t.e1;
            }
            else closest_tuple_lab(P, T.pal, c, closest, &lists);
        }
        reinterpret_cast<int4*>(out4)[i] = make_int4(closest[0], closest[1], closest[2], closest[3]);
    }
}
} // namespace nq

namespace nq {

static const int8_t h_blue_fast[4096] = {
#include "../../include/nq_blue_noise_64x64.inc"
};
void upload_tables_fast(const double gamma[256], double exp1_5, double exp1_75, hipStream_t s) {
    ConstTables t;            // (the copy below is waited for; g_tab is per device: every handle uploads to its own device)
    for (int i = 0; i < 256; ++i) t.gamma[i] = gamma[i];
    t.exp1_5 = exp1_5; t.exp1_75 = exp1_75;
    for (int i = 0; i < 4096; ++i) t.blue[i] = h_blue_fast[i];
    (void) hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tab), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
    (void) hipStreamSynchronize(s);
}

static inline CellLists to_lists_fast(const ListsView& v) {
    CellLists l; l.closest = v.closest; l.closestCount = v.closestCount; l.nearest = v.nearest; l.nearestCount = v.nearestCount;
    return l;
}

// The specialised kernel of nq_dither_fast.inc, when the configuration is inside its domain (returns false otherwise: the caller
// runs the generic kernel on every tile).  d_failed: int[1 + tiles] -- the tiles it hands back ({count, indices}); the caller
// then runs the generic kernel over that list.
bool gilbert_fast_eligible(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const ListsView& lv) {
    const int tilepx = T.tile_w * T.tile_h;
    const bool common = P.K > 32 && P.K <= 256 && !P.hasSemi && !P.rewriteA0 && !G.sortedByYDiff && !G.hasAlphaW && G.DITHER_MAX == 25 &&
                        fast_weights_match(G.weights) && lv.nearest && tilepx >= 1 && tilepx <= 1024 &&
                        fast_lds_bytes(tilepx, fast_stride_bytes(tilepx)) <= 160 * 1024 - 512;
    if (P.kind == 1) return common && lv.closest && P.ratio >= 0;
    // PnnQuantizer: dither = true (nearestColorIndex), no transparent colour (the RGB nearest lists exist for such images only), no saliencies
    return common && G.dither && !P.hasAlpha && !G.hasSaliencies;
}
// the packed records are needed by the LAB lookups and by the RGB dither kernel
bool fast_pack_wanted(const DevParams& P, const ListsView& lv) {
    return fast_lookup_eligible(P, lv) || (P.kind == 0 && P.K > 32 && P.K <= 256 && !P.hasSemi && !P.hasAlpha && lv.nearest && lv.closest);
}
// the packed list records every specialised kernel reads (the ABI calls this right behind the list builders, in front of the stage
// event of the per-pixel pass)
void launch_pack_lists(const ListsView& lv, void* d_packed, hipStream_t s) {
    hipLaunchKernelGGL(pack_lists_kernel, dim3(65536 / 256), dim3(256), 0, s, to_lists_fast(lv), (uint4*) d_packed, (uint4*) d_packed + 2 * 65536);
}
static FastArgs fast_args(const DevParams& P, const ListsView& lv, void* d_packed, hipStream_t s) {
    FastArgs F;
    std::memset(&F, 0, sizeof F);
    // err of NQ/PnnLABQuantizer.java:421-445 as a quadratic form (every YUV term is (coeff * d)^2)
    double sq[3] = {0, 0, 0};
    static const float kc[3][3] = {{0.299f, 0.587f, 0.114f}, {-0.14713f, -0.28886f, 0.436f}, {0.615f, -0.51499f, -0.10001f}};
    for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) sq[c] += (double) kc[i][c] * (double) kc[i][c];
    F.qa = (float) (P.PR * (1 - P.ratio) + P.ratio * sq[0]);
    F.qb = (float) (P.PG * (1 - P.ratio) + P.ratio * sq[1]);
    F.qc = (float) (P.PB * (1 - P.ratio) + P.ratio * sq[2]);
    F.packed = (const uint4*) d_packed;
    F.cont = F.packed + 2 * 65536;
    (void) lv; (void) s;
    return F;
}
hipError_t launch_gilbert_fast(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const ListsView& lv, const int* d_pixels,
                               const float* d_saliency, const int* d_palette, long long seed, unsigned short* d_index, int* d_argb,
                               int* d_failed, void* d_packed, hipStream_t s) {
    const int ntiles = T.tiles_x * T.tiles_y;
    const int tilepx = T.tile_w * T.tile_h;
    FastArgs F = fast_args(P, lv, d_packed, s);
    F.limiterDiv = (float) (1 + std::sqrt((double) G.ditherMax));
    F.strideBytes = fast_stride_bytes(tilepx);
    F.failedCap = ntiles;
    F.failed = d_failed;
#ifdef NQ_FAST_KNOCKOUT
    if (const char* e = std::getenv("NQ_FAST_DEBUG")) F.debug = std::atoi(e);
#endif
    F.vecOut = (T.tile_w % 4 == 0) && (T.width % 4 == 0) && ((uintptr_t) d_index % 8 == 0) && (!d_argb || (uintptr_t) d_argb % 16 == 0);
    hipError_t e = hipMemsetAsync(d_failed, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const size_t lds = fast_lds_bytes(tilepx, F.strideBytes);
    const int grid = (ntiles + 255) / 256;
    if (lds > 64 * 1024) {
        // more than 64 KB of dynamic LDS (16x16 tiles) needs the opt-in.  The attribute belongs to the function ON THE CURRENT DEVICE, so it
        // is set at every such launch (a host-side call, no process-wide flag: handles of several devices and threads stay independent)
        e = hipFuncSetAttribute((const void*) gilbert_fast_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (e != hipSuccess) return e;
    }
    if (P.kind == 1)
        hipLaunchKernelGGL((gilbert_fast_kernel<2, 1>), dim3(grid), dim3(256), lds, s, P, G, T, to_lists_fast(lv), F, d_pixels, d_saliency, d_palette,
                           seed, d_index, d_argb);
    else {
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute((const void*) gilbert_fast_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((gilbert_fast_kernel<2, 0>), dim3(grid), dim3(256), lds, s, P, G, T, to_lists_fast(lv), F, d_pixels, d_saliency, d_palette,
                           seed, d_index, d_argb);
    }
    return hipGetLastError();
}

// the lookups alone (nq_nearest_index, nq_closest_tuple, LOOKUP_ONLY) through the same device functions
bool fast_lookup_eligible(const DevParams& P, const ListsView& lv) {
    return P.kind == 1 && P.K > 32 && P.K <= 256 && !P.hasSemi && !P.rewriteA0 && lv.closest && lv.nearest && P.ratio >= 0;
}
static inline int fast_grid(int64_t n) { int64_t g = (n + 255) / 256; return (int) (g < 1 ? 1 : g > 256 * 16 ? 256 * 16 : g); }
void launch_fast_nearest_index(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_colors, int64_t M,
                               short* d_out, hipStream_t s) {
    const FastArgs F = fast_args(P, lv, d_packed, s);
    hipLaunchKernelGGL(fast_nearest_index_kernel, dim3(fast_grid(M)), dim3(256), 0, s, P, to_lists_fast(lv), F, d_palette, d_colors, (long long) M, d_out);
}
void launch_fast_closest_tuple(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_colors, int64_t M,
                               int* d_out4, hipStream_t s) {
    const FastArgs F = fast_args(P, lv, d_packed, s);
    hipLaunchKernelGGL(fast_closest_tuple_kernel, dim3(fast_grid(M)), dim3(256), 0, s, P, to_lists_fast(lv), F, d_palette, d_colors, (long long) M, d_out4);
}
void launch_fast_lookup_only(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_pixels, int64_t N,
                             unsigned short* d_index, int* d_argb, unsigned* d_todo /* [N + 1] */, hipStream_t s) {
    const FastArgs F = fast_args(P, lv, d_packed, s);
    (void) hipMemsetAsync(d_todo, 0, sizeof(unsigned), s);
    int64_t g1 = (N + 255) / 256;
    if (g1 > 256 * 32) g1 = 256 * 32;                // 32 workgroups per CU in flight: 8 wavefronts per SIMD, four passes over them
    if (g1 < 1) g1 = 1;
    hipLaunchKernelGGL(fast_lookup_pass1_kernel, dim3((unsigned) g1), dim3(256), 0, s, P, F, d_palette, d_pixels, (long long) N, d_index, d_argb, d_todo);
    hipLaunchKernelGGL(fast_lookup_pass2_kernel, dim3(1024), dim3(256), 0, s, P, to_lists_fast(lv), F, d_palette, d_pixels, d_index, d_argb,
                       (const unsigned*) d_todo);
}
void launch_fast_bluenoise(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_pixels, int width, int height,
                           int y_origin, float weight, long long seed, unsigned short* d_index, int* d_argb, hipStream_t s) {
    const FastArgs F = fast_args(P, lv, d_packed, s);
    hipLaunchKernelGGL(fast_bluenoise_kernel, dim3(fast_grid((int64_t) width * height)), dim3(256), 0, s, P, to_lists_fast(lv), F, d_palette, d_pixels,
                       width, height, y_origin, weight, seed, d_index, d_argb);
}

} // namespace nq
