/*
 * nq_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See nq_oracle.h for scope and
 * parity status ("parity unpinned" for the AndroidX LAB arithmetic; the reference has no tests).
 *
 * Plain C11, single thread, compile with -O2 -ffp-contract=off (Java float/double arithmetic is strict
 * IEEE-754 without fused multiply-add; x86-64 SSE gives FLT_EVAL_METHOD==0).
 *
 * NQ/ = /root/reference/nQuant.master/src/main/java/com/android/nQuant/
 * Every function cites the reference lines it restates.  Java semantics reproduced on purpose:
 *   (int)/(byte) narrowing of doubles (saturating, NaN->0); Math.round = floor(x+1/2); float vs double per
 *   expression; HashMap only as memo (+ keySet order for the few-colours early return); ArrayDeque FIFO;
 *   PriorityQueue array order under its iterator; java.util.Random 48-bit LCG with an injected seed.
 * Squares written Math.pow(x, 2) in the reference are evaluated as x*x (exact-result case of pow).
 */
#define _POSIX_C_SOURCE 200809L
#include "nq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------ */
/* Java numeric helpers                                                                              */
/* ------------------------------------------------------------------------------------------------ */
static inline int32_t j_d2i(double d) {           /* JLS 5.1.3 narrowing double -> int */
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t) d;
}
static inline int8_t j_d2b(double d) { return (int8_t) (uint8_t) (j_d2i(d) & 0xFF); } /* (byte) of a double */
static inline int64_t j_round(double a) {        /* Math.round(double): floor(a + 1/2), exact */
    if (a != a) return 0;
    if (fabs(a) >= 4503599627370496.0) return (int64_t) a;
    double f = floor(a);
    return (int64_t) f + ((a - f) >= 0.5 ? 1 : 0);
}
static inline int32_t i_add_wrap(int32_t a, int32_t b) { return (int32_t) ((uint32_t) a + (uint32_t) b); }
static inline double sqr(double v) { return v * v; }                 /* NQ/BitmapUtilities.java:17-20 */
#define J_PI 3.141592653589793
#define J_E  2.718281828459045

/* android.graphics.Color */
static inline int c_alpha(int32_t c) { return (int) (((uint32_t) c) >> 24); }
static inline int c_red(int32_t c)   { return (c >> 16) & 0xFF; }
static inline int c_green(int32_t c) { return (c >> 8) & 0xFF; }
static inline int c_blue(int32_t c)  { return c & 0xFF; }
static inline int32_t c_argb(int a, int r, int g, int b) {
    return (int32_t) (((uint32_t) a << 24) | ((uint32_t) r << 16) | ((uint32_t) g << 8) | (uint32_t) b);
}
#define BYTE_MAX 255
#define COLOR_BLACK ((int32_t) 0xFF000000u)
#define COLOR_WHITE ((int32_t) 0xFFFFFFFFu)

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------ */
/* Blue noise (DATA: NQ/BlueNoise.java:13-178)                                                       */
/* ------------------------------------------------------------------------------------------------ */
static const int8_t TELL_BLUE_NOISE[4096] = {
#include "../include/nq_blue_noise_64x64.inc"
};
int8_t nqo_blue_noise(int i) { return TELL_BLUE_NOISE[i & 4095]; }

/* NQ/BitmapUtilities.java:8-15 */
int32_t nqo_get_color_index(int32_t c, int hasSemiTransparency, int hasTransparency) {
    if (hasSemiTransparency)
        return (c_alpha(c) & 0xF0) << 8 | (c_red(c) & 0xF0) << 4 | (c_green(c) & 0xF0) | (c_blue(c) >> 4);
    if (hasTransparency)
        return (c_alpha(c) & 0x80) << 8 | (c_red(c) & 0xF8) << 7 | (c_green(c) & 0xF8) << 2 | (c_blue(c) >> 3);
    return (c_red(c) & 0xF8) << 8 | (c_green(c) & 0xFC) << 3 | (c_blue(c) >> 3);
}

/* ------------------------------------------------------------------------------------------------ */
/* CIELABConvertor                                                                                   */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { float alpha, A, B, L; } Lab;   /* NQ/CIELABConvertor.java:51-56 */

/* androidx.core.graphics.ColorUtils.colorToLAB (third party, androidx.core:core pulled by
 * androidx.appcompat:appcompat:1.4.0-alpha03, nQuant.master/build.gradle:36) -- published algorithm restated:
 * RGBToXYZ (sRGB companding, D65 matrix x100) then XYZToLAB (white 95.047/100/108.883, eps .008856, kappa 903.3,
 * pivot by Math.pow(c, 1/3.0)).  Call site NQ/CIELABConvertor.java:58-69. */
static double srgb_to_linear(int ch) {
    double s = ch / 255.0;
    return s < 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4);
}
static double pivot_xyz(double c) { return c > 0.008856 ? pow(c, 1 / 3.0) : (903.3 * c + 16) / 116; }
static Lab RGB2LAB(int32_t c1) {
    double sr = srgb_to_linear(c_red(c1)), sg = srgb_to_linear(c_green(c1)), sb = srgb_to_linear(c_blue(c1));
    double X = 100 * (sr * 0.4124 + sg * 0.3576 + sb * 0.1805);
    double Y = 100 * (sr * 0.2126 + sg * 0.7152 + sb * 0.0722);
    double Z = 100 * (sr * 0.0193 + sg * 0.1192 + sb * 0.9505);
    double x = pivot_xyz(X / 95.047), y = pivot_xyz(Y / 100.0), z = pivot_xyz(Z / 108.883);
    double l0 = fmax(0, 116 * y - 16), l1 = 500 * (x - y), l2 = 200 * (y - z);
    Lab lab;
    lab.alpha = (float) c_alpha(c1);
    lab.L = (float) l0; lab.A = (float) l1; lab.B = (float) l2;
    return lab;
}
void nqo_rgb2lab(int32_t c, float* o) { Lab l = RGB2LAB(c); o[0] = l.alpha; o[1] = l.L; o[2] = l.A; o[3] = l.B; }

/* NQ/CIELABConvertor.java:71-75 */
static double gammaToLinear(int channel) {
    const double c = channel / 255.0;
    return c < 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
}

/* ColorUtils.LABToColor = LABToXYZ + XYZToColor (published algorithm restated); NQ/CIELABConvertor.java:77-80.
 * Returns INT32 via out, status -1 where setAlphaComponent would throw (alpha outside 0..255). */
static int LAB2RGB(Lab lab, int32_t* out) {
    double l = lab.L, a = lab.A, b = lab.B;
    double fy = (l + 16) / 116, fx = a / 500 + fy, fz = fy - b / 200;
    double tmp = pow(fx, 3);
    double xr = tmp > 0.008856 ? tmp : (116 * fx - 16) / 903.3;
    double yr = l > 903.3 * 0.008856 ? pow(fy, 3) : l / 903.3;
    tmp = pow(fz, 3);
    double zr = tmp > 0.008856 ? tmp : (116 * fz - 16) / 903.3;
    double x = xr * 95.047, y = yr * 100.0, z = zr * 108.883;
    double r = (x * 3.2406 + y * -1.5372 + z * -0.4986) / 100;
    double g = (x * -0.9689 + y * 1.8758 + z * 0.0415) / 100;
    double bb = (x * 0.0557 + y * -0.2040 + z * 1.0570) / 100;
    r = r > 0.0031308 ? 1.055 * pow(r, 1 / 2.4) - 0.055 : 12.92 * r;
    g = g > 0.0031308 ? 1.055 * pow(g, 1 / 2.4) - 0.055 : 12.92 * g;
    bb = bb > 0.0031308 ? 1.055 * pow(bb, 1 / 2.4) - 0.055 : 12.92 * bb;
    int64_t ri = j_round(r * 255), gi = j_round(g * 255), bi = j_round(bb * 255);
    int R = ri < 0 ? 0 : ri > 255 ? 255 : (int) ri;
    int G = gi < 0 ? 0 : gi > 255 ? 255 : (int) gi;
    int B = bi < 0 ? 0 : bi > 255 ? 255 : (int) bi;
    int alpha = j_d2i(lab.alpha);
    int status = (alpha < 0 || alpha > 255) ? -1 : 0;
    *out = (int32_t) ((((uint32_t) alpha) << 24) | ((uint32_t) R << 16) | ((uint32_t) G << 8) | (uint32_t) B);
    return status;
}
int32_t nqo_lab2rgb(float alpha, float L, float A, float B) {
    Lab l; l.alpha = alpha; l.L = L; l.A = A; l.B = B; int32_t o; LAB2RGB(l, &o); return o;
}

/* NQ/CIELABConvertor.java:86-89 */
static float deg2Rad(double deg) { return (float) (deg * (J_PI / 180.0)); }

/* :91-98 */
static float L_prime_div_k_L_S_L(Lab lab1, Lab lab2) {
    const float k_L = 1.0f;
    float deltaLPrime = lab2.L - lab1.L;
    float barLPrime = (lab1.L + lab2.L) / 2.0f;
    double p = sqr((double) (barLPrime - 50.0f));
    float S_L = (float) (1 + (((double) 0.015f * p) / sqrt(20 + p)));
    return deltaLPrime / (k_L * S_L);
}
/* :100-118 */
static float C_prime_div_k_L_S_L(Lab lab1, Lab lab2, double* a1Prime, double* a2Prime, double* CPrime1, double* CPrime2) {
    const float k_C = 1.0f;
    const float pow25To7 = 6103515625.0f;
    float C1 = (float) sqrt((double) ((lab1.A * lab1.A) + (lab1.B * lab1.B)));
    float C2 = (float) sqrt((double) ((lab2.A * lab2.A) + (lab2.B * lab2.B)));
    float barC = (C1 + C2) / 2.0f;
    double barC7 = pow((double) barC, 7);
    float G = (float) ((double) 0.5f * (1 - sqrt(barC7 / (barC7 + (double) pow25To7))));
    *a1Prime = (1.0 + G) * lab1.A;
    *a2Prime = (1.0 + G) * lab2.A;
    *CPrime1 = sqrt((*a1Prime * *a1Prime) + (double) (lab1.B * lab1.B));
    *CPrime2 = sqrt((*a2Prime * *a2Prime) + (double) (lab2.B * lab2.B));
    float deltaCPrime = (float) *CPrime2 - (float) *CPrime1;
    float barCPrime = ((float) *CPrime1 + (float) *CPrime2) / 2.0f;
    float S_C = 1 + (0.045f * barCPrime);
    return deltaCPrime / (k_C * S_C);
}
/* :120-185 */
static float H_prime_div_k_L_S_L(Lab lab1, Lab lab2, double a1Prime, double a2Prime, double CPrime1, double CPrime2,
                                 double* barCPrime, double* barhPrime) {
    const float k_H = 1.0f;
    const float deg360InRad = deg2Rad(360.0f);
    const float deg180InRad = deg2Rad(180.0f);
    double CPrimeProduct = CPrime1 * CPrime2;
    double hPrime1;
    if ((double) lab1.B == 0.0 && a1Prime == 0.0) hPrime1 = 0.0;
    else {
        hPrime1 = atan2((double) lab1.B, a1Prime);
        if (hPrime1 < 0) hPrime1 += deg360InRad;
    }
    double hPrime2;
    if ((double) lab2.B == 0.0 && a2Prime == 0.0) hPrime2 = 0.0;
    else {
        hPrime2 = atan2((double) lab2.B, a2Prime);
        if (hPrime2 < 0) hPrime2 += deg360InRad;
    }
    double deltahPrime;
    if (CPrimeProduct == 0.0) deltahPrime = 0;
    else {
        deltahPrime = hPrime2 - hPrime1;
        if (deltahPrime < -deg180InRad) deltahPrime += deg360InRad;
        else if (deltahPrime > deg180InRad) deltahPrime -= deg360InRad;
    }
    double deltaHPrime = 2.0 * sqrt(CPrimeProduct) * sin(deltahPrime / 2.0);
    double hPrimeSum = hPrime1 + hPrime2;
    if ((CPrime1 * CPrime2) == 0.0) *barhPrime = hPrimeSum;
    else {
        if (fabs(hPrime1 - hPrime2) <= deg180InRad) *barhPrime = hPrimeSum / 2.0;
        else {
            if (hPrimeSum < deg360InRad) *barhPrime = (hPrimeSum + deg360InRad) / 2.0;
            else *barhPrime = (hPrimeSum - deg360InRad) / 2.0;
        }
    }
    *barCPrime = (CPrime1 + CPrime2) / 2.0;
    double bh = *barhPrime;
    double T = 1.0 - (0.17 * cos(bh - deg2Rad(30.0f))) + (0.24 * cos(2.0 * bh)) +
               (0.32 * cos((3.0 * bh) + deg2Rad(6.0f))) - (0.20 * cos((4.0 * bh) - deg2Rad(63.0f)));
    double S_H = 1 + ((double) 0.015f * *barCPrime * T);
    return (float) (deltaHPrime / (k_H * S_H));
}
/* :187-194 */
static float R_T(double barCPrime, double barhPrime, float C_prime_div, float H_prime_div) {
    const double pow25To7 = 6103515625.0;
    double deltaTheta = deg2Rad(30.0f) * exp(-sqr((barhPrime - deg2Rad(275.0f)) / deg2Rad(25.0f)));
    double bc7 = pow(barCPrime, 7.0);
    double R_C = 2.0 * sqrt(bc7 / (bc7 + pow25To7));
    double rt = (-sin(2.0 * deltaTheta)) * R_C;
    return (float) (rt * C_prime_div * H_prime_div);
}
/* :201-213 (dead code in the reference; a known-answer hook for the four pieces above) */
float nqo_ciede2000(const float* l1, const float* l2) {
    Lab lab1 = {255, l1[1], l1[2], l1[0]}, lab2 = {255, l2[1], l2[2], l2[0]};
    float dL = L_prime_div_k_L_S_L(lab1, lab2);
    double a1, a2, c1, c2, bc, bh;
    float dC = C_prime_div_k_L_S_L(lab1, lab2, &a1, &a2, &c1, &c2);
    float dH = H_prime_div_k_L_S_L(lab1, lab2, a1, a2, c1, c2, &bc, &bh);
    float rt = R_T(bc, bh, dC, dH);
    return (float) (sqr((double) dL) + sqr((double) dC) + sqr((double) dH) + rt);
}
/* the four CIEDE2000 terms of n pairs {L1, A1, B1, L2, A2, B2} exactly as find_nn obtains them (NQ/PnnLABQuantizer.java:93-110 calls the
 * four functions above in this order): out4[4 i + 0..3] = L', C', H' (each already divided by k S), R_T.  Checker of the GPU's
 * branch-free evaluation (nq_selftest_ciede). */
void nqo_ciede_terms(const float* pairs, int64_t n, float* out4) {
    for (int64_t i = 0; i < n; ++i) {
        const float* q = pairs + 6 * i;
        Lab lab1 = {255, q[1], q[2], q[0]}, lab2 = {255, q[4], q[5], q[3]};
        double a1, a2, c1, c2, bc, bh;
        out4[4 * i + 0] = L_prime_div_k_L_S_L(lab1, lab2);
        out4[4 * i + 1] = C_prime_div_k_L_S_L(lab1, lab2, &a1, &a2, &c1, &c2);
        out4[4 * i + 2] = H_prime_div_k_L_S_L(lab1, lab2, a1, a2, c1, c2, &bc, &bh);
        out4[4 * i + 3] = R_T(bc, bh, out4[4 * i + 1], out4[4 * i + 2]);
    }
}
/* :215-227 */
static double color2Y(int32_t c) {
    double sr = gammaToLinear(c_red(c)), sg = gammaToLinear(c_green(c)), sb = gammaToLinear(c_blue(c));
    return sr * 0.2126 + sg * 0.7152 + sb * 0.0722;
}
static double Y_Diff(int32_t c1, int32_t c2) {
    double y = color2Y(c1), y2 = color2Y(c2);
    return fabs(y2 - y) * 100;
}
/* :229-238 */
static double color2U(int32_t c) { return -0.09991 * c_red(c) - 0.33609 * c_green(c) + 0.436 * c_blue(c); }
static double U_Diff(int32_t c1, int32_t c2) { return fabs(color2U(c2) - color2U(c1)); }
double nqo_y_diff(int32_t c1, int32_t c2) { return Y_Diff(c1, c2); }
double nqo_u_diff(int32_t c1, int32_t c2) { return U_Diff(c1, c2); }

/* NQ/BlueNoise.java:180-197 */
static int32_t blue_diffuse(int32_t pixel, int32_t qPixel, float weight, float strength, int x, int y) {
    int r_pix = c_red(pixel), g_pix = c_green(pixel), b_pix = c_blue(pixel), a_pix = c_alpha(pixel);
    float adj = (TELL_BLUE_NOISE[(x & 63) | (y & 63) << 6] + 0.5f) / 127.5f;
    adj += (((x + y) & 1) - 0.5f) * strength / 8.0f;
    adj *= weight;
    r_pix = j_d2i(fmin(255.0, fmax((double) (r_pix + (adj * (r_pix - c_red(qPixel)))), 0.0)));
    g_pix = j_d2i(fmin(255.0, fmax((double) (g_pix + (adj * (g_pix - c_green(qPixel)))), 0.0)));
    b_pix = j_d2i(fmin(255.0, fmax((double) (b_pix + (adj * (b_pix - c_blue(qPixel)))), 0.0)));
    a_pix = j_d2i(fmin(255.0, fmax((double) (a_pix + (adj * (a_pix - c_alpha(qPixel)))), 0.0)));
    return c_argb(a_pix, r_pix, g_pix, b_pix);
}
int32_t nqo_blue_diffuse(int32_t p, int32_t q, float w, float s, int x, int y) { return blue_diffuse(p, q, w, s, x, y); }

/* ------------------------------------------------------------------------------------------------ */
/* java.util.Random (48-bit LCG), seed injected                                                      */
/* ------------------------------------------------------------------------------------------------ */
#define JR_MULT 0x5DEECE66DLL
#define JR_MASK ((1LL << 48) - 1)
void nqo_jrandom_seed(int64_t* st, int64_t seed) { *st = (seed ^ JR_MULT) & JR_MASK; }
static int32_t jr_next(int64_t* st, int bits) {
    *st = (int64_t) (((uint64_t) *st * (uint64_t) JR_MULT + 0xBULL) & (uint64_t) JR_MASK);
    return (int32_t) (*st >> (48 - bits));
}
int32_t nqo_jrandom_next_int(int64_t* st) { return jr_next(st, 32); }
int32_t nqo_jrandom_next_int_bound(int64_t* st, int32_t bound) {
    int32_t r = jr_next(st, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) r = (int32_t) (((int64_t) bound * (int64_t) r) >> 31);
    else {
        for (int32_t u = r; i_add_wrap(u - (r = u % bound), m) < 0; u = jr_next(st, 31)) { }
    }
    return r;
}

/* stream selector of the tiled decomposition (splitmix64 finaliser; same function in csrc/nq_device.h) */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* ------------------------------------------------------------------------------------------------ */
/* int -> blob hash map (memo semantics of java.util.HashMap<Integer,...>), insertion order kept      */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t* keys; uint8_t* used; uint8_t* vals; size_t vsize, cap, n;
    int32_t* order; size_t order_cap;
} imap;
static void imap_init(imap* m, size_t vsize) { memset(m, 0, sizeof *m); m->vsize = vsize; }
static void imap_free(imap* m) { free(m->keys); free(m->used); free(m->vals); free(m->order); memset(m, 0, sizeof *m); }
static void imap_clear(imap* m) { size_t v = m->vsize; imap_free(m); m->vsize = v; }
static inline size_t imap_hash(int32_t k) { uint32_t h = (uint32_t) k * 0x9E3779B1u; h ^= h >> 15; return h; }
static void* imap_get(const imap* m, int32_t key) {
    if (!m->cap) return NULL;
    size_t i = imap_hash(key) & (m->cap - 1);
    while (m->used[i]) { if (m->keys[i] == key) return m->vals + i * m->vsize; i = (i + 1) & (m->cap - 1); }
    return NULL;
}
static void imap_put(imap* m, int32_t key, const void* val);
static void imap_grow(imap* m) {
    imap old = *m;
    size_t ncap = old.cap ? old.cap * 2 : 1024;
    m->keys = malloc(ncap * sizeof(int32_t)); m->used = calloc(ncap, 1); m->vals = malloc(ncap * m->vsize);
    m->cap = ncap; m->n = 0;
    for (size_t i = 0; i < old.cap; ++i) if (old.used[i]) {
        size_t j = imap_hash(old.keys[i]) & (ncap - 1);
        while (m->used[j]) j = (j + 1) & (ncap - 1);
        m->used[j] = 1; m->keys[j] = old.keys[i]; memcpy(m->vals + j * m->vsize, old.vals + i * m->vsize, m->vsize); m->n++;
    }
    free(old.keys); free(old.used); free(old.vals);
}
static void imap_put(imap* m, int32_t key, const void* val) {
    if ((m->n + 1) * 10 >= m->cap * 6) imap_grow(m);
    size_t i = imap_hash(key) & (m->cap - 1);
    while (m->used[i]) {
        if (m->keys[i] == key) { memcpy(m->vals + i * m->vsize, val, m->vsize); return; }
        i = (i + 1) & (m->cap - 1);
    }
    m->used[i] = 1; m->keys[i] = key; memcpy(m->vals + i * m->vsize, val, m->vsize);
    if (m->n >= m->order_cap) { m->order_cap = m->order_cap ? m->order_cap * 2 : 1024; m->order = realloc(m->order, m->order_cap * sizeof(int32_t)); }
    m->order[m->n++] = key;
}
/* keySet() iteration order of java.util.HashMap<Integer,?> (OpenJDK 8+; treeified buckets ignored):
 * table capacity 16 doubling when size exceeds 0.75*capacity; bucket = (h ^ h>>>16) & (cap-1), h = key;
 * inside a bucket insertion order (resize keeps relative order). */
static size_t imap_java_keyset(const imap* m, int32_t* out) {
    size_t cap = 16;
    while ((double) m->n > 0.75 * (double) cap) cap <<= 1;
    size_t n = m->n, k = 0;
    /* stable counting by bucket */
    size_t* cnt = calloc(cap + 1, sizeof(size_t));
    for (size_t i = 0; i < n; ++i) { uint32_t h = (uint32_t) m->order[i]; h ^= h >> 16; cnt[(h & (cap - 1)) + 1]++; }
    for (size_t b = 0; b < cap; ++b) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < n; ++i) { uint32_t h = (uint32_t) m->order[i]; h ^= h >> 16; out[cnt[h & (cap - 1)]++] = m->order[i]; k++; }
    free(cnt);
    return k;
}

/* ------------------------------------------------------------------------------------------------ */
/* Quantizer object (fields of NQ/PnnQuantizer.java:17-33 and NQ/PnnLABQuantizer.java:18-22)          */
/* ------------------------------------------------------------------------------------------------ */
static const float coeffs[3][3] = {            /* NQ/PnnQuantizer.java:26-30 */
    {0.299f, 0.587f, 0.114f},
    {-0.14713f, -0.28886f, 0.436f},
    {0.615f, -0.51499f, -0.10001f}
};

struct nqo_quantizer {
    int kind;
    int alphaThreshold;
    int hasSemiTransparency;
    int m_transparentPixelIndex;
    int width, height;
    int32_t* pixels;
    int32_t m_transparentColor;
    double PR, PG, PB, PA;
    double ratio, weight;
    imap closestMap, nearestMap;
    /* LAB */
    int isNano;
    float* saliencies;
    imap pixelMap;
    int64_t rng;
    int64_t seed;            /* the injected seed (per-tile / per-pixel streams of the tiled restatement derive from it) */
    /* bookkeeping (not in the reference) */
    int no_cache;            /* cache-miss semantics for the tiled restatement / pure lookups */
    int n_bands;             /* > 1: the LAB histogram sums restart at every band (nqo_set_bands) and the partials are added in band order */
    int band_row[65];        /* first row of band b; band_row[n_bands] = height */
    int saliencies_partial;  /* the saliency map covers a tile-row range only (nqo_dither_tile_rows): dropped after the pass */
    int64_t frozen_distinct; /* >=0: value used in place of pixelMap.size() by the tiled BlueNoise weight */
    int texicab, quan_rt, maxbins, nMaxColors, paletteLength;
    int64_t distinct_after_hist;
    double t_stage[6];
    int64_t find_nn_calls;
};

nqo_quantizer* nqo_create(int kind, const int32_t* argb, int width, int height) {
    nqo_quantizer* q = calloc(1, sizeof *q);
    q->kind = kind;
    q->alphaThreshold = 0xF;
    q->m_transparentPixelIndex = -1;
    q->width = width; q->height = height;
    size_t n = (size_t) width * (size_t) height;
    q->pixels = malloc((n ? n : 1) * sizeof(int32_t));
    memcpy(q->pixels, argb, n * sizeof(int32_t));
    q->m_transparentColor = c_argb(0, BYTE_MAX, BYTE_MAX, BYTE_MAX);
    q->PR = 0.299; q->PG = 0.587; q->PB = 0.114; q->PA = .3333;
    q->ratio = .5; q->weight = 1;
    imap_init(&q->closestMap, 4 * sizeof(int32_t));
    imap_init(&q->nearestMap, sizeof(int16_t));
    imap_init(&q->pixelMap, sizeof(Lab));
    nqo_jrandom_seed(&q->rng, 0);
    q->frozen_distinct = -1;
    return q;
}
void nqo_destroy(nqo_quantizer* q) {
    if (!q) return;
    free(q->pixels); free(q->saliencies);
    imap_free(&q->closestMap); imap_free(&q->nearestMap); imap_free(&q->pixelMap);
    free(q);
}
/* multi-GPU band split: rows [row_start[b], row_start[b+1]) form band b; n_bands <= 1 switches it off */
void nqo_set_bands(nqo_quantizer* q, int n_bands, const int32_t* row_start) {
    q->n_bands = (n_bands > 1 && n_bands <= 64) ? n_bands : 0;
    for (int b = 0; b < q->n_bands; ++b) q->band_row[b] = row_start[b];
    if (q->n_bands) q->band_row[q->n_bands] = q->height;
}
void nqo_set_seed(nqo_quantizer* q, int64_t seed) { q->seed = seed; nqo_jrandom_seed(&q->rng, seed); }
void nqo_get_params(const nqo_quantizer* q, nqo_params* o) {
    memset(o, 0, sizeof *o);
    o->kind = q->kind; o->nMaxColors = q->nMaxColors; o->hasSemiTransparency = q->hasSemiTransparency;
    o->transparentPixelIndex = q->m_transparentPixelIndex; o->transparentColor = q->m_transparentColor;
    o->isNano = q->isNano; o->texicab = q->texicab; o->quan_rt = q->quan_rt; o->maxbins = q->maxbins;
    o->paletteLength = q->paletteLength;
    o->PR = q->PR; o->PG = q->PG; o->PB = q->PB; o->PA = q->PA; o->ratio = q->ratio; o->weight = q->weight;
    o->distinctColors = q->distinct_after_hist;
}
void nqo_set_params(nqo_quantizer* q, const nqo_params* p) {
    q->nMaxColors = p->nMaxColors; q->hasSemiTransparency = p->hasSemiTransparency;
    q->m_transparentPixelIndex = p->transparentPixelIndex; q->m_transparentColor = p->transparentColor;
    q->isNano = p->isNano; q->texicab = p->texicab; q->quan_rt = p->quan_rt; q->maxbins = p->maxbins;
    q->paletteLength = p->paletteLength;
    q->PR = p->PR; q->PG = p->PG; q->PB = p->PB; q->PA = p->PA; q->ratio = p->ratio; q->weight = p->weight;
    q->distinct_after_hist = p->distinctColors;
}
void nqo_get_stage_seconds(const nqo_quantizer* q, double* o) { memcpy(o, q->t_stage, sizeof q->t_stage); }
int64_t nqo_get_find_nn_calls(const nqo_quantizer* q) { return q->find_nn_calls; }

static inline int hasAlpha(const nqo_quantizer* q) { return q->m_transparentPixelIndex > -1; } /* NQ/PnnQuantizer.java:458-460 */

/* NQ/PnnLABQuantizer.java:34-42 */
static Lab getLab(nqo_quantizer* q, int32_t c) {
    Lab* got = imap_get(&q->pixelMap, c);
    if (got) return *got;
    Lab lab1 = RGB2LAB(c);
    imap_put(&q->pixelMap, c, &lab1);
    return lab1;
}

/* ------------------------------------------------------------------------------------------------ */
/* pnnquan: RGB  (NQ/PnnQuantizer.java:51-267)                                                       */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { double ac, rc, gc, bc; float cnt, err; int nn, fw, bk, tm, mtm; int present; } PnnbinRGB;

/* :57-116.  The scan itself, without side effects: bin index idx (the blue-noise row choice), its count n1 and means w[4] = a, r, g, b, the
 * forward list from `first`, `skip` = a bin to be passed over as if it had been unlinked (-1: none; the virtual-merge check only). */
static void find_nn_rgb_core(const nqo_quantizer* q, const PnnbinRGB* bins, int idx, int first, float n1, const double* w, int skip,
                             double* err_out, int* nn_out) {
    int nn = 0;
    double err = 1e100;
    const double wa = w[0], wr = w[1], wg = w[2], wb = w[3];
    int start = 0;
    if (TELL_BLUE_NOISE[idx & 4095] > 0)
        start = (q->PG < coeffs[0][1]) ? 3 : 1;
    const double ratio = q->ratio, PR = q->PR, PG = q->PG, PB = q->PB, PA = q->PA;
    for (int i = first; i != 0; i = bins[i].fw) {
        if (i == skip) continue;
        double n2 = bins[i].cnt, nerr2 = (n1 * n2) / (n1 + n2);
        if (nerr2 >= err) continue;
        double nerr = 0.0;
        if (q->hasSemiTransparency) {
            nerr += nerr2 * PA * sqr(bins[i].ac - wa);
            if (nerr >= err) continue;
        }
        nerr += nerr2 * (1 - ratio) * PR * sqr(bins[i].rc - wr);
        if (nerr >= err) continue;
        nerr += nerr2 * (1 - ratio) * PG * sqr(bins[i].gc - wg);
        if (nerr >= err) continue;
        nerr += nerr2 * (1 - ratio) * PB * sqr(bins[i].bc - wb);
        if (nerr >= err) continue;
        for (int j = start; j < 3; ++j) {
            nerr += nerr2 * ratio * sqr(coeffs[j][0] * (bins[i].rc - wr));
            if (nerr >= err) break;
            nerr += nerr2 * ratio * sqr(coeffs[j][1] * (bins[i].gc - wg));
            if (nerr >= err) break;
            nerr += nerr2 * ratio * sqr(coeffs[j][2] * (bins[i].bc - wb));
            if (nerr >= err) break;
        }
        err = nerr;           /* unconditional: reference quirk, SURVEY 8a row P7 */
        nn = i;
    }
    *err_out = err; *nn_out = nn;
}
static void find_nn_rgb(nqo_quantizer* q, PnnbinRGB* bins, int idx) {
    q->find_nn_calls++;
    PnnbinRGB* bin1 = &bins[idx];
    const double w[4] = {bin1->ac, bin1->rc, bin1->gc, bin1->bc};
    double err; int nn;
    find_nn_rgb_core(q, bins, idx, bin1->fw, bin1->cnt, w, -1, &err, &nn);
    bin1->err = (float) err;
    bin1->nn = nn;
}
/* (the switch and the counters of the virtual-merge self-check: defined with the LAB loop below) */
static int g_vm_check;
static int64_t g_vm_merges, g_vm_diffs;

/* (int) Math.cbrt(cnt) for an integer-valued count.  The double result on perfect cubes is libm dependent (glibc returns
 * 14.999999999999998 for 3375.0; HotSpot runs fdlibm, ART runs bionic/msun), so the truncation is unpinned exactly there.
 * Oracle and GPU both take the mathematically intended value: the exact integer cube root (DESIGN.md "unpinned"). */
static int icbrt_count(double c) {
    int q = j_d2i(cbrt(c));
    while ((double) (q + 1) * (q + 1) * (q + 1) <= c) ++q;
    while (q > 0 && (double) q * q * q > c) --q;
    return q;
}

/* getQuanFn: RGB :123-132, LAB NQ/PnnLABQuantizer.java:117-128 */
static float quan_fn(int kind, int nMaxColors, int quan_rt, float cnt) {
    if (kind == 0) {
        if (quan_rt > 0) {
            if (nMaxColors < 64) return (float) sqrt((double) cnt);
            return (float) j_d2i(sqrt((double) cnt));
        }
        if (quan_rt < 0) return (float) icbrt_count((double) cnt);
        return cnt;
    }
    if (quan_rt > 0) {
        if (quan_rt > 1) return (float) pow((double) cnt, 0.75);
        if (nMaxColors < 64) return (float) j_d2i(sqrt((double) cnt));
        return (float) sqrt((double) cnt);
    }
    return cnt;
}

static int pnnquan_rgb(nqo_quantizer* q, int nMaxColors, int32_t* palette) {
    int quan_rt = 1;
    const size_t N = (size_t) q->width * q->height;
    PnnbinRGB* bins = calloc(65536 + 1, sizeof *bins);
    double t0 = now_s();
    /* :140-154 histogram */
    for (size_t p = 0; p < N; ++p) {
        int32_t pixel = q->pixels[p];
        if (c_alpha(pixel) <= q->alphaThreshold) pixel = q->m_transparentColor;
        int index = nqo_get_color_index(pixel, q->hasSemiTransparency, nMaxColors < 64 || q->m_transparentPixelIndex >= 0);
        PnnbinRGB* tb = &bins[index];
        tb->present = 1;
        tb->ac += c_alpha(pixel); tb->rc += c_red(pixel); tb->gc += c_green(pixel); tb->bc += c_blue(pixel);
        tb->cnt++;
    }
    /* :157-170 compaction + means */
    int maxbins = 0;
    for (int i = 0; i < 65536; ++i) {
        if (!bins[i].present) continue;
        float d = 1.0f / bins[i].cnt;
        bins[i].ac *= d; bins[i].rc *= d; bins[i].gc *= d; bins[i].bc *= d;
        bins[maxbins++] = bins[i];
    }
    /* slots >= maxbins keep stale copies in Java too (objects); fw/bk of the compacted ones start at 0 */
    q->t_stage[1] += now_s() - t0; t0 = now_s();
    if (nMaxColors < 16) quan_rt = -1;
    q->weight = fmin(0.9, nMaxColors * 1.0 / maxbins);
    if (q->weight < .04 && q->PG >= coeffs[0][1]) {
        q->PR = q->PG = q->PB = q->PA = 1;
        if (nMaxColors >= 64) quan_rt = 0;
    }
    q->quan_rt = quan_rt; q->maxbins = maxbins;
    int j = 0;
    for (; j < maxbins - 1; ++j) {
        bins[j].fw = j + 1;
        bins[j + 1].bk = j;
        bins[j].cnt = quan_fn(0, nMaxColors, quan_rt, bins[j].cnt);
    }
    bins[j].cnt = quan_fn(0, nMaxColors, quan_rt, bins[j].cnt);

    int h, l, l2;
    int* heap = calloc(65536 + 1, sizeof(int));
    /* :196-207 */
    for (int i = 0; i < maxbins; i++) {
        find_nn_rgb(q, bins, i);
        float err = bins[i].err;
        for (l = ++heap[0]; l > 1; l = l2) {
            l2 = l >> 1;
            if (bins[h = heap[l2]].err <= err) break;
            heap[l] = h;
        }
        heap[l] = i;
    }
    q->t_stage[2] += now_s() - t0; t0 = now_s();
    /* :210-255 */
    int extbins = maxbins - nMaxColors;
    for (int i = 0; i < extbins;) {
        PnnbinRGB* tb;
        for (;;) {
            int b1 = heap[1];
            tb = &bins[b1];
            if ((tb->tm >= tb->mtm) && (bins[tb->nn].mtm <= tb->tm)) break;
            if (tb->mtm == 0xFFFF) b1 = heap[1] = heap[heap[0]--];
            else { find_nn_rgb(q, bins, b1); tb->tm = i; }
            float err = bins[b1].err;
            for (l = 1; (l2 = l + l) <= heap[0]; l = l2) {
                if ((l2 < heap[0]) && (bins[heap[l2]].err > bins[heap[l2 + 1]].err)) ++l2;
                if (err <= bins[h = heap[l2]].err) break;
                heap[l] = h;
            }
            heap[l] = b1;
        }
        PnnbinRGB* nb = &bins[tb->nn];
        float n1 = tb->cnt, n2 = nb->cnt;
        float d = 1.0f / (n1 + n2);
        double v_err = 0; int v_nn = 0;
        if (g_vm_check) {      /* the scan before the merge: merged count and means, the neighbour passed over (see the LAB loop) */
            const double m[4] = {d * (float) j_round(n1 * tb->ac + n2 * nb->ac), d * (float) j_round(n1 * tb->rc + n2 * nb->rc),
                                 d * (float) j_round(n1 * tb->gc + n2 * nb->gc), d * (float) j_round(n1 * tb->bc + n2 * nb->bc)};
            find_nn_rgb_core(q, bins, (int) (tb - bins), tb->fw, n1 + n2, m, tb->nn, &v_err, &v_nn);
        }
        tb->ac = d * (float) j_round(n1 * tb->ac + n2 * nb->ac);   /* float * long -> float */
        tb->rc = d * (float) j_round(n1 * tb->rc + n2 * nb->rc);
        tb->gc = d * (float) j_round(n1 * tb->gc + n2 * nb->gc);
        tb->bc = d * (float) j_round(n1 * tb->bc + n2 * nb->bc);
        tb->cnt += n2;
        tb->mtm = ++i;
        bins[nb->bk].fw = nb->fw;
        bins[nb->fw].bk = nb->bk;
        nb->mtm = 0xFFFF;
        if (g_vm_check) {      /* ... and the scan the loop's next turn will make */
            const double m[4] = {tb->ac, tb->rc, tb->gc, tb->bc};
            double r_err; int r_nn;
            find_nn_rgb_core(q, bins, (int) (tb - bins), tb->fw, tb->cnt, m, -1, &r_err, &r_nn);
            ++g_vm_merges;
            if ((float) r_err != (float) v_err || r_nn != v_nn) ++g_vm_diffs;
        }
    }
    q->t_stage[3] += now_s() - t0;
    /* :258-266 */
    int plen = extbins > 0 ? nMaxColors : maxbins;
    int k = 0;
    for (int i = 0; k < plen; ++k) {
        palette[k] = c_argb(j_d2i(bins[i].ac), j_d2i(bins[i].rc), j_d2i(bins[i].gc), j_d2i(bins[i].bc));
        i = bins[i].fw;
    }
    free(heap); free(bins);
    return plen;
}

/* ------------------------------------------------------------------------------------------------ */
/* pnnquan: LAB  (NQ/PnnLABQuantizer.java:28-327)                                                    */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { float ac, Lc, Ac, Bc, err, cnt; int nn, fw, bk, tm, mtm; int present; } PnnbinLAB;

/* :44-115.  The scan itself, without side effects: count n1 and means lab1 of the bin, the forward list from `start`, `skip` = a bin to be
 * passed over as if it had been unlinked (-1: none; only the virtual-merge check below uses it). */
static void find_nn_lab_core(const nqo_quantizer* q, const PnnbinLAB* bins, int start, float n1, Lab lab1, int skip, int texicab,
                             double* err_out, int* nn_out) {
    int nn = 0;
    double err = 1e100;
    const double ratio = q->ratio;
    for (int i = start; i != 0; i = bins[i].fw) {
        if (i == skip) continue;
        float n2 = bins[i].cnt;
        double nerr2 = (n1 * n2) / (n1 + n2);
        if (nerr2 >= err) continue;
        Lab lab2; lab2.alpha = bins[i].ac; lab2.L = bins[i].Lc; lab2.A = bins[i].Ac; lab2.B = bins[i].Bc;
        double alphaDiff = q->hasSemiTransparency ? sqr((double) (lab2.alpha - lab1.alpha)) / exp(1.75) : 0;
        double nerr = nerr2 * alphaDiff;
        if (nerr >= err) continue;
        if (!texicab) {
            nerr += (1 - ratio) * nerr2 * sqr((double) (lab2.L - lab1.L));
            if (nerr >= err) continue;
            nerr += (1 - ratio) * nerr2 * sqr((double) (lab2.A - lab1.A));
            if (nerr >= err) continue;
            nerr += (1 - ratio) * nerr2 * sqr((double) (lab2.B - lab1.B));
        } else {
            nerr += (1 - ratio) * nerr2 * (double) fabsf(lab2.L - lab1.L);
            if (nerr >= err) continue;
            nerr += (1 - ratio) * nerr2 * sqrt(sqr((double) (lab2.A - lab1.A)) + sqr((double) (lab2.B - lab1.B)));
        }
        if (nerr > err) continue;
        float deltaL = L_prime_div_k_L_S_L(lab1, lab2);
        nerr += ratio * nerr2 * sqr((double) deltaL);
        if (nerr > err) continue;
        double a1Prime, a2Prime, CPrime1, CPrime2;
        float deltaC = C_prime_div_k_L_S_L(lab1, lab2, &a1Prime, &a2Prime, &CPrime1, &CPrime2);
        nerr += ratio * nerr2 * sqr((double) deltaC);
        if (nerr > err) continue;
        double barCPrime, barhPrime;
        float deltaH = H_prime_div_k_L_S_L(lab1, lab2, a1Prime, a2Prime, CPrime1, CPrime2, &barCPrime, &barhPrime);
        nerr += ratio * nerr2 * sqr((double) deltaH);
        if (nerr > err) continue;
        nerr += ratio * nerr2 * (double) R_T(barCPrime, barhPrime, deltaC, deltaH);
        if (nerr > err) continue;
        err = nerr;
        nn = i;
    }
    *err_out = err; *nn_out = nn;
}
static void find_nn_lab(nqo_quantizer* q, PnnbinLAB* bins, int idx, int texicab) {
    q->find_nn_calls++;
    PnnbinLAB* bin1 = &bins[idx];
    Lab lab1; lab1.alpha = bin1->ac; lab1.L = bin1->Lc; lab1.A = bin1->Ac; lab1.B = bin1->Bc;
    double err; int nn;
    find_nn_lab_core(q, bins, bin1->fw, bin1->cnt, lab1, -1, texicab, &err, &nn);
    bin1->err = (float) err;
    bin1->nn = nn;
}
/* Self-check of the premise behind the GPU's "virtual merge" (csrc/nq_merge.inc): the find_nn that follows the merge of a bin with its
 * neighbour equals a scan made BEFORE that merge with the merged count and means and with the neighbour passed over.  When switched on,
 * every merge of the LAB loop (and of the RGB loop, for which the GPU has no virtual merge yet) computes both and counts the merges and the differences (bit for bit: err as float, nn). */
/* (g_vm_check, g_vm_merges, g_vm_diffs: declared with the RGB loop above) */
void nqo_debug_virtual_merge(int on, int64_t* out2) {
    if (out2) { out2[0] = g_vm_merges; out2[1] = g_vm_diffs; }
    g_vm_check = on; g_vm_merges = g_vm_diffs = 0;
}

static int pnnquan_lab(nqo_quantizer* q, int nMaxColors, int32_t* palette) {
    int quan_rt = 1;
    const size_t N = (size_t) q->width * q->height;
    PnnbinLAB* bins = calloc(65536 + 1, sizeof *bins);
    free(q->saliencies);
    q->saliencies = nMaxColors >= 128 ? NULL : calloc(N ? N : 1, sizeof(float));
    float saliencyBase = .1f;
    double t0 = now_s();
    /* :139-157.  Banded restatement (n_bands > 1, the multi-GPU split of SURVEY 8e): the float sums of a bin restart at every band
     * and the band partials are added in band order, all in float32 -- exactly what nq_palette_from_histograms_device does with the
     * gathered per-band histograms; the count is exact and saturates at 2^24 like `cnt += 1.0f`.  n_bands <= 1: the reference. */
    const int nb = q->n_bands > 1 ? q->n_bands : 1;
    PnnbinLAB* part = nb > 1 ? calloc(65536 + 1, sizeof *part) : NULL;
    double* exact_cnt = nb > 1 ? calloc(65536 + 1, sizeof(double)) : NULL;
    for (int b = 0; b < nb; ++b) {
        const size_t p_lo = nb > 1 ? (size_t) q->band_row[b] * q->width : 0, p_hi = nb > 1 ? (size_t) q->band_row[b + 1] * q->width : N;
        PnnbinLAB* acc = nb > 1 ? part : bins;
        if (nb > 1) memset(part, 0, (65536 + 1) * sizeof *part);
        for (size_t p = p_lo; p < p_hi; ++p) {
            int32_t pixel = q->pixels[p];
            if (c_alpha(pixel) <= q->alphaThreshold) pixel = q->m_transparentColor;
            int index = nqo_get_color_index(pixel, q->hasSemiTransparency, nMaxColors < 64 || q->m_transparentPixelIndex >= 0);
            Lab lab1 = getLab(q, pixel);
            PnnbinLAB* tb = &acc[index];
            tb->present = 1;
            tb->ac += lab1.alpha; tb->Lc += lab1.L; tb->Ac += lab1.A; tb->Bc += lab1.B;
            tb->cnt += 1.0f;
            if (nb > 1) exact_cnt[index] += 1.0;
            if (q->saliencies)
                q->saliencies[p] = saliencyBase + (1 - saliencyBase) * lab1.L / 100.0f * lab1.alpha / 255.0f;
        }
        if (nb > 1)
            for (int i = 0; i < 65536; ++i) {
                if (!part[i].present) continue;
                PnnbinLAB* tb = &bins[i];
                if (!tb->present) { tb->present = 1; tb->ac = part[i].ac; tb->Lc = part[i].Lc; tb->Ac = part[i].Ac; tb->Bc = part[i].Bc; }
                else { tb->ac += part[i].ac; tb->Lc += part[i].Lc; tb->Ac += part[i].Ac; tb->Bc += part[i].Bc; }
                tb->cnt = (float) (exact_cnt[i] > 16777216.0 ? 16777216.0 : exact_cnt[i]);
            }
    }
    free(part); free(exact_cnt);
    q->distinct_after_hist = (int64_t) q->pixelMap.n;
    /* :160-173 */
    int maxbins = 0;
    for (int i = 0; i < 65536; ++i) {
        if (!bins[i].present) continue;
        float d = 1.0f / bins[i].cnt;
        bins[i].ac *= d; bins[i].Lc *= d; bins[i].Ac *= d; bins[i].Bc *= d;
        bins[maxbins++] = bins[i];
    }
    q->t_stage[1] += now_s() - t0; t0 = now_s();
    /* :175-191 */
    double proportional = sqr(nMaxColors) / maxbins;
    if ((q->m_transparentPixelIndex >= 0 || q->hasSemiTransparency) && nMaxColors < 32) quan_rt = -1;
    q->weight = fmin(0.9, nMaxColors * 1.0 / maxbins);
    q->isNano = q->weight <= .015;
    const double weight = q->weight;
    if ((nMaxColors < 16 && weight < .0075) || weight < .001 || (weight > .0015 && weight < .0022)) quan_rt = 2;
    if (weight < .04 && q->PG < 1 && q->PG >= coeffs[0][1]) {
        if (nMaxColors >= 64) quan_rt = 0;
    }
    if (nMaxColors > 16 && nMaxColors < 64) {
        double weightB = nMaxColors / 8000.0;
        if (fabs(weightB - weight) < .001) quan_rt = 2;
    }
    q->maxbins = maxbins;
    /* :193-206 few distinct colours: palette = pixelMap.keySet() in HashMap order */
    if ((int64_t) q->pixelMap.n <= nMaxColors) {
        int32_t* keys = malloc((q->pixelMap.n + 1) * sizeof(int32_t));
        size_t nk = imap_java_keyset(&q->pixelMap, keys);
        int k = 0;
        for (size_t t = 0; t < nk; ++t) {
            int32_t pixel = keys[t];
            palette[k++] = pixel;
            if (k > 1 && c_alpha(pixel) == 0) { palette[k - 1] = palette[0]; palette[0] = pixel; }
        }
        free(keys); free(bins);
        q->quan_rt = quan_rt; q->texicab = 0;
        return k;
    }
    /* :208-217 */
    int j = 0;
    for (; j < maxbins - 1; ++j) {
        bins[j].fw = j + 1;
        bins[j + 1].bk = j;
        bins[j].cnt = quan_fn(1, nMaxColors, quan_rt, bins[j].cnt);
    }
    bins[j].cnt = quan_fn(1, nMaxColors, quan_rt, bins[j].cnt);
    /* :219-241 */
    const int texicab = proportional > .0225 && !q->hasSemiTransparency;
    if (q->hasSemiTransparency) q->ratio = .5;
    else if (quan_rt != 0 && nMaxColors < 64) {
        if (proportional > .018 && proportional < .022) q->ratio = fmin(1.0, proportional + weight * exp(3.13));
        else if (proportional > .1) q->ratio = fmin(1.0, 1.0 - weight);
        else if (proportional > .04) q->ratio = fmin(1.0, weight * exp(1.56));
        else if (proportional > .025 && (weight < .002 || weight > .0022)) q->ratio = fmin(1.0, proportional + weight * exp(3.66));
        else q->ratio = fmin(1.0, proportional + weight * exp(1.718));
    }
    else if (nMaxColors > 256) q->ratio = fmin(1.0, 1 - 1.0 / proportional);
    else q->ratio = fmin(1.0, 1 - weight * .7);
    if (!q->hasSemiTransparency && quan_rt < 0) q->ratio = fmin(1.0, weight * exp(3.13));
    q->quan_rt = quan_rt; q->texicab = texicab;

    int h, l, l2;
    int* heap = calloc(65536 + 1, sizeof(int));
    /* :246-257 */
    for (int i = 0; i < maxbins; ++i) {
        find_nn_lab(q, bins, i, texicab);
        float err = bins[i].err;
        for (l = ++heap[0]; l > 1; l = l2) {
            l2 = l >> 1;
            if (bins[h = heap[l2]].err <= err) break;
            heap[l] = h;
        }
        heap[l] = i;
    }
    q->t_stage[2] += now_s() - t0; t0 = now_s();
    /* :259-264 */
    if (quan_rt > 0 && nMaxColors < 64 && proportional > .035 && proportional < .1) {
        const int dir = proportional > .04 ? 1 : -1;
        const double margin = dir > 0 ? .002 : .0025;
        const double delta = weight > margin && weight < .003 ? 1.872 : 1.632;
        q->ratio = fmin(1.0, proportional + dir * weight * exp(delta));
    }
    /* :267-312 */
    int extbins = maxbins - nMaxColors;
    for (int i = 0; i < extbins;) {
        PnnbinLAB* tb;
        for (;;) {
            int b1 = heap[1];
            tb = &bins[b1];
            if ((tb->tm >= tb->mtm) && (bins[tb->nn].mtm <= tb->tm)) break;
            if (tb->mtm == 0xFFFF) b1 = heap[1] = heap[heap[0]--];
            else { find_nn_lab(q, bins, b1, texicab); tb->tm = i; }
            float err = bins[b1].err;
            for (l = 1; (l2 = l + l) <= heap[0]; l = l2) {
                if ((l2 < heap[0]) && (bins[heap[l2]].err > bins[heap[l2 + 1]].err)) ++l2;
                if (err <= bins[h = heap[l2]].err) break;
                heap[l] = h;
            }
            heap[l] = b1;
        }
        PnnbinLAB* nb = &bins[tb->nn];
        float n1 = tb->cnt, n2 = nb->cnt;
        float d = 1.0f / (n1 + n2);
        double v_err = 0; int v_nn = 0;
        if (g_vm_check) {      /* the scan as a helper of the GPU's merge team makes it: before the merge, on the unchanged lists */
            Lab m; m.alpha = d * (n1 * tb->ac + n2 * nb->ac); m.L = d * (n1 * tb->Lc + n2 * nb->Lc);
            m.A = d * (n1 * tb->Ac + n2 * nb->Ac); m.B = d * (n1 * tb->Bc + n2 * nb->Bc);
            find_nn_lab_core(q, bins, tb->fw, n1 + n2, m, tb->nn, texicab, &v_err, &v_nn);
        }
        tb->ac = d * (n1 * tb->ac + n2 * nb->ac);
        tb->Lc = d * (n1 * tb->Lc + n2 * nb->Lc);
        tb->Ac = d * (n1 * tb->Ac + n2 * nb->Ac);
        tb->Bc = d * (n1 * tb->Bc + n2 * nb->Bc);
        tb->cnt += n2;
        tb->mtm = ++i;
        bins[nb->bk].fw = nb->fw;
        bins[nb->fw].bk = nb->bk;
        nb->mtm = 0xFFFF;
        if (g_vm_check) {      /* ... and the scan the loop's next turn will make */
            Lab m; m.alpha = tb->ac; m.L = tb->Lc; m.A = tb->Ac; m.B = tb->Bc;
            double r_err; int r_nn;
            find_nn_lab_core(q, bins, tb->fw, tb->cnt, m, -1, texicab, &r_err, &r_nn);
            ++g_vm_merges;
            if ((float) r_err != (float) v_err || r_nn != v_nn) ++g_vm_diffs;
        }
    }
    q->t_stage[3] += now_s() - t0;
    /* :315-326 */
    int plen = extbins > 0 ? nMaxColors : maxbins;
    int k = 0, status = 0;
    for (int i = 0; k < plen; ++k) {
        Lab lab1;
        lab1.alpha = (float) j_d2i(bins[i].ac);
        lab1.L = bins[i].Lc; lab1.A = bins[i].Ac; lab1.B = bins[i].Bc;
        status |= LAB2RGB(lab1, &palette[k]);
        i = bins[i].fw;
    }
    free(heap); free(bins);
    return status ? -1 : plen;
}

int nqo_pnnquan(nqo_quantizer* q, int nMaxColors, int32_t* out_palette) {
    q->nMaxColors = nMaxColors;
    int k = q->kind == 0 ? pnnquan_rgb(q, nMaxColors, out_palette) : pnnquan_lab(q, nMaxColors, out_palette);
    q->paletteLength = k;
    return k;
}

/* ------------------------------------------------------------------------------------------------ */
/* nearest / closest lookups                                                                          */
/* ------------------------------------------------------------------------------------------------ */
/* NQ/PnnQuantizer.java:269-311 */
static int16_t nearest_rgb(nqo_quantizer* q, const int32_t* palette, int K, int32_t c, int pos) {
    (void) pos;
    const int32_t offset = q->weight > .015 ? c : nqo_get_color_index(c, q->hasSemiTransparency, q->m_transparentPixelIndex >= 0);
    if (!q->no_cache) { int16_t* got = imap_get(&q->nearestMap, offset); if (got) return *got; }
    int16_t k = 0;
    if (c_alpha(c) <= q->alphaThreshold) c = q->m_transparentColor;
    if (K > 2 && hasAlpha(q) && c_alpha(c) > q->alphaThreshold) k = 1;
    double pr = q->PR, pg = q->PG, pb = q->PB, pa = q->PA;
    if (K < 3) pr = pg = pb = pa = 1;
    double mindist = 2147483647;
    for (int16_t i = k; i < K; ++i) {
        int32_t c2 = palette[i];
        double curdist = pa * sqr(c_alpha(c2) - c_alpha(c));
        if (curdist > mindist) continue;
        curdist += pr * sqr(c_red(c2) - c_red(c));
        if (curdist > mindist) continue;
        curdist += pg * sqr(c_green(c2) - c_green(c));
        if (curdist > mindist) continue;
        curdist += pb * sqr(c_blue(c2) - c_blue(c));
        if (curdist > mindist) continue;
        mindist = curdist;
        k = i;
    }
    if (!q->no_cache) imap_put(&q->nearestMap, offset, &k);
    return k;
}

/* NQ/PnnQuantizer.java:313-375; out4 (nullable) receives the closest[] tuple */
static int16_t closest_rgb(nqo_quantizer* q, const int32_t* palette, int K, int32_t c, int pos, int32_t* out4) {
    int16_t k = 0;
    if (c_alpha(c) <= q->alphaThreshold) { if (out4) out4[0] = out4[1] = out4[2] = out4[3] = -1; return nearest_rgb(q, palette, K, c, pos); }
    const int32_t offset = q->weight > .015 ? c : nqo_get_color_index(c, q->hasSemiTransparency, q->m_transparentPixelIndex >= 0);
    int32_t closest[4];
    int32_t* got = q->no_cache ? NULL : imap_get(&q->closestMap, c);
    if (got) memcpy(closest, got, sizeof closest);
    else {
        closest[0] = closest[1] = 0;
        closest[2] = closest[3] = INT32_MAX;
        double pr = q->PR, pg = q->PG, pb = q->PB, pa = q->PA;
        if (K < 3) pr = pg = pb = pa = 1;
        for (; k < K; ++k) {
            int32_t c2 = palette[k];
            double err = pr * sqr(c_red(c2) - c_red(c));
            if (err >= closest[3]) continue;
            err += pg * sqr(c_green(c2) - c_green(c));
            if (err >= closest[3]) continue;
            err += pb * sqr(c_blue(c2) - c_blue(c));
            if (err >= closest[3]) continue;
            if (q->hasSemiTransparency) err += pa * sqr(c_alpha(c2) - c_alpha(c));
            if (err < closest[2]) {
                closest[1] = closest[0]; closest[3] = closest[2];
                closest[0] = k; closest[2] = j_d2i(err);
            } else if (err < closest[3]) {
                closest[1] = k; closest[3] = j_d2i(err);
            }
        }
        if (closest[3] == INT32_MAX) closest[1] = closest[0];
        if (!q->no_cache) imap_put(&q->closestMap, offset, closest);
    }
    if (out4) memcpy(out4, closest, sizeof closest);
    int MAX_ERR = K << 2;
    int idx = (pos + 1) % 2;
    if (closest[3] * .67 < (closest[3] - closest[2])) idx = 0;
    else if (closest[0] > closest[1]) idx = pos % 2;
    if (closest[idx + 2] >= MAX_ERR || (hasAlpha(q) && closest[idx] == 0))
        return nearest_rgb(q, palette, K, c, pos);
    return (int16_t) closest[idx];
}

/* NQ/PnnLABQuantizer.java:330-404 */
static int16_t nearest_lab(nqo_quantizer* q, const int32_t* palette, int K, int32_t c, int pos) {
    (void) pos;
    const int32_t offset = !q->isNano ? c : nqo_get_color_index(c, q->hasSemiTransparency, q->m_transparentPixelIndex >= 0);
    if (!q->no_cache) { int16_t* got = imap_get(&q->nearestMap, offset); if (got) return *got; }
    int16_t k = 0;
    if (c_alpha(c) <= q->alphaThreshold) c = q->m_transparentColor;
    if (K > 2 && hasAlpha(q) && c_alpha(c) > q->alphaThreshold) k = 1;
    double mindist = 2147483647;
    Lab lab1 = getLab(q, c);
    for (int16_t i = k; i < K; ++i) {
        int32_t c2 = palette[i];
        double curdist = q->hasSemiTransparency ? sqr(c_alpha(c2) - c_alpha(c)) / exp(1.5) : 0;
        if (curdist > mindist) continue;
        Lab lab2 = getLab(q, c2);
        if (K <= 4) {
            curdist = sqr(c_red(c2) - c_red(c)) + sqr(c_green(c2) - c_green(c)) + sqr(c_blue(c2) - c_blue(c));
            if (q->hasSemiTransparency) curdist += sqr(c_alpha(c2) - c_alpha(c));
        }
        else if (q->hasSemiTransparency || K < 16) {
            curdist += sqr((double) (lab2.L - lab1.L));
            if (curdist > mindist) continue;
            curdist += sqr((double) (lab2.A - lab1.A));
            if (curdist > mindist) continue;
            curdist += sqr((double) (lab2.B - lab1.B));
        }
        else if (K > 32) {
            curdist += (double) fabsf(lab2.L - lab1.L);
            if (curdist > mindist) continue;
            curdist += sqrt(sqr((double) (lab2.A - lab1.A)) + sqr((double) (lab2.B - lab1.B)));
        }
        else {
            float deltaL = L_prime_div_k_L_S_L(lab1, lab2);
            curdist += sqr((double) deltaL);
            if (curdist > mindist) continue;
            double a1Prime, a2Prime, CPrime1, CPrime2;
            float deltaC = C_prime_div_k_L_S_L(lab1, lab2, &a1Prime, &a2Prime, &CPrime1, &CPrime2);
            curdist += sqr((double) deltaC);
            if (curdist > mindist) continue;
            double barCPrime, barhPrime;
            float deltaH = H_prime_div_k_L_S_L(lab1, lab2, a1Prime, a2Prime, CPrime1, CPrime2, &barCPrime, &barhPrime);
            curdist += sqr((double) deltaH);
            if (curdist > mindist) continue;
            curdist += (double) R_T(barCPrime, barhPrime, deltaC, deltaH);
        }
        if (curdist > mindist) continue;
        mindist = curdist;
        k = i;
    }
    if (!q->no_cache) imap_put(&q->nearestMap, offset, &k);
    return k;
}

/* event counters (test diagnostics: how often each branch of the per-pixel pass is taken) */
static int64_t g_dbg[16];
void nqo_debug_counters(int64_t* out16, int reset) { if (out16) memcpy(out16, g_dbg, sizeof g_dbg); if (reset) memset(g_dbg, 0, sizeof g_dbg); }

/* NQ/PnnLABQuantizer.java:407-474 */
static int16_t closest_lab(nqo_quantizer* q, const int32_t* palette, int K, int32_t c, int pos, int32_t* out4, int64_t* rng) {
    if (c_alpha(c) <= q->alphaThreshold) { if (out4) out4[0] = out4[1] = out4[2] = out4[3] = -1; return nearest_lab(q, palette, K, c, pos); }
    const int32_t offset = !q->isNano ? c : nqo_get_color_index(c, q->hasSemiTransparency, q->m_transparentPixelIndex >= 0);
    int32_t closest[4];
    int32_t* got = q->no_cache ? NULL : imap_get(&q->closestMap, c);
    if (got) memcpy(closest, got, sizeof closest);
    else {
        closest[0] = closest[1] = 0;
        closest[2] = closest[3] = INT32_MAX;
        const double PR = q->PR, PG = q->PG, PB = q->PB, PA = q->PA, ratio = q->ratio;
        for (int16_t k = 0; k < K; ++k) {
            int32_t c2 = palette[k];
            double err = PR * (1 - ratio) * sqr(c_red(c2) - c_red(c));
            if (err >= closest[3]) continue;
            err += PG * (1 - ratio) * sqr(c_green(c2) - c_green(c));
            if (err >= closest[3]) continue;
            err += PB * (1 - ratio) * sqr(c_blue(c2) - c_blue(c));
            if (err >= closest[3]) continue;
            if (q->hasSemiTransparency) err += PA * sqr(c_alpha(c2) - c_alpha(c));
            for (int i = 0; i < 3; ++i) {
                err += ratio * sqr(coeffs[i][0] * (c_red(c2) - c_red(c)));
                if (err >= closest[3]) break;
                err += ratio * sqr(coeffs[i][1] * (c_green(c2) - c_green(c)));
                if (err >= closest[3]) break;
                err += ratio * sqr(coeffs[i][2] * (c_blue(c2) - c_blue(c)));
                if (err >= closest[3]) break;
            }
            if (err < closest[2]) {
                closest[1] = closest[0]; closest[3] = closest[2];
                closest[0] = k; closest[2] = j_d2i(err);
            } else if (err < closest[3]) {
                closest[1] = k; closest[3] = j_d2i(err);
            }
        }
        if (closest[3] == INT32_MAX) closest[1] = closest[0];
        if (!q->no_cache) imap_put(&q->closestMap, offset, closest);
    }
    if (out4) { memcpy(out4, closest, sizeof closest); if (!rng) return 0; }
    int idx = 1;
    g_dbg[1]++; if (closest[2] == 0) g_dbg[7]++; if (closest[2] >= K) g_dbg[8]++;
    if (closest[2] == 0 || (nqo_jrandom_next_int_bound(rng, 32767) % i_add_wrap(closest[3], closest[2])) <= closest[3])
        idx = 0;
    int MAX_ERR = K;
    if (closest[idx + 2] >= MAX_ERR || closest[idx] == 0 || c_alpha(palette[closest[idx]]) < c_alpha(c)) {
        g_dbg[2]++;
        return nearest_lab(q, palette, K, c, pos);
    }
    return (int16_t) closest[idx];
}

/* the Ditherable built by getDitherFn: RGB NQ/PnnQuantizer.java:377-391, LAB NQ/PnnLABQuantizer.java:476-490 */
typedef struct { nqo_quantizer* q; int dither; int64_t* rng; } ditherable;
static int16_t ditherable_nearest(ditherable* d, const int32_t* palette, int K, int32_t c, int pos) {
    nqo_quantizer* q = d->q;
    if (q->kind == 0) {
        if (d->dither) return nearest_rgb(q, palette, K, c, pos);
        return closest_rgb(q, palette, K, c, pos, NULL);
    }
    if (K <= 4) return nearest_lab(q, palette, K, c, pos);
    return closest_lab(q, palette, K, c, pos, NULL, d->rng);
}

void nqo_nearest_index(nqo_quantizer* q, const int32_t* palette, int K, const int32_t* colors, int64_t M, int16_t* out) {
    int save = q->no_cache; q->no_cache = 1;
    for (int64_t i = 0; i < M; ++i)
        out[i] = q->kind == 0 ? nearest_rgb(q, palette, K, colors[i], 0) : nearest_lab(q, palette, K, colors[i], 0);
    q->no_cache = save;
    imap_clear(&q->pixelMap);
}
void nqo_closest_tuple(nqo_quantizer* q, const int32_t* palette, int K, const int32_t* colors, int64_t M, int32_t* out) {
    int save = q->no_cache; q->no_cache = 1;
    for (int64_t i = 0; i < M; ++i) {
        if (q->kind == 0) closest_rgb(q, palette, K, colors[i], 0, out + 4 * i);
        else closest_lab(q, palette, K, colors[i], 0, out + 4 * i, NULL);
    }
    q->no_cache = save;
    imap_clear(&q->pixelMap);
}

/* ------------------------------------------------------------------------------------------------ */
/* GilbertCurve (NQ/GilbertCurve.java)                                                               */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { double yDiff; float p[4]; } ErrorBox;    /* :16-32, p in r,g,b,a order */
#define QCAP 64
typedef struct {
    int8_t ditherMax, DITHER_MAX;
    float beta;
    float weights[QCAP]; int nweights;
    int dither, hasAlpha, sortedByYDiff;
    int width, height;
    double weight;
    const int32_t* pixels; const int32_t* palette; int K;
    int32_t* qPixels;
    ditherable* dth;
    const float* saliencies;
    ErrorBox eq[QCAP]; int qhead, qsize;   /* ArrayDeque (ring) or PriorityQueue (array heap from index 0) */
    int margin, thresold;
    int x0, y0;                            /* tile origin (0,0 for the reference's whole-image curve) */
} GilbertCurve;

/* constructor :50-112.  NB the parameter `weight` (signed) shadows the field (abs value, :61). */
static void gc_init(GilbertCurve* g, int width, int height, const int32_t* image, const int32_t* palette, int K,
                    int32_t* qPixels, ditherable* dth, const float* saliencies, double weight, int dither) {
    memset(g, 0, sizeof *g);
    g->width = width; g->height = height; g->pixels = image; g->palette = palette; g->K = K; g->qPixels = qPixels;
    g->dth = dth;
    g->hasAlpha = weight < 0;
    g->saliencies = saliencies;
    g->dither = dither;
    g->weight = fabs(weight);
    g->margin = weight < .0025 ? 12 : weight < .004 ? 8 : 6;
    g->sortedByYDiff = K > 128 && weight >= .02 && (!g->hasAlpha || weight < .18);
    float beta = K > 4 ? (float) (.6f - .00625f * K) : 1;
    if (K > 4) {
        double boundary = .005 - .0000625 * K;
        beta = (float) (weight > boundary ? .25 : fmin(1.5, beta + K * weight));
        if (K > 16 && K <= 32 && weight < .003) beta += .075f;
        else if (weight < .0015 || (K > 32 && K < 256)) beta += .1f;
        if ((K >= 64 && (weight > .012 && weight < .0125)) || (weight > .025 && weight < .03)) beta += .05f;
        else if (K > 32 && K < 64 && weight < .015) beta = .55f;
        else if (K > 16 && K <= 32 && weight <= .005) beta += (float) (.05 + weight * K);
    }
    else beta *= .95f;
    if (K > 64 || (K > 4 && weight > .02)) beta *= .4f;
    if (K > 64 && weight < .02) beta = .18f;
    int8_t DITHER_MAX = weight < .015 ? ((weight > .0025) ? (int8_t) 25 : 16) : 9;
    if (weight > .99) { beta = (float) weight; DITHER_MAX = 25; }
    double edge = g->hasAlpha ? 1 : exp(weight) - .25;
    double deviation = weight > .002 ? -.25 : 1;
    int8_t ditherMax = (g->hasAlpha || DITHER_MAX > 9) ? j_d2b(sqr(sqrt((double) DITHER_MAX) + edge * deviation))
                                                       : j_d2b(DITHER_MAX * (saliencies != NULL ? 2 : J_E));
    const int density = K > 16 ? 3200 : 1500;
    if (K / weight > 5000 && (weight > .045 || (weight > .01 && K < 64))) ditherMax = j_d2b(sqr(5 + edge));
    else if (weight < .03 && K / weight < density && K >= 16 && K < 256) ditherMax = j_d2b(sqr(5 + edge));
    g->thresold = DITHER_MAX > 9 ? -112 : -64;
    g->beta = beta; g->DITHER_MAX = DITHER_MAX; g->ditherMax = ditherMax;
    g->nweights = 0;
}
float nqo_gilbert_params(int K, double weight, int hasSaliencies, int32_t* o) {
    GilbertCurve g; float dummy = 0;
    gc_init(&g, 1, 1, NULL, NULL, K, NULL, NULL, hasSaliencies ? &dummy : NULL, weight, 1);
    o[0] = g.margin; o[1] = g.sortedByYDiff; o[2] = g.DITHER_MAX; o[3] = g.ditherMax; o[4] = g.thresold;
    return g.beta;
}

/* java.util.PriorityQueue<ErrorBox> with comparator Double.compare(o2.yDiff, o1.yDiff) (:87-94) */
static int dcompare(double a, double b) {          /* Double.compare */
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t ab, bb; memcpy(&ab, &a, 8); memcpy(&bb, &b, 8);
    /* canonical NaN not needed; -0.0 < 0.0 */
    return ab == bb ? 0 : (ab < bb ? -1 : 1);
}
static int eb_cmp(const ErrorBox* o1, const ErrorBox* o2) { return dcompare(o2->yDiff, o1->yDiff); }
static void pq_offer(GilbertCurve* g, const ErrorBox* x) {
    int k = g->qsize++;
    while (k > 0) {
        int parent = (k - 1) >> 1;
        if (eb_cmp(x, &g->eq[parent]) >= 0) break;
        g->eq[k] = g->eq[parent];
        k = parent;
    }
    g->eq[k] = *x;
}
static void pq_poll(GilbertCurve* g) {
    int n = --g->qsize;
    ErrorBox x = g->eq[n];
    if (n > 0) {
        int k = 0, half = n >> 1;
        while (k < half) {
            int child = 2 * k + 1, right = child + 1;
            if (right < n && eb_cmp(&g->eq[child], &g->eq[right]) > 0) child = right;
            if (eb_cmp(&x, &g->eq[child]) <= 0) break;
            g->eq[k] = g->eq[child];
            k = child;
        }
        g->eq[k] = x;
    }
}
static void q_add(GilbertCurve* g, const ErrorBox* e) {
    if (g->sortedByYDiff) pq_offer(g, e);
    else { g->eq[(g->qhead + g->qsize) % QCAP] = *e; g->qsize++; }
}
static void q_poll(GilbertCurve* g) {
    if (g->sortedByYDiff) pq_poll(g);
    else { g->qhead = (g->qhead + 1) % QCAP; g->qsize--; }
}
static const ErrorBox* q_at(const GilbertCurve* g, int t) {     /* t-th element in iterator order */
    return g->sortedByYDiff ? &g->eq[t] : &g->eq[(g->qhead + t) % QCAP];
}

/* :336-354 */
static void initWeights(GilbertCurve* g, int size) {
    const float weightRatio = (float) pow((double) (343.0f + 1.0f), (double) (1.0f / (size - 1.0f)));
    float weight = 1.0f, sumweight = 0.0f;
    g->nweights = size;
    for (int c = 0; c < size; ++c) {
        ErrorBox z; memset(&z, 0, sizeof z);
        q_add(g, &z);
        sumweight += (g->weights[size - c - 1] = weight);
        weight /= weightRatio;
    }
    weight = 0.0f;
    for (int c = 0; c < size; ++c) weight += (g->weights[c] /= sumweight);
    g->weights[0] += 1.0f - weight;
}

/* :114-123 */
static float normalDistribution(float x, float peak) {
    const float mean = .5f, stdDev = .1f;
    double exponent = -sqr((double) (x - mean)) / (2 * sqr((double) stdDev));
    double pdf = (1 / (stdDev * sqrt(2 * J_PI))) * exp(exponent);
    double maxPdf = 1 / (stdDev * sqrt(2 * J_PI));
    double scaledPdf = (pdf / maxPdf) * peak;
    return (float) fmax(0.0, fmin((double) peak, scaledPdf));
}

/* :125-185 */
static int ditherPixel(GilbertCurve* g, int x, int y, int32_t c2, float beta) {
    const int bidx = x + y * g->width;
    const int32_t pixel = g->pixels[bidx];
    const int K = g->K;
    const int32_t* palette = g->palette;
    const float* sal = g->saliencies;
    int r_pix = c_red(c2), g_pix = c_green(c2), b_pix = c_blue(c2), a_pix = c_alpha(c2);
    const double weight = g->weight;
    const float strength = 1 / 3.0f;
    const int acceptedDiff = (2 > K - g->margin) ? 2 : K - g->margin;
    if (K <= 4 && sal[bidx] > .2f && sal[bidx] < .25f)
        c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], beta * 2 / sal[bidx], strength, x, y);
    else if (K <= 4 || Y_Diff(pixel, c2) < (2 * acceptedDiff)) {
        if (K > 64) {
            float kappa = sal[bidx] < .6f ? beta * .15f / sal[bidx] : beta * .4f / sal[bidx];
            c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], kappa, strength, x, y);
        }
        else if (K > 16 && weight < .005)
            c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], beta * normalDistribution(sal[bidx], .5f) + beta, strength, x, y);
        else
            c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], beta * .5f / sal[bidx], strength, x, y);
    }
    double gamma = (K <= 32 && weight < .01 && weight > .007) ? 1 - beta : beta;
    if (K > 4 && Y_Diff(pixel, c2) > (gamma * acceptedDiff)) {
        g_dbg[4]++;
        if (g->margin > 6 || gamma > beta) {
            float kappa = sal[bidx] < .4f ? beta * .4f * sal[bidx] : beta * .4f / sal[bidx];
            int32_t c1 = c_argb(a_pix, r_pix, g_pix, b_pix);
            if (K > 32 && sal[bidx] < .9)
                kappa = beta * normalDistribution(sal[bidx], 2.0f);
            else {
                if (weight >= .0015 && sal[bidx] < .6) c1 = pixel;
                if (weight >= .005 && sal[bidx] < .6)
                    kappa = beta * normalDistribution(sal[bidx], weight < .0008 ? 2.5f : 1.75f);
                else if (K >= 32 || Y_Diff(c1, c2) > (gamma * J_PI * acceptedDiff)) {
                    double ub = 1 - K / 320.0;
                    if (sal[bidx] > .15 && sal[bidx] < ub)
                        kappa = beta * (!g->sortedByYDiff && weight < .0025 ? .55f : .5f) / sal[bidx];
                    else
                        kappa = beta * normalDistribution(sal[bidx], weight < .0025 ? 1.82f : 2.0f);
                }
            }
            c2 = blue_diffuse(c1, palette[g->qPixels[bidx]], kappa, strength, x, y);
        }
        else if (K <= 32 && weight >= .004)
            c2 = blue_diffuse(c2, palette[g->qPixels[bidx]], beta * normalDistribution(sal[bidx], .25f), strength, x, y);
        else
            c2 = c_argb(a_pix, r_pix, g_pix, b_pix);
    }
    if (g->DITHER_MAX < 16 && K > 4 && sal[bidx] < .6f && Y_Diff(pixel, c2) > g->margin - 1)
        c2 = c_argb(a_pix, r_pix, g_pix, b_pix);
    if (K > 32 && sal[bidx] > .95) {
        g_dbg[5]++;
        float kappa = beta * fmaxf(.05f, .75f - K / 128.0f) * sal[bidx];
        c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], kappa, strength, x, y);
    }
    {   /* diagnostics only: does the colour handed to the lookup lie in the 5-6-5 cell of the undithered colour? */
        const int32_t c0 = c_argb(a_pix, r_pix, g_pix, b_pix);
        if (((c0 ^ c2) & 0x00F8FCF8) == 0) g_dbg[11]++;
    }
    return ditherable_nearest(g->dth, palette, K, c2, bidx);
}

/* :187-280 */
static void diffusePixel(GilbertCurve* g, int x, int y) {
    const int bidx = x + y * g->width;
    const int32_t pixel = g->pixels[bidx];
    const int K = g->K;
    const int32_t* palette = g->palette;
    const float* sal = g->saliencies;
    ErrorBox error;
    error.yDiff = 0;
    error.p[0] = (float) c_red(pixel); error.p[1] = (float) c_green(pixel);
    error.p[2] = (float) c_blue(pixel); error.p[3] = (float) c_alpha(pixel);

    float maxErr = (float) (g->DITHER_MAX - 1);
    int i = g->sortedByYDiff ? g->nweights - 1 : 0;
    for (int t = 0; t < g->qsize; ++t) {
        if (i < 0 || i >= g->nweights) break;
        const ErrorBox* eb = q_at(g, t);
        for (int j = 0; j < 4; ++j) {
            error.p[j] += eb->p[j] * g->weights[i];
            if (error.p[j] > maxErr) maxErr = error.p[j];
        }
        i += g->sortedByYDiff ? -1 : 1;
    }

    int r_pix = j_d2i(fmin(255.0, fmax((double) error.p[0], 0.0)));
    int g_pix = j_d2i(fmin(255.0, fmax((double) error.p[1], 0.0)));
    int b_pix = j_d2i(fmin(255.0, fmax((double) error.p[2], 0.0)));
    int a_pix = j_d2i(fmin(255.0, fmax((double) error.p[3], 0.0)));

    int32_t c2 = c_argb(a_pix, r_pix, g_pix, b_pix);
    if (sal != NULL && g->dither && !g->sortedByYDiff && (!g->hasAlpha || c_alpha(pixel) < a_pix)) {
        if ((K >= 256 && sal[bidx] > .99f) || (g->hasAlpha && (c_alpha(pixel) - a_pix) < (.5 * g->margin))) {
            g_dbg[6]++;
            g->qPixels[bidx] = ditherable_nearest(g->dth, palette, K, c2, bidx);
        }
        else
            g->qPixels[bidx] = ditherPixel(g, x, y, c2, g->beta);
    }
    else if (K <= 32 && a_pix > 0xF0) {
        g->qPixels[bidx] = ditherable_nearest(g->dth, palette, K, c2, bidx);
        const int acceptedDiff = (2 > K - g->margin) ? 2 : K - g->margin;
        if (sal != NULL && (Y_Diff(pixel, c2) > acceptedDiff || U_Diff(pixel, c2) > (2 * acceptedDiff))) {
            const float strength = 1 / 3.0f;
            c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], 1 / sal[bidx], strength, x, y);
            g->qPixels[bidx] = ditherable_nearest(g->dth, palette, K, c2, bidx);
        }
    }
    else
        g->qPixels[bidx] = ditherable_nearest(g->dth, palette, K, c2, bidx);

    if (g->qsize >= g->DITHER_MAX) q_poll(g);
    else if (g->qsize != 0) initWeights(g, g->qsize);

    c2 = palette[g->qPixels[bidx]];
    error.p[0] = (float) (r_pix - c_red(c2));
    error.p[1] = (float) (g_pix - c_green(c2));
    error.p[2] = (float) (b_pix - c_blue(c2));
    error.p[3] = (float) (a_pix - c_alpha(c2));

    int denoise = K > 2;
    int diffuse = TELL_BLUE_NOISE[bidx & 4095] > g->thresold;
    error.yDiff = g->sortedByYDiff ? Y_Diff(pixel, c2) : 1;
    int illusion = !diffuse && TELL_BLUE_NOISE[j_d2i(error.yDiff * 4096) & 4095] > g->thresold;

    int unaccepted = 0;
    int errLength = denoise ? 3 : 0;
    g_dbg[0]++;
    { int any = 0; for (int j = 0; j < errLength; ++j) if (fabsf(error.p[j]) >= g->ditherMax) any = 1; if (any) { g_dbg[3]++; if (diffuse) g_dbg[9]++; } if (maxErr > (float) (g->DITHER_MAX - 1)) g_dbg[10]++; }
    for (int j = 0; j < errLength; ++j) {
        if (fabsf(error.p[j]) >= g->ditherMax) {
            if (g->sortedByYDiff && sal != NULL) unaccepted = 1;
            if (diffuse)
                error.p[j] = (float) tanh((double) (error.p[j] / maxErr * 20)) * (g->ditherMax - 1);
            else if (illusion)
                error.p[j] = (float) ((double) (error.p[j] / maxErr) * error.yDiff) * (g->ditherMax - 1);
            else
                error.p[j] /= (float) (1 + sqrt((double) g->ditherMax));
        }
        if (g->sortedByYDiff && sal == NULL && fabsf(error.p[j]) >= g->DITHER_MAX) unaccepted = 1;
    }

    if (unaccepted) {
        if (sal != NULL) g->qPixels[bidx] = ditherPixel(g, x, y, c2, g->beta);
        else if (Y_Diff(pixel, c2) > 3 && U_Diff(pixel, c2) > 3) {
            const float strength = 1 / 3.0f;
            c2 = blue_diffuse(pixel, palette[g->qPixels[bidx]], strength, strength, x, y);
            g->qPixels[bidx] = ditherable_nearest(g->dth, palette, K, c2, bidx);
        }
    }
    q_add(g, &error);
    /* :278-279 deferred: the caller keeps the index and maps to ARGB afterwards (identical result: each
       qPixels[bidx] is written only while its own pixel is visited). */
}

/* :282-334 */
typedef struct { int32_t* xy; int64_t n; } pathbuf;
static inline int isignum(int v) { return (v > 0) - (v < 0); }
static void generate2d(pathbuf* pb, int x, int y, int ax, int ay, int bx, int by) {
    int w = abs(ax + ay), h = abs(bx + by);
    int dax = isignum(ax), day = isignum(ay), dbx = isignum(bx), dby = isignum(by);
    if (h == 1) {
        for (int i = 0; i < w; ++i) { pb->xy[2 * pb->n] = x; pb->xy[2 * pb->n + 1] = y; pb->n++; x += dax; y += day; }
        return;
    }
    if (w == 1) {
        for (int i = 0; i < h; ++i) { pb->xy[2 * pb->n] = x; pb->xy[2 * pb->n + 1] = y; pb->n++; x += dbx; y += dby; }
        return;
    }
    int ax2 = ax / 2, ay2 = ay / 2, bx2 = bx / 2, by2 = by / 2;
    int w2 = abs(ax2 + ay2), h2 = abs(bx2 + by2);
    if (2 * w > 3 * h) {
        if ((w2 % 2) != 0 && w > 2) { ax2 += dax; ay2 += day; }
        generate2d(pb, x, y, ax2, ay2, bx, by);
        generate2d(pb, x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by);
        return;
    }
    if ((h2 % 2) != 0 && h > 2) { bx2 += dbx; by2 += dby; }
    generate2d(pb, x, y, bx2, by2, ax2, ay2);
    generate2d(pb, x + bx2, y + by2, ax, ay, bx - bx2, by - by2);
    generate2d(pb, x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), -bx2, -by2, -(ax - ax2), -(ay - ay2));
}
/* run() :356-365 curve part */
int64_t nqo_gilbert_path(int width, int height, int32_t* out_xy) {
    pathbuf pb = {out_xy, 0};
    if (width <= 0 || height <= 0) return 0;
    if (width >= height) generate2d(&pb, 0, 0, width, 0, 0, height);
    else generate2d(&pb, 0, 0, 0, height, width, 0);
    return pb.n;
}

/* GilbertCurve.dither :367-373 over one rectangle [x0,x0+tw) x [y0,y0+th) of the image (whole image in the
 * reference).  qPixels receives palette indices. */
/* tile_chain (the tiled restatement only; 0 = the reference's single chain): a chain of the sorted-by-yDiff mode starts with its queue
 * in the STEADY STATE -- the growth 1 -> 3 -> 7 -> 15 (-> 31) of :231-234 run with zero ErrorBoxes in place of the first pixels' errors --
 * instead of empty.  The reference grows the queue once per image; a decomposition that restarted the growth at every tile would put
 * the erratic first steps of it (queue sizes 1 and 3: one or three weights) at the start of every tile: measured with this oracle on
 * 2000 random colours / 192^2: 56 pixels with deltaE > 40 against the source with 4x4 tiles, 14 with 8x8, 0 in the sequential chain,
 * and 0 at every tile size with the steady-state start (the mean error also drops slightly below the 64x64-tile figure).  The
 * other modes already start that way: run() pre-fills their ArrayDeque with DITHER_MAX zero boxes (:358-359). */
static void gilbert_run(nqo_quantizer* q, ditherable* dth, const int32_t* palette, int K, const float* saliencies,
                        double weight, int dither, int32_t* qIndex, int x0, int y0, int tw, int th, int tile_chain) {
    GilbertCurve g;
    gc_init(&g, q->width, q->height, q->pixels, palette, K, qIndex, dth, saliencies, weight, dither);
    if (!g.sortedByYDiff) initWeights(&g, g.DITHER_MAX);
    else if (tile_chain) {
        ErrorBox z; memset(&z, 0, sizeof z);
        q_add(&g, &z);
        while (g.qsize < g.DITHER_MAX) { const int size = g.qsize; initWeights(&g, size); q_add(&g, &z); }
    }
    int32_t* xy = malloc((size_t) 2 * tw * th * sizeof(int32_t) + 8);
    int64_t n = nqo_gilbert_path(tw, th, xy);
    for (int64_t s = 0; s < n; ++s) diffusePixel(&g, x0 + xy[2 * s], y0 + xy[2 * s + 1]);
    free(xy);
}

/* NQ/BlueNoise.java:207-222 (qIndex in: indices from the gilbert pass; out: final index + ARGB) */
static void bluenoise_dither(nqo_quantizer* q, ditherable* dth, const int32_t* palette, int K, int32_t* qIndex,
                             int32_t* out_argb, float weight, int per_pixel_rng, int y_lo, int y_hi) {
    const float strength = 1 / 3.0f;
    int64_t px_rng;
    int64_t* save_rng = dth->rng;
    for (int y = y_lo; y < y_hi; ++y) {
        for (int x = 0; x < q->width; ++x) {
            const int bidx = x + y * q->width;
            int32_t pixel = q->pixels[bidx];
            int32_t qPixel = palette[qIndex[bidx]];
            int32_t c1 = blue_diffuse(pixel, qPixel, weight, strength, x, y);
            if (per_pixel_rng) {   /* tiled restatement: one Random(mix64((seed ^ tag) + bidx)) per pixel */
                nqo_jrandom_seed(&px_rng, (int64_t) mix64(((uint64_t) q->seed ^ 0xB10E5EEDULL) + (uint64_t) bidx));
                dth->rng = &px_rng;
            }
            int k = ditherable_nearest(dth, palette, K, c1, bidx);
            qIndex[bidx] = k;
            out_argb[bidx] = palette[k];
        }
    }
    dth->rng = save_rng;
}

/* row_first / row_count (tiled mode only; row_count < 0 = all): restrict the pass to that range of TILE ROWS -- tiles are independent
 * chains with their own Random(mix64(seed + tileIndex)), so any subset equals the same tiles of the whole-image run; pixels outside
 * the range come back as index 0.  Lets the tests check a full-size image (BASELINE cfg 3-5) on a sample of its tile rows. */
static int dither_impl(nqo_quantizer* q, const int32_t* palette, int K, int dither, int tile_w, int tile_h,
                       int row_first, int row_count, int32_t* out_argb, int32_t* out_index) {
    const size_t N = (size_t) q->width * q->height;
    const int tiled = tile_w > 0 && tile_h > 0;
    int y_lo = 0, y_hi = q->height;
    if (tiled && row_count >= 0) {
        y_lo = row_first * tile_h; y_hi = (row_first + row_count) * tile_h;
        if (y_lo > q->height) y_lo = q->height;
        if (y_hi > q->height) y_hi = q->height;
    }
    ditherable dth = {q, dither, &q->rng};
    double t0 = now_s();
    const double weight_in = q->weight;
    if (q->hasSemiTransparency) q->weight *= -1;          /* RGB :396-397, LAB :496-497 */
    if (q->kind == 1) {
        /* LAB :499-508 */
        if (dither && q->saliencies == NULL && (K <= 256 || q->weight > .99)) {
            q->saliencies = calloc(N ? N : 1, sizeof(float));
            float saliencyBase = .1f;
            for (size_t i = (size_t) y_lo * q->width; i < (size_t) y_hi * q->width; ++i) {
                Lab lab1 = getLab(q, q->pixels[i]);
                q->saliencies[i] = saliencyBase + (1 - saliencyBase) * lab1.L / 100.0f * lab1.alpha / 255.0f;
            }
            if (tiled && row_count >= 0) q->saliencies_partial = 1;
        }
    }
    const float* sal = q->kind == 1 ? q->saliencies : NULL;
    int32_t* qIndex = calloc(N ? N : 1, sizeof(int32_t));
    if (!tiled) {
        gilbert_run(q, &dth, palette, K, sal, q->weight, dither, qIndex, 0, 0, q->width, q->height, 0);
    } else {
        q->no_cache = 1;
        int64_t tile_rng;
        int tix = 0;
        for (int ty = 0; ty < q->height; ty += tile_h)
            for (int tx = 0; tx < q->width; tx += tile_w, ++tix) {
                if (ty < y_lo || ty >= y_hi) continue;
                int tw = q->width - tx < tile_w ? q->width - tx : tile_w;
                int th = q->height - ty < tile_h ? q->height - ty : tile_h;
                /* per-tile stream: java.util.Random(mix64(seed + tileIndex)) */
                nqo_jrandom_seed(&tile_rng, (int64_t) mix64((uint64_t) q->seed + (uint64_t) tix));
                dth.rng = &tile_rng;
                gilbert_run(q, &dth, palette, K, sal, q->weight, dither, qIndex, tx, ty, tw, th, 1);
            }
        dth.rng = &q->rng;
    }
    q->t_stage[4] += now_s() - t0; t0 = now_s();
    /* GilbertCurve.java:278-279 + BlueNoise post-pass RGB :400-401, LAB :511-515 */
    if (!dither && K > 32) {
        float bw = 1.0f;
        if (q->kind == 1) {
            int64_t sz = (tiled && q->frozen_distinct >= 0) ? q->frozen_distinct : (int64_t) q->pixelMap.n;
            double delta = sqr(K) / (double) sz;
            bw = delta > 0.023 ? 1.0f : (float) (37.013 * delta + 0.906);
        }
        for (size_t i = 0; i < N; ++i) out_argb[i] = palette[qIndex[i]];
        bluenoise_dither(q, &dth, palette, K, qIndex, out_argb, bw, tiled, y_lo, y_hi);
    } else {
        for (size_t i = 0; i < N; ++i) out_argb[i] = palette[qIndex[i]];
    }
    if (out_index) memcpy(out_index, qIndex, N * sizeof(int32_t));
    free(qIndex);
    q->no_cache = 0;
    imap_clear(&q->closestMap); imap_clear(&q->nearestMap);
    if (q->kind == 1) imap_clear(&q->pixelMap);
    if (q->saliencies_partial) { free(q->saliencies); q->saliencies = NULL; q->saliencies_partial = 0; }
    if (tiled) q->weight = weight_in;      /* the tiled restatement may be called repeatedly on one object (the GPU handle keeps its weight too) */
    q->t_stage[5] += now_s() - t0;
    return 0;
}

int nqo_dither(nqo_quantizer* q, const int32_t* palette, int K, int dither, int32_t* out_argb, int32_t* out_index) {
    return dither_impl(q, palette, K, dither, 0, 0, 0, -1, out_argb, out_index);
}
int nqo_dither_tiled(nqo_quantizer* q, const int32_t* palette, int K, int dither, int tile_w, int tile_h,
                     int32_t* out_argb, int32_t* out_index) {
    q->frozen_distinct = q->distinct_after_hist;
    return dither_impl(q, palette, K, dither, tile_w, tile_h, 0, -1, out_argb, out_index);
}
int nqo_dither_tile_rows(nqo_quantizer* q, const int32_t* palette, int K, int dither, int tile_w, int tile_h, int row_first, int row_count,
                         int32_t* out_argb, int32_t* out_index) {
    q->frozen_distinct = q->distinct_after_hist;
    return dither_impl(q, palette, K, dither, tile_w, tile_h, row_first, row_count, out_argb, out_index);
}

/* The ditherers' static entry points with caller-supplied saliencies / weight (NQ/GilbertCurve.java:367-373, NQ/BlueNoise.java:207-222);
 * the quantizer object is the Ditherable.  tile_w/tile_h <= 0: one curve over the image with the object's caches and Random (they are
 * NOT cleared here, so a following nqo_bluenoise_dither_stage continues them as inside dither()); > 0: the tiled restatement. */
int nqo_gilbert_dither_stage(nqo_quantizer* q, const int32_t* palette, int K, const float* saliencies, double weight, int dither,
                             int tile_w, int tile_h, int32_t* out_qpixels, int32_t* out_index) {
    const size_t N = (size_t) q->width * q->height;
    const int tiled = tile_w > 0 && tile_h > 0;
    ditherable dth = {q, dither, &q->rng};
    int32_t* qIndex = calloc(N ? N : 1, sizeof(int32_t));
    if (!tiled) {
        imap_clear(&q->closestMap); imap_clear(&q->nearestMap);
        gilbert_run(q, &dth, palette, K, saliencies, weight, dither, qIndex, 0, 0, q->width, q->height, 0);
    } else {
        q->no_cache = 1;
        int64_t tile_rng;
        int tix = 0;
        for (int ty = 0; ty < q->height; ty += tile_h)
            for (int tx = 0; tx < q->width; tx += tile_w, ++tix) {
                int tw = q->width - tx < tile_w ? q->width - tx : tile_w;
                int th = q->height - ty < tile_h ? q->height - ty : tile_h;
                nqo_jrandom_seed(&tile_rng, (int64_t) mix64((uint64_t) q->seed + (uint64_t) tix));
                dth.rng = &tile_rng;
                gilbert_run(q, &dth, palette, K, saliencies, weight, dither, qIndex, tx, ty, tw, th, 1);
            }
        q->no_cache = 0;
    }
    for (size_t i = 0; i < N; ++i) out_qpixels[i] = (dither || K <= 32) ? palette[qIndex[i]] : qIndex[i];     /* :278-279 */
    if (out_index) memcpy(out_index, qIndex, N * sizeof(int32_t));
    free(qIndex);
    return 0;
}
int nqo_bluenoise_dither_stage(nqo_quantizer* q, const int32_t* palette, int K, int32_t* io_qpixels, float weight, int tiled,
                               int32_t* out_index) {
    const size_t N = (size_t) q->width * q->height;
    ditherable dth = {q, 0, &q->rng};
    int32_t* qIndex = malloc((N ? N : 1) * sizeof(int32_t));
    memcpy(qIndex, io_qpixels, N * sizeof(int32_t));
    if (tiled) q->no_cache = 1;
    bluenoise_dither(q, &dth, palette, K, qIndex, io_qpixels, weight, tiled, 0, q->height);
    q->no_cache = 0;
    if (out_index) memcpy(out_index, qIndex, N * sizeof(int32_t));
    free(qIndex);
    imap_clear(&q->closestMap); imap_clear(&q->nearestMap);
    return 0;
}

/* NQ/PnnQuantizer.java:410-436 */
void nqo_prescan(nqo_quantizer* q, int nMaxColors) {
    double t0 = now_s();
    const size_t N = (size_t) q->width * q->height;
    int semiTransCount = 0;
    for (size_t i = 0; i < N; ++i) {
        int32_t pixel = q->pixels[i];
        int alfa = (pixel >> 24) & 0xff, r = (pixel >> 16) & 0xff, g = (pixel >> 8) & 0xff, b = pixel & 0xff;
        q->pixels[i] = c_argb(alfa, r, g, b);
        if (alfa < 0xE0) {
            if (alfa == 0) {
                q->m_transparentPixelIndex = (int) i;
                if (nMaxColors > 2) q->m_transparentColor = q->pixels[i];
                else q->pixels[i] = q->m_transparentColor;
            }
            else if (alfa > q->alphaThreshold) ++semiTransCount;
        }
    }
    q->hasSemiTransparency = semiTransCount > 0;
    if (nMaxColors <= 32) q->PR = q->PG = q->PB = q->PA = 1;
    else { q->PR = coeffs[0][0]; q->PG = coeffs[0][1]; q->PB = coeffs[0][2]; }
    q->nMaxColors = nMaxColors;
    q->t_stage[0] += now_s() - t0;
}

/* NQ/PnnQuantizer.java:409-456 */
int nqo_convert(nqo_quantizer* q, int nMaxColors, int dither,
                int32_t* out_argb, int32_t* out_index, int32_t* out_palette, int32_t* out_K) {
    nqo_prescan(q, nMaxColors);
    int K;
    if (nMaxColors > 2) {
        K = nqo_pnnquan(q, nMaxColors, out_palette);
        if (K < 0) return -1;
    } else {
        K = nMaxColors;
        q->weight = 1;
        if (q->m_transparentPixelIndex >= 0) { out_palette[0] = q->m_transparentColor; out_palette[1] = COLOR_BLACK; }
        else { out_palette[0] = COLOR_BLACK; out_palette[1] = COLOR_WHITE; }
        q->paletteLength = K;
    }
    *out_K = K;
    return nqo_dither(q, out_palette, K, dither, out_argb, out_index);
}
