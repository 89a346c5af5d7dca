// nq_device.h -- device-side building blocks shared by every gfx950 kernel of libnquant_hip.so.
//
// Everything here mirrors the arithmetic of the reference (NQ/ = nQuant.master/src/main/java/com/android/nQuant/)
// type for type: Java float stays f32, Java double stays f64, no fused multiply-add (the whole library is
// compiled with -ffp-contract=off), Java (int) narrowing made explicit.  Transcendentals come from the ROCm
// device library (OCML); tables that only depend on an 8-bit channel are computed once on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nq_kernels.h"

namespace nq {

// ------------------------------------------------------------------------------------------------
// constants shared by all kernels of this translation unit
// ------------------------------------------------------------------------------------------------
struct ConstTables {
    double gamma[256];     // CIELABConvertor.gammaToLinear(ch) == ColorUtils sRGB companding (NQ/CIELABConvertor.java:71-75)
    double exp1_5;         // Math.exp(1.5)   (NQ/PnnLABQuantizer.java:348)
    double exp1_75;        // Math.exp(1.75)  (NQ/PnnLABQuantizer.java:62)
    int8_t blue[4096];     // TELL_BLUE_NOISE (NQ/BlueNoise.java:13-178), data
};
static __constant__ ConstTables g_tab;   // one copy per device translation unit (no relocatable device code): each TU has its upload_tables*()

// struct DevParams: nq_kernels.h

#define NQ_PI 3.141592653589793
#define NQ_E 2.718281828459045

// ---- Java numeric narrowing -------------------------------------------------------------------
__device__ __forceinline__ int j_d2i(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return 2147483647;
    if (d <= -2147483648.0) return (int) 0x80000000;
    return (int) d;
}
__device__ __forceinline__ long long j_round(double a) {      // Math.round(double)
    if (a != a) return 0;
    if (fabs(a) >= 4503599627370496.0) return (long long) a;
    double f = floor(a);
    return (long long) f + ((a - f) >= 0.5 ? 1 : 0);
}
__device__ __forceinline__ double sqr(double v) { return v * v; }

// ---- android.graphics.Color ---------------------------------------------------------------------
__device__ __forceinline__ int c_alpha(int c) { return (int) (((unsigned) c) >> 24); }
__device__ __forceinline__ int c_red(int c) { return (c >> 16) & 0xFF; }
__device__ __forceinline__ int c_green(int c) { return (c >> 8) & 0xFF; }
__device__ __forceinline__ int c_blue(int c) { return c & 0xFF; }
__device__ __forceinline__ int c_argb(int a, int r, int g, int b) {
    return (int) (((unsigned) a << 24) | ((unsigned) r << 16) | ((unsigned) g << 8) | (unsigned) b);
}

// NQ/BitmapUtilities.java:8-15
__device__ __forceinline__ int getColorIndex(int c, bool hasSemiTransparency, bool hasTransparency) {
    if (hasSemiTransparency)
        return (c_alpha(c) & 0xF0) << 8 | (c_red(c) & 0xF0) << 4 | (c_green(c) & 0xF0) | (c_blue(c) >> 4);
    if (hasTransparency)
        return (c_alpha(c) & 0x80) << 8 | (c_red(c) & 0xF8) << 7 | (c_green(c) & 0xF8) << 2 | (c_blue(c) >> 3);
    return (c_red(c) & 0xF8) << 8 | (c_green(c) & 0xFC) << 3 | (c_blue(c) >> 3);
}

// ---- CIELABConvertor ----------------------------------------------------------------------------
struct Lab { float alpha, A, B, L; };

// RGB2LAB: NQ/CIELABConvertor.java:58-69 over androidx ColorUtils.colorToLAB (published algorithm)
__device__ __forceinline__ double pivot_xyz(double c) { return c > 0.008856 ? pow(c, 1 / 3.0) : (903.3 * c + 16) / 116; }
__device__ __forceinline__ void rgb_to_xyz(int c1, double& X, double& Y, double& Z) {
    double sr = g_tab.gamma[c_red(c1)], sg = g_tab.gamma[c_green(c1)], sb = g_tab.gamma[c_blue(c1)];
    X = 100 * (sr * 0.4124 + sg * 0.3576 + sb * 0.1805);
    Y = 100 * (sr * 0.2126 + sg * 0.7152 + sb * 0.0722);
    Z = 100 * (sr * 0.0193 + sg * 0.1192 + sb * 0.9505);
}
__device__ __forceinline__ Lab RGB2LAB(int c1) {
    double X, Y, Z;
    rgb_to_xyz(c1, X, Y, Z);
    double x = pivot_xyz(X / 95.047), y = pivot_xyz(Y / 100.0), z = pivot_xyz(Z / 108.883);
    Lab lab;
    lab.alpha = (float) c_alpha(c1);
    lab.L = (float) fmax(0.0, 116 * y - 16);
    lab.A = (float) (500 * (x - y));
    lab.B = (float) (200 * (y - z));
    return lab;
}
// Cube root for the PER-PIXEL Lab conversions (histogram, nearest fallback, saliency map): hardware exp2/log2 seed for x^(-1/3)
// (3e-7 relative), ONE division-free Newton step (-> 2e-13), one residual correction on y = x r^2 with an explicit fma (-> rounding
// level).  Measured against exact arithmetic on 20 000 points of [0.008856, 1.2]: <= 0.72 ulp, the same as with two Newton steps
// (libm's pow(x, 1/3.0) itself is 1.2 ulp from the true cube root because 1/3.0 is not 1/3).  Every consumer narrows the Lab value
// to float, so the two agree except with probability ~1e-9 per evaluation (tests: the whole 2^24 colour cube, zero flips).
__device__ __forceinline__ double cbrt_fast(double x) {
    const float r0 = __builtin_amdgcn_exp2f(-0.333333343f * __builtin_amdgcn_logf((float) x));
    double r = (double) r0;
    const double third = 1.0 / 3.0;
    const double r3 = r * r * r; r = r + r * (1.0 - x * r3) * third;
    double y = x * r * r;
    const double res = fma(y * y, y, -x);
    return y - res * (r * r) * third;
}
__device__ __forceinline__ double pivot_xyz_fast(double c) { return c > 0.008856 ? cbrt_fast(c) : (903.3 * c + 16) / 116; }
// RGB2LAB with the gamma table taken from `gamma` (LDS copy) and the fast cube root
__device__ __forceinline__ Lab RGB2LAB_fast(int c1, const double* __restrict__ gamma) {
    const double sr = gamma[c_red(c1)], sg = gamma[c_green(c1)], sb = gamma[c_blue(c1)];
    const double X = 100 * (sr * 0.4124 + sg * 0.3576 + sb * 0.1805);
    const double Y = 100 * (sr * 0.2126 + sg * 0.7152 + sb * 0.0722);
    const double Z = 100 * (sr * 0.0193 + sg * 0.1192 + sb * 0.9505);
    const double x = pivot_xyz_fast(X / 95.047), y = pivot_xyz_fast(Y / 100.0), z = pivot_xyz_fast(Z / 108.883);
    Lab lab;
    lab.alpha = (float) c_alpha(c1);
    lab.L = (float) fmax(0.0, 116 * y - 16);
    lab.A = (float) (500 * (x - y));
    lab.B = (float) (200 * (y - z));
    return lab;
}
// L channel only (the saliency map needs nothing else): same arithmetic as RGB2LAB().L
__device__ __forceinline__ float RGB2L(int c1) {
    double sr = g_tab.gamma[c_red(c1)], sg = g_tab.gamma[c_green(c1)], sb = g_tab.gamma[c_blue(c1)];
    double Y = 100 * (sr * 0.2126 + sg * 0.7152 + sb * 0.0722);
    double y = pivot_xyz_fast(Y / 100.0);
    return (float) fmax(0.0, 116 * y - 16);
}
// saliency of one pixel: NQ/PnnLABQuantizer.java:156 / :506
__device__ __forceinline__ float saliency_of(int c) {
    const float saliencyBase = .1f;
    float L = RGB2L(c), alpha = (float) c_alpha(c);
    return saliencyBase + (1 - saliencyBase) * L / 100.0f * alpha / 255.0f;
}

// LAB2RGB: NQ/CIELABConvertor.java:77-80 over ColorUtils.LABToColor; *ok=false where setAlphaComponent throws
__device__ __forceinline__ int LAB2RGB(Lab lab, bool* ok) {
    double l = lab.L, a = lab.A, b = lab.B;
    double fy = (l + 16) / 116, fx = a / 500 + fy, fz = fy - b / 200;
    double tmp = pow(fx, 3.0);
    double xr = tmp > 0.008856 ? tmp : (116 * fx - 16) / 903.3;
    double yr = l > 903.3 * 0.008856 ? pow(fy, 3.0) : l / 903.3;
    tmp = pow(fz, 3.0);
    double zr = tmp > 0.008856 ? tmp : (116 * fz - 16) / 903.3;
    double x = xr * 95.047, y = yr * 100.0, z = zr * 108.883;
    double r = (x * 3.2406 + y * -1.5372 + z * -0.4986) / 100;
    double g = (x * -0.9689 + y * 1.8758 + z * 0.0415) / 100;
    double bb = (x * 0.0557 + y * -0.2040 + z * 1.0570) / 100;
    r = r > 0.0031308 ? 1.055 * pow(r, 1 / 2.4) - 0.055 : 12.92 * r;
    g = g > 0.0031308 ? 1.055 * pow(g, 1 / 2.4) - 0.055 : 12.92 * g;
    bb = bb > 0.0031308 ? 1.055 * pow(bb, 1 / 2.4) - 0.055 : 12.92 * bb;
    long long ri = j_round(r * 255), gi = j_round(g * 255), bi = j_round(bb * 255);
    int R = ri < 0 ? 0 : ri > 255 ? 255 : (int) ri;
    int G = gi < 0 ? 0 : gi > 255 ? 255 : (int) gi;
    int B = bi < 0 ? 0 : bi > 255 ? 255 : (int) bi;
    int alpha = j_d2i((double) lab.alpha);
    *ok = !(alpha < 0 || alpha > 255);
    return (int) ((((unsigned) alpha) << 24) | ((unsigned) R << 16) | ((unsigned) G << 8) | (unsigned) B);
}

// :86-89
__device__ __forceinline__ float deg2Rad(double deg) { return (float) (deg * (NQ_PI / 180.0)); }

// :91-98
__device__ __forceinline__ float L_prime_div_k_L_S_L(const Lab& lab1, const Lab& lab2) {
    const float k_L = 1.0f;
    float deltaLPrime = lab2.L - lab1.L;
    float barLPrime = (lab1.L + lab2.L) / 2.0f;
    double p = sqr((double) (barLPrime - 50.0f));
    float S_L = (float) (1 + (((double) 0.015f * p) / sqrt(20 + p)));
    return deltaLPrime / (k_L * S_L);
}
// :100-118
__device__ __forceinline__ float C_prime_div_k_L_S_L(const Lab& lab1, const Lab& lab2, double& a1Prime, double& a2Prime,
                                                     double& CPrime1, double& CPrime2) {
    const float k_C = 1.0f;
    const float pow25To7 = 6103515625.0f;
    float C1 = (float) sqrt((double) ((lab1.A * lab1.A) + (lab1.B * lab1.B)));
    float C2 = (float) sqrt((double) ((lab2.A * lab2.A) + (lab2.B * lab2.B)));
    float barC = (C1 + C2) / 2.0f;
    double barC7 = pow((double) barC, 7.0);
    float G = (float) ((double) 0.5f * (1 - sqrt(barC7 / (barC7 + (double) pow25To7))));
    a1Prime = (1.0 + G) * lab1.A;
    a2Prime = (1.0 + G) * lab2.A;
    CPrime1 = sqrt((a1Prime * a1Prime) + (double) (lab1.B * lab1.B));
    CPrime2 = sqrt((a2Prime * a2Prime) + (double) (lab2.B * lab2.B));
    float deltaCPrime = (float) CPrime2 - (float) CPrime1;
    float barCPrime = ((float) CPrime1 + (float) CPrime2) / 2.0f;
    float S_C = 1 + (0.045f * barCPrime);
    return deltaCPrime / (k_C * S_C);
}
// :120-185
__device__ __forceinline__ float H_prime_div_k_L_S_L(const Lab& lab1, const Lab& lab2, double a1Prime, double a2Prime,
                                                     double CPrime1, double CPrime2, double& barCPrime, double& barhPrime) {
    const float k_H = 1.0f;
    const float deg360InRad = deg2Rad(360.0);
    const float deg180InRad = deg2Rad(180.0);
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
    if ((CPrime1 * CPrime2) == 0.0) barhPrime = hPrimeSum;
    else {
        if (fabs(hPrime1 - hPrime2) <= deg180InRad) barhPrime = hPrimeSum / 2.0;
        else {
            if (hPrimeSum < deg360InRad) barhPrime = (hPrimeSum + deg360InRad) / 2.0;
            else barhPrime = (hPrimeSum - deg360InRad) / 2.0;
        }
    }
    barCPrime = (CPrime1 + CPrime2) / 2.0;
    double bh = barhPrime;
    double T = 1.0 - (0.17 * cos(bh - deg2Rad(30.0))) + (0.24 * cos(2.0 * bh)) +
               (0.32 * cos((3.0 * bh) + deg2Rad(6.0))) - (0.20 * cos((4.0 * bh) - deg2Rad(63.0)));
    double S_H = 1 + ((double) 0.015f * barCPrime * T);
    return (float) (deltaHPrime / (k_H * S_H));
}
// :187-194
__device__ __forceinline__ float R_T(double barCPrime, double barhPrime, float C_prime_div, float H_prime_div) {
    const double pow25To7 = 6103515625.0;
    double deltaTheta = deg2Rad(30.0) * exp(-sqr((barhPrime - deg2Rad(275.0)) / deg2Rad(25.0)));
    double bc7 = pow(barCPrime, 7.0);
    double R_C = 2.0 * sqrt(bc7 / (bc7 + pow25To7));
    double rt = (-sin(2.0 * deltaTheta)) * R_C;
    return (float) (rt * C_prime_div * H_prime_div);
}

// ------------------------------------------------------------------------------------------------
// The four CIEDE2000 terms of one pair in ONE branch-free pass (the exact phase of find_nn is a single serial chain of these).
// The literal functions above call the device library's pow / atan2 / sin / cos / exp: ~1500 instructions with internal branches.
// Every result of theirs that matters is a FLOAT (deltaL', deltaC', deltaH', R_T are narrowed before they are used, :97,:117,:184,
// :193) or a comparison of an angle with a float constant.  Here the same statement sequence is evaluated with bounded-argument
// f64 kernels (Taylor / Cody-Waite, errors of a few ulp, all arguments are bounded: angles <= 9 pi, exponent <= 0) and each
// narrowing is accepted only when the value lies farther from a float rounding boundary (resp. the angle farther from the
// constant) than the accumulated error bound -- then EVERY implementation with ulp-level errors, the device library's and glibc's
// included, narrows to the same float.  Otherwise ok = false and the caller runs the literal functions.
// (selftest: nq_selftest_ciede, tests/test_gpu_boundary.py -- fast-or-fallback == literal, bit for bit, on millions of pairs.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double nq_pow7(double x) { const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2; return x4 * x3; }   // <= 2.5 ulp
// e^a for a <= 0 (|error| < 1e-15 relative): Cody-Waite + degree-12 Taylor on |f| <= ln2 / 2
__device__ __forceinline__ double nq_exp_neg(double a) {
    const double n = rint(a * 1.4426950408889634);
    double f = fma(n, -0.6931471803691238, a);
    f = fma(n, -1.9082149292705877e-10, f);
    double p = 2.08767569878681e-09;
    p = fma(p, f, 2.505210838544172e-08); p = fma(p, f, 2.755731922398589e-07); p = fma(p, f, 2.7557319223985893e-06);
    p = fma(p, f, 2.48015873015873e-05); p = fma(p, f, 0.0001984126984126984); p = fma(p, f, 0.001388888888888889);
    p = fma(p, f, 0.008333333333333333); p = fma(p, f, 0.041666666666666664); p = fma(p, f, 0.16666666666666666);
    p = fma(p, f, 0.5); p = fma(p, f, 1.0); p = fma(p, f, 1.0);
    return ldexp(p, (int) fmax(n, -1080.0));
}
// sin and cos of |x| <= 64 (|error| < 4e-16 absolute): x = k pi/2 + r (fdlibm's two-part pi/2), Taylor to r^17 / r^18 on |r| <= pi/4
__device__ __forceinline__ void nq_sincos(double x, double& sn, double& cs) {
    const double k = rint(x * 0.6366197723675814);
    double r = fma(k, -1.5707963267341256, x);
    r = fma(k, -6.077100506506192e-11, r);
    const double r2 = r * r;
    double ps = 2.8114572543455206e-15;                 // 1/17!
    ps = fma(ps, r2, -7.647163731819816e-13); ps = fma(ps, r2, 1.6059043836821613e-10); ps = fma(ps, r2, -2.505210838544172e-08);
    ps = fma(ps, r2, 2.7557319223985893e-06); ps = fma(ps, r2, -0.0001984126984126984); ps = fma(ps, r2, 0.008333333333333333);
    ps = fma(ps, r2, -0.16666666666666666);
    const double sv = fma(r * r2, ps, r);
    double pc = 1.5619206968586225e-16;                 // 1/18!
    pc = fma(pc, r2, -4.779477332387385e-14); pc = fma(pc, r2, 1.1470745597729725e-11); pc = fma(pc, r2, -2.08767569878681e-09);
    pc = fma(pc, r2, 2.755731922398589e-07); pc = fma(pc, r2, -2.48015873015873e-05); pc = fma(pc, r2, 0.001388888888888889);
    pc = fma(pc, r2, -0.041666666666666664); pc = fma(pc, r2, 0.5);
    const double cv = fma(-r2, pc, 1.0);
    const int q = (int) k;
    const bool odd = (q & 1) != 0;
    double s0 = odd ? cv : sv, c0 = odd ? sv : cv;
    if (q & 2) s0 = -s0;
    if ((q + 1) & 2) c0 = -c0;
    sn = s0; cs = c0;
}
// atan2(y, x), not both zero (|error| < 6e-16 absolute): t = min/max in [0, 1], atan(t) = atan(c) + atan((t - c) / (1 + t c)) with
// c = round(4 t) / 4 (|argument| <= 1/8, Taylor to s^17), then the octant / quadrant / sign of IEEE atan2 (signed zeros included)
__device__ __forceinline__ double nq_atan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double t = mn / mx;
    const double ci = rint(t * 4.0);
    const double c = ci * 0.25;
    const double sr = (t - c) / fma(t, c, 1.0);
    const double s2 = sr * sr;
    double p = 0.058823529411764705;                    // 1/17
    p = fma(p, s2, -0.06666666666666667); p = fma(p, s2, 0.07692307692307693); p = fma(p, s2, -0.09090909090909091);
    p = fma(p, s2, 0.1111111111111111); p = fma(p, s2, -0.14285714285714285); p = fma(p, s2, 0.2); p = fma(p, s2, -0.3333333333333333);
    const double a = fma(sr * s2, p, sr);
    const double base = ci < 0.5 ? 0.0 : ci < 1.5 ? 0.24497866312686414 : ci < 2.5 ? 0.4636476090008061 : ci < 3.5 ? 0.6435011087932844
                                                                                                                   : 0.7853981633974483;
    double r = base + a;
    if (ay > ax) r = 1.5707963267948966 - r;
    if (x < 0.0 || (x == 0.0 && __builtin_signbit(x))) r = 3.141592653589793 - r;
    return copysign(r, y);
}
// do (v - m) and (v + m) narrow to the same float?
__device__ __forceinline__ bool nq_narrow_safe(double v, double m) { return (float) (v - m) == (float) (v + m); }

// deltaL', deltaC', deltaH', R_T of the pair (NQ/CIELABConvertor.java:91-98, :100-118, :120-185, :187-194); false = undecided
__device__ __forceinline__ bool ciede_terms_fast(const Lab& lab1, const Lab& lab2, float& deltaL, float& deltaC, float& deltaH, float& rt_out) {
    bool ok = true;
    // :91-98 (sqrt is correctly rounded everywhere: identical values)
    deltaL = L_prime_div_k_L_S_L(lab1, lab2);
    // :100-118
    const float pow25To7f = 6103515625.0f;
    const float C1 = (float) sqrt((double) ((lab1.A * lab1.A) + (lab1.B * lab1.B)));
    const float C2 = (float) sqrt((double) ((lab2.A * lab2.A) + (lab2.B * lab2.B)));
    const float barC = (C1 + C2) / 2.0f;
    const double barC7 = nq_pow7((double) barC);
    const double ratioC = barC7 / (barC7 + (double) pow25To7f);
    const double Gd = (double) 0.5f * (1 - sqrt(ratioC));
    // Two evaluations whose barC^7 differ by <= 3.5 ulp feed quotients that differ by 3.5 ulp * (1 - ratio) relative, i.e. after the
    // (correctly rounded) division and square root by at most one ulp of a number <= 1 each: |difference of Gd| <= 2.2e-16 +
    // 2e-16 (1 - ratio).  The subtraction 1 - sqrt cancels absolutely, not relatively, so the margin is absolute.
    ok = ok && nq_narrow_safe(Gd, 6e-16 + 1e-15 * (1.0 - ratioC));
    const float G = (float) Gd;
    const double a1Prime = (1.0 + G) * lab1.A, a2Prime = (1.0 + G) * lab2.A;
    const double CPrime1 = sqrt((a1Prime * a1Prime) + (double) (lab1.B * lab1.B));
    const double CPrime2 = sqrt((a2Prime * a2Prime) + (double) (lab2.B * lab2.B));
    {
        const float deltaCPrime = (float) CPrime2 - (float) CPrime1;
        const float barCPrimeF = ((float) CPrime1 + (float) CPrime2) / 2.0f;
        const float S_C = 1 + (0.045f * barCPrimeF);
        deltaC = deltaCPrime / S_C;
    }
    // :120-185
    const float deg360InRad = deg2Rad(360.0), deg180InRad = deg2Rad(180.0);
    const double CPrimeProduct = CPrime1 * CPrime2;
    double hPrime1 = 0.0, hPrime2 = 0.0;
    if (!((double) lab1.B == 0.0 && a1Prime == 0.0)) { hPrime1 = nq_atan2((double) lab1.B, a1Prime); if (hPrime1 < 0) hPrime1 += deg360InRad; }
    if (!((double) lab2.B == 0.0 && a2Prime == 0.0)) { hPrime2 = nq_atan2((double) lab2.B, a2Prime); if (hPrime2 < 0) hPrime2 += deg360InRad; }
    const double AERR = 4e-15;                           // bound on the error of an angle / of a sum or difference of two
    double deltahPrime = 0;
    if (CPrimeProduct != 0.0) {
        deltahPrime = hPrime2 - hPrime1;
        ok = ok && fabs(fabs(deltahPrime) - (double) deg180InRad) > 16 * AERR;      // (also decides :166 below)
        if (deltahPrime < -deg180InRad) deltahPrime += deg360InRad;
        else if (deltahPrime > deg180InRad) deltahPrime -= deg360InRad;
    }
    double sh, ch_unused;
    nq_sincos(deltahPrime / 2.0, sh, ch_unused);
    const double deltaHPrime = 2.0 * sqrt(CPrimeProduct) * sh;
    const double hPrimeSum = hPrime1 + hPrime2;
    double barhPrime;
    if (CPrimeProduct == 0.0) barhPrime = hPrimeSum;
    else {
        if (fabs(hPrime1 - hPrime2) <= deg180InRad) barhPrime = hPrimeSum / 2.0;
        else {
            ok = ok && fabs(hPrimeSum - (double) deg360InRad) > 16 * AERR;
            if (hPrimeSum < deg360InRad) barhPrime = (hPrimeSum + deg360InRad) / 2.0;
            else barhPrime = (hPrimeSum - deg360InRad) / 2.0;
        }
    }
    const double barCPrime = (CPrime1 + CPrime2) / 2.0;
    const double bh = barhPrime;
    double s1, c1, s2_, c2, s3, c3, s4, c4;
    nq_sincos(bh - deg2Rad(30.0), s1, c1);
    nq_sincos(2.0 * bh, s2_, c2);
    nq_sincos((3.0 * bh) + deg2Rad(6.0), s3, c3);
    nq_sincos((4.0 * bh) - deg2Rad(63.0), s4, c4);
    const double T = 1.0 - (0.17 * c1) + (0.24 * c2) + (0.32 * c3) - (0.20 * c4);
    const double S_H = 1 + ((double) 0.015f * barCPrime * T);
    const double Hd = deltaHPrime / S_H;
    // error of Hd: sin(dh/2) carries the absolute angle error AERR (relative AERR / |dh/2| when the hues nearly coincide), T an
    // absolute 2e-15 (four cosines of arguments with error <= 4 AERR), S_H <= 0.015 * 182 * that: relative 6e-15
    {
        const double rel = 1e-13 + (deltahPrime != 0.0 ? 4 * AERR / fabs(deltahPrime) : 0.0);
        ok = ok && rel < 1e-9 && nq_narrow_safe(Hd, fabs(Hd) * rel);
    }
    deltaH = (float) Hd;
    // :187-194
    const double zz = (barhPrime - deg2Rad(275.0)) / deg2Rad(25.0);
    const double deltaTheta = deg2Rad(30.0) * nq_exp_neg(-(zz * zz));
    const double bc7 = nq_pow7(barCPrime);
    const double R_C = 2.0 * sqrt(bc7 / (bc7 + 6103515625.0));
    double s5, c5;
    nq_sincos(2.0 * deltaTheta, s5, c5);
    const double rtv = ((-s5) * R_C) * deltaC * deltaH;
    // error of rtv: the exponent zz^2 <= 8100 carries 2 |zz| AERR / 0.436 absolute -> relative 6e-13 on exp, the rest is ulps
    ok = ok && nq_narrow_safe(rtv, fabs(rtv) * 2e-12);
    rt_out = (float) rtv;
    (void) ch_unused; (void) s1; (void) s2_; (void) s3; (void) s4; (void) c5;
    return ok;
}
// ---- the same pass spread over the four lanes of a quad (lanes 4k .. 4k+3 hold the SAME pair; r = lane & 3) ----
// The evaluation is one dependent chain of ~1250 instructions whatever the number of active lanes, and the merge loop waits for it in
// every find_nn.  Here the independent transcendental evaluations of a pair run side by side in the quad -- the three square roots of
// stage 1, the two C', the two atan2, the four cosines of T, the two sines -- each lane evaluating the SAME function on its own
// argument, results broadcast with DPP quad_perm; everything else is computed redundantly by the four lanes.  Every value is produced
// by the same operations on the same operands as in ciede_terms_fast, so the outputs are identical (nq_selftest_ciede compares them).
template <int K> __device__ __forceinline__ int quad_bcast_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}
template <int K> __device__ __forceinline__ float quad_bcast(float x) { return __int_as_float(quad_bcast_i<K>(__float_as_int(x))); }
template <int K> __device__ __forceinline__ double quad_bcast(double x) {
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned) quad_bcast_i<K>((int) (unsigned) b), hi = (unsigned) quad_bcast_i<K>((int) (unsigned) (b >> 32));
    return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}
__device__ __forceinline__ bool ciede_terms_fast_quad(const Lab& lab1, const Lab& lab2, int r, float& deltaL, float& deltaC, float& deltaH, float& rt_out) {
    bool ok = true;
    const bool second = (r & 1) != 0;
    const float Aq = second ? lab2.A : lab1.A, Bq = second ? lab2.B : lab1.B;         // lanes 0, 2: colour 1; lanes 1, 3: colour 2
    // stage 1, three square roots at once: C1 (lane 0), C2 (lane 1), sqrt(20 + p) of S_L (lane 2)        :91-98, :100-104
    const float deltaLPrime = lab2.L - lab1.L;
    const float barLPrime = (lab1.L + lab2.L) / 2.0f;
    const double pL = sqr((double) (barLPrime - 50.0f));
    const double arg1 = r == 2 ? 20 + pL : (double) ((Aq * Aq) + (Bq * Bq));
    const double root1 = sqrt(arg1);
    const float C1 = quad_bcast<0>((float) root1), C2 = quad_bcast<1>((float) root1);
    const double rootL = quad_bcast<2>(root1);
    {
        const float S_L = (float) (1 + (((double) 0.015f * pL) / rootL));
        deltaL = deltaLPrime / (1.0f * S_L);
    }
    const float pow25To7f = 6103515625.0f;
    const float barC = (C1 + C2) / 2.0f;
    const double barC7 = nq_pow7((double) barC);
    const double ratioC = barC7 / (barC7 + (double) pow25To7f);
    const double Gd = (double) 0.5f * (1 - sqrt(ratioC));
    ok = ok && nq_narrow_safe(Gd, 6e-16 + 1e-15 * (1.0 - ratioC));
    const float G = (float) Gd;
    // stage 2: a' and C' of both colours                                                                   :106-112
    const double aq = (1.0 + G) * Aq;
    const double CPq = sqrt((aq * aq) + (double) (Bq * Bq));
    const double CPrime1 = quad_bcast<0>(CPq), CPrime2 = quad_bcast<1>(CPq);
    {
        const float deltaCPrime = (float) CPrime2 - (float) CPrime1;
        const float barCPrimeF = ((float) CPrime1 + (float) CPrime2) / 2.0f;
        const float S_C = 1 + (0.045f * barCPrimeF);
        deltaC = deltaCPrime / S_C;
    }
    // stage 3: both hue angles                                                                              :120-140
    const float deg360InRad = deg2Rad(360.0), deg180InRad = deg2Rad(180.0);
    const double CPrimeProduct = CPrime1 * CPrime2;
    double hq = 0.0;
    if (!((double) Bq == 0.0 && aq == 0.0)) { hq = nq_atan2((double) Bq, aq); if (hq < 0) hq += deg360InRad; }
    const double hPrime1 = quad_bcast<0>(hq), hPrime2 = quad_bcast<1>(hq);
    const double AERR = 4e-15;
    double deltahPrime = 0;
    if (CPrimeProduct != 0.0) {
        deltahPrime = hPrime2 - hPrime1;
        ok = ok && fabs(fabs(deltahPrime) - (double) deg180InRad) > 16 * AERR;
        if (deltahPrime < -deg180InRad) deltahPrime += deg360InRad;
        else if (deltahPrime > deg180InRad) deltahPrime -= deg360InRad;
    }
    const double hPrimeSum = hPrime1 + hPrime2;
    double barhPrime;
    if (CPrimeProduct == 0.0) barhPrime = hPrimeSum;
    else {
        if (fabs(hPrime1 - hPrime2) <= deg180InRad) barhPrime = hPrimeSum / 2.0;
        else {
            ok = ok && fabs(hPrimeSum - (double) deg360InRad) > 16 * AERR;
            if (hPrimeSum < deg360InRad) barhPrime = (hPrimeSum + deg360InRad) / 2.0;
            else barhPrime = (hPrimeSum - deg360InRad) / 2.0;
        }
    }
    const double barCPrime = (CPrime1 + CPrime2) / 2.0;
    const double bh = barhPrime;
    // stage 4: the four cosines of T, one per lane                                                          :159-163
    const double arg4 = r == 0 ? bh - deg2Rad(30.0) : r == 1 ? 2.0 * bh : r == 2 ? (3.0 * bh) + deg2Rad(6.0) : (4.0 * bh) - deg2Rad(63.0);
    double s4q, c4q;
    nq_sincos(arg4, s4q, c4q);
    const double c1 = quad_bcast<0>(c4q), c2 = quad_bcast<1>(c4q), c3 = quad_bcast<2>(c4q), c4 = quad_bcast<3>(c4q);
    const double T = 1.0 - (0.17 * c1) + (0.24 * c2) + (0.32 * c3) - (0.20 * c4);
    const double S_H = 1 + ((double) 0.015f * barCPrime * T);
    // stage 5: exp of R_T (all lanes), then the two sines: sin(dh'/2) (lanes 0, 2), sin(2 dTheta) (lanes 1, 3)   :142, :187-194
    const double zz = (barhPrime - deg2Rad(275.0)) / deg2Rad(25.0);
    const double deltaTheta = deg2Rad(30.0) * nq_exp_neg(-(zz * zz));
    const double arg5 = second ? 2.0 * deltaTheta : deltahPrime / 2.0;
    double s5q, c5q;
    nq_sincos(arg5, s5q, c5q);
    const double sh = quad_bcast<0>(s5q), s5 = quad_bcast<1>(s5q);
    // stage 6, two square roots at once: sqrt(C1' C2') (lanes 0, 2), sqrt of the R_C ratio (lanes 1, 3)
    const double bc7 = nq_pow7(barCPrime);
    const double arg6 = second ? bc7 / (bc7 + 6103515625.0) : CPrimeProduct;
    const double root6 = sqrt(arg6);
    const double rootP = quad_bcast<0>(root6), rootR = quad_bcast<1>(root6);
    const double deltaHPrime = 2.0 * rootP * sh;
    const double Hd = deltaHPrime / S_H;
    {
        const double rel = 1e-13 + (deltahPrime != 0.0 ? 4 * AERR / fabs(deltahPrime) : 0.0);
        ok = ok && rel < 1e-9 && nq_narrow_safe(Hd, fabs(Hd) * rel);
    }
    deltaH = (float) Hd;
    const double R_C = 2.0 * rootR;
    const double rtv = ((-s5) * R_C) * deltaC * deltaH;
    ok = ok && nq_narrow_safe(rtv, fabs(rtv) * 2e-12);
    rt_out = (float) rtv;
    (void) s4q; (void) c5q;
    return ok;
}

// the same four floats by the literal functions (device library)
__device__ __forceinline__ void ciede_terms_literal(const Lab& lab1, const Lab& lab2, float& deltaL, float& deltaC, float& deltaH, float& rt_out) {
    deltaL = L_prime_div_k_L_S_L(lab1, lab2);
    double a1Prime, a2Prime, CPrime1, CPrime2, barCPrime, barhPrime;
    deltaC = C_prime_div_k_L_S_L(lab1, lab2, a1Prime, a2Prime, CPrime1, CPrime2);
    deltaH = H_prime_div_k_L_S_L(lab1, lab2, a1Prime, a2Prime, CPrime1, CPrime2, barCPrime, barhPrime);
    rt_out = R_T(barCPrime, barhPrime, deltaC, deltaH);
}

// ... out of line: the fast pass declines 0.2-3 % of the pairs, and the device library's pow / atan2 / sin / cos / exp inlined at every
// call site of the exact evaluation (~1500 instructions each, with their own register appetite) were most of the merge kernel's code
struct CiedeTerms { float dL, dC, dH, rt; };
__device__ __attribute__((noinline)) CiedeTerms ciede_terms_literal_ool(float L1, float A1, float B1, float L2, float A2, float B2) {
    Lab a, b;
    a.alpha = 255.f; a.L = L1; a.A = A1; a.B = B1;
    b.alpha = 255.f; b.L = L2; b.A = A2; b.B = B2;
    CiedeTerms t;
    ciede_terms_literal(a, b, t.dL, t.dC, t.dH, t.rt);
    return t;
}

// :215-227 / :229-238
__device__ __forceinline__ double color2Y(int c) {
    double sr = g_tab.gamma[c_red(c)], sg = g_tab.gamma[c_green(c)], sb = g_tab.gamma[c_blue(c)];
    return sr * 0.2126 + sg * 0.7152 + sb * 0.0722;
}
__device__ __forceinline__ double Y_Diff(int c1, int c2) {
    double y = color2Y(c1), y2 = color2Y(c2);
    return fabs(y2 - y) * 100;
}
__device__ __forceinline__ double color2U(int c) { return -0.09991 * c_red(c) - 0.33609 * c_green(c) + 0.436 * c_blue(c); }
__device__ __forceinline__ double U_Diff(int c1, int c2) { return fabs(color2U(c2) - color2U(c1)); }

// table-parameterised forms for the dither chains (tables in LDS); same arithmetic
__device__ __forceinline__ double color2Y_t(int c, const double* __restrict__ gamma) {
    const double sr = gamma[c_red(c)], sg = gamma[c_green(c)], sb = gamma[c_blue(c)];
    return sr * 0.2126 + sg * 0.7152 + sb * 0.0722;
}
__device__ __forceinline__ double Y_Diff_y(double y, double y2) { return fabs(y2 - y) * 100; }
// BlueNoise.diffuse with the clamp done in float: (int) min(255, max((double) f, 0.0)) == (int) fminf(255.f, fmaxf(f, 0.f))
// (float -> double is exact, NaN cannot occur)
__device__ __forceinline__ int blue_diffuse_t(int pixel, int qPixel, float weight, float strength, int x, int y,
                                              const signed char* __restrict__ blue) {
    int r_pix = c_red(pixel), g_pix = c_green(pixel), b_pix = c_blue(pixel), a_pix = c_alpha(pixel);
    float adj = (blue[(x & 63) | (y & 63) << 6] + 0.5f) / 127.5f;
    adj += (((x + y) & 1) - 0.5f) * strength / 8.0f;
    adj *= weight;
    r_pix = (int) fminf(255.0f, fmaxf(r_pix + (adj * (r_pix - c_red(qPixel))), 0.0f));
    g_pix = (int) fminf(255.0f, fmaxf(g_pix + (adj * (g_pix - c_green(qPixel))), 0.0f));
    b_pix = (int) fminf(255.0f, fmaxf(b_pix + (adj * (b_pix - c_blue(qPixel))), 0.0f));
    a_pix = (int) fminf(255.0f, fmaxf(a_pix + (adj * (a_pix - c_alpha(qPixel))), 0.0f));
    return c_argb(a_pix, r_pix, g_pix, b_pix);
}

// NQ/BlueNoise.java:180-197
__device__ __forceinline__ int blue_diffuse(int pixel, int qPixel, float weight, float strength, int x, int y) {
    int r_pix = c_red(pixel), g_pix = c_green(pixel), b_pix = c_blue(pixel), a_pix = c_alpha(pixel);
    float adj = (g_tab.blue[(x & 63) | (y & 63) << 6] + 0.5f) / 127.5f;
    adj += (((x + y) & 1) - 0.5f) * strength / 8.0f;
    adj *= weight;
    r_pix = j_d2i(fmin(255.0, fmax((double) (r_pix + (adj * (r_pix - c_red(qPixel)))), 0.0)));
    g_pix = j_d2i(fmin(255.0, fmax((double) (g_pix + (adj * (g_pix - c_green(qPixel)))), 0.0)));
    b_pix = j_d2i(fmin(255.0, fmax((double) (b_pix + (adj * (b_pix - c_blue(qPixel)))), 0.0)));
    a_pix = j_d2i(fmin(255.0, fmax((double) (a_pix + (adj * (a_pix - c_alpha(qPixel)))), 0.0)));
    return c_argb(a_pix, r_pix, g_pix, b_pix);
}

// GilbertCurve.normalDistribution (NQ/GilbertCurve.java:114-123)
__device__ __forceinline__ float normalDistribution(float x, float peak) {
    const float mean = .5f, stdDev = .1f;
    double exponent = -sqr((double) (x - mean)) / (2 * sqr((double) stdDev));
    double pdf = (1 / (stdDev * sqrt(2 * NQ_PI))) * exp(exponent);
    double maxPdf = 1 / (stdDev * sqrt(2 * NQ_PI));
    double scaledPdf = (pdf / maxPdf) * peak;
    return (float) fmax(0.0, fmin((double) peak, scaledPdf));
}

// ---- java.util.Random ---------------------------------------------------------------------------
#define NQ_JR_MULT 0x5DEECE66DLL
#define NQ_JR_MASK ((1LL << 48) - 1)
__host__ __device__ __forceinline__ long long jr_seed(long long seed) { return (seed ^ NQ_JR_MULT) & NQ_JR_MASK; }
__device__ __forceinline__ int jr_next(long long& st, int bits) {
    st = (long long) (((unsigned long long) st * (unsigned long long) NQ_JR_MULT + 0xBULL) & (unsigned long long) NQ_JR_MASK);
    return (int) (st >> (48 - bits));
}
__device__ __forceinline__ int jr_next_int_bound(long long& st, int bound) {
    int r = jr_next(st, 31);
    int m = bound - 1;
    if ((bound & m) == 0) r = (int) (((long long) bound * (long long) r) >> 31);
    else {
        for (int u = r; (int) ((unsigned) (u - (r = u % bound)) + (unsigned) m) < 0; u = jr_next(st, 31)) { }
    }
    return r;
}
// stream selector of the parallel decomposition (documented in DESIGN.md): splitmix64 finaliser
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// NQ/PnnQuantizer.java:26-30
__device__ __constant__ const float k_coeffs[3][3] = {
    {0.299f, 0.587f, 0.114f},
    {-0.14713f, -0.28886f, 0.436f},
    {0.615f, -0.51499f, -0.10001f}
};

// ------------------------------------------------------------------------------------------------
// palette staged in LDS: ARGB words + (LAB) the Lab of every entry, getLab(palette[i]) of
// NQ/PnnLABQuantizer.java:352
// ------------------------------------------------------------------------------------------------
struct PalView {
    const int* argb;     // [K]
    const float* L;      // [K] (LAB only)
    const float* A;
    const float* B;
    const double* gamma;        // gammaToLinear[256]: LDS copy when staged with tables, else the constant-memory table
    const signed char* blue;    // TELL_BLUE_NOISE[4096]: likewise
};

// ---- candidate lists per 5-6-5 colour cell (built by nq_lists.inc; exactness argument there) -------------------------
#define NQ_LIST_CAP 32
#define NQ_LIST_FULLSCAN 255
struct CellLists {
    const unsigned char* closest;        // [65536][NQ_LIST_CAP] palette indices in ascending order, or nullptr (full scans)
    const unsigned char* closestCount;   // [65536]; NQ_LIST_FULLSCAN = scan the whole palette
    const unsigned char* nearest;        // same for nearestColorIndex (LAB, K > 32, no semi-transparency), or nullptr
    const unsigned char* nearestCount;
};
struct CellList {                        // one cell's list in four 64-bit words
    unsigned long long w0, w1, w2, w3;
    int n;
    __device__ __forceinline__ int at(int i) const {
        const unsigned long long w = i < 16 ? (i < 8 ? w0 : w1) : (i < 24 ? w2 : w3);
        return (int) ((w >> ((i & 7) * 8)) & 0xFF);
    }
};
__device__ __forceinline__ int cell_of(int c) { return (c_red(c) & 0xF8) << 8 | (c_green(c) & 0xFC) << 3 | (c_blue(c) >> 3); }
__device__ __forceinline__ CellList load_cell_list(const unsigned char* __restrict__ lists, const unsigned char* __restrict__ counts, int cell) {
    CellList l;
    l.n = counts[cell];
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(lists + (size_t) cell * NQ_LIST_CAP);
    const ulonglong2 a = p[0], b = p[1];
    l.w0 = a.x; l.w1 = a.y; l.w2 = b.x; l.w3 = b.y;
    return l;
}

// ---- nearestColorIndex, cache-miss semantics ----------------------------------------------------
// RGB: NQ/PnnQuantizer.java:276-310
__device__ __forceinline__ int nearest_rgb(const DevParams& P, const PalView& pal, int c, const CellLists* lists = nullptr) {
    const int K = P.K;
    int k = 0;
    if (c_alpha(c) <= 0xF) c = P.transparentColor;
    if (K > 2 && P.hasAlpha && c_alpha(c) > 0xF) k = 1;
    double pr = P.PR, pg = P.PG, pb = P.PB, pa = P.PA;
    if (K < 3) pr = pg = pb = pa = 1;
    double mindist = 2147483647.0;
    const int ca = c_alpha(c), cr = c_red(c), cg = c_green(c), cb = c_blue(c);
    if (lists && lists->nearest && ca == 255 && k == 0) {
        // exact candidate list of the colour's cell (nq_lists.inc), scanned in index order with the unchanged arithmetic
        const CellList cl = load_cell_list(lists->nearest, lists->nearestCount, cell_of(c));
        if (cl.n != NQ_LIST_FULLSCAN) {
            for (int t = 0; t < cl.n; ++t) {
                const int i = cl.at(t);
                const int c2 = pal.argb[i];
                double curdist = pa * sqr((double) (c_alpha(c2) - ca));
                curdist += pr * sqr((double) (c_red(c2) - cr));
                curdist += pg * sqr((double) (c_green(c2) - cg));
                curdist += pb * sqr((double) (c_blue(c2) - cb));
                if (curdist > mindist) continue;          // (the partial-sum gates of the reference only skip what this one skips)
                mindist = curdist;
                k = i;
            }
            return k;
        }
    }
    for (int i = k; i < K; ++i) {
        int c2 = pal.argb[i];
        double curdist = pa * sqr((double) (c_alpha(c2) - ca));
        if (curdist > mindist) continue;
        curdist += pr * sqr((double) (c_red(c2) - cr));
        if (curdist > mindist) continue;
        curdist += pg * sqr((double) (c_green(c2) - cg));
        if (curdist > mindist) continue;
        curdist += pb * sqr((double) (c_blue(c2) - cb));
        if (curdist > mindist) continue;
        mindist = curdist;
        k = i;
    }
    return k;
}

// LAB: NQ/PnnLABQuantizer.java:337-401
// `seen` (REFERENCE_SEQUENTIAL only, else nullptr): byte per palette entry, set where the reference calls getLab(palette[i])
// (NQ/PnnLABQuantizer.java:345-350: every entry that passes the alpha gate) -- those colours enter pixelMap
__device__ __forceinline__ int nearest_lab(const DevParams& P, const PalView& pal, int c, const CellLists* lists = nullptr,
                                           unsigned char* seen = nullptr) {
    const int K = P.K;
    int k = 0;
    if (c_alpha(c) <= 0xF) c = P.transparentColor;
    if (K > 2 && P.hasAlpha && c_alpha(c) > 0xF) k = 1;
    double mindist = 2147483647.0;
    const Lab lab1 = RGB2LAB_fast(c, pal.gamma);
    const int ca = c_alpha(c);
    if (seen && !P.hasSemi) for (int i = k; i < K; ++i) seen[i] = 1;     // without the alpha term every entry passes the gate
    if (K <= 4) {
        const int cr = c_red(c), cg = c_green(c), cb = c_blue(c);
        for (int i = k; i < K; ++i) {
            int c2 = pal.argb[i];
            double curdist = P.hasSemi ? sqr((double) (c_alpha(c2) - ca)) / g_tab.exp1_5 : 0;
            if (curdist > mindist) continue;
            if (seen) seen[i] = 1;
            curdist = sqr((double) (c_red(c2) - cr)) + sqr((double) (c_green(c2) - cg)) + sqr((double) (c_blue(c2) - cb));
            if (P.hasSemi) curdist += sqr((double) (c_alpha(c2) - ca));
            if (curdist > mindist) continue;
            mindist = curdist;
            k = i;
        }
    } else if (P.hasSemi || K < 16) {
        for (int i = k; i < K; ++i) {
            double curdist = P.hasSemi ? sqr((double) (c_alpha(pal.argb[i]) - ca)) / g_tab.exp1_5 : 0;
            if (curdist > mindist) continue;
            if (seen) seen[i] = 1;
            curdist += sqr((double) (pal.L[i] - lab1.L));
            if (curdist > mindist) continue;
            curdist += sqr((double) (pal.A[i] - lab1.A));
            if (curdist > mindist) continue;
            curdist += sqr((double) (pal.B[i] - lab1.B));
            if (curdist > mindist) continue;
            mindist = curdist;
            k = i;
        }
    } else if (K > 32) {
        bool done = false;
        if (lists && lists->nearest && !(P.hasAlpha && k == 0)) {
            const CellList cl = load_cell_list(lists->nearest, lists->nearestCount, cell_of(c));
            if (cl.n != NQ_LIST_FULLSCAN) {
                float minf = 3.0e38f;       // float shadow of mindist, only used to skip clearly worse entries
                for (int t = 0; t < cl.n; ++t) {
                    const int i = cl.at(t);
                    // float32 reject: its error (< 1e-5 relative) is far inside the margin, so only entries that cannot be
                    // the minimum (nor tie with it) skip the exact f64 evaluation
                    const float dLf = fabsf(pal.L[i] - lab1.L), dAf = pal.A[i] - lab1.A, dBf = pal.B[i] - lab1.B;
                    const float d32 = dLf + __builtin_amdgcn_sqrtf(dAf * dAf + dBf * dBf);
                    if (d32 > minf * 1.001f + 1e-3f) continue;
                    minf = fminf(minf, d32);
                    double curdist = (double) dLf;
                    if (curdist > mindist) continue;
                    curdist += sqrt(sqr((double) (pal.A[i] - lab1.A)) + sqr((double) (pal.B[i] - lab1.B)));
                    if (curdist > mindist) continue;
                    mindist = curdist;
                    k = i;
                }
                done = true;
            }
        }
        if (!done)
        for (int i = k; i < K; ++i) {
            double curdist = (double) fabsf(pal.L[i] - lab1.L);     // hasSemi is false here: curdist starts at 0
            if (curdist > mindist) continue;
            curdist += sqrt(sqr((double) (pal.A[i] - lab1.A)) + sqr((double) (pal.B[i] - lab1.B)));
            if (curdist > mindist) continue;
            mindist = curdist;
            k = i;
        }
    } else {
        for (int i = k; i < K; ++i) {
            Lab lab2; lab2.alpha = (float) c_alpha(pal.argb[i]); lab2.L = pal.L[i]; lab2.A = pal.A[i]; lab2.B = pal.B[i];
            double curdist = 0;
            float deltaL = L_prime_div_k_L_S_L(lab1, lab2);
            curdist += sqr((double) deltaL);
            if (curdist > mindist) continue;
            double a1Prime, a2Prime, CPrime1, CPrime2;
            float deltaC = C_prime_div_k_L_S_L(lab1, lab2, a1Prime, a2Prime, CPrime1, CPrime2);
            curdist += sqr((double) deltaC);
            if (curdist > mindist) continue;
            double barCPrime, barhPrime;
            float deltaH = H_prime_div_k_L_S_L(lab1, lab2, a1Prime, a2Prime, CPrime1, CPrime2, barCPrime, barhPrime);
            curdist += sqr((double) deltaH);
            if (curdist > mindist) continue;
            curdist += (double) R_T(barCPrime, barhPrime, deltaC, deltaH);
            if (curdist > mindist) continue;
            mindist = curdist;
            k = i;
        }
    }
    return k;
}

__device__ __forceinline__ int nearest_any(const DevParams& P, const PalView& pal, int c, const CellLists* lists = nullptr,
                                           unsigned char* seen = nullptr) {
    return P.kind == 0 ? nearest_rgb(P, pal, c, lists) : nearest_lab(P, pal, c, lists, seen);
}

// ---- closest[] tuple ------------------------------------------------------------------------------
// one palette entry of the RGB loop: NQ/PnnQuantizer.java:330-356
__device__ __forceinline__ void closest_step_rgb(const DevParams& P, const PalView& pal, int k, int ca, int cr, int cg, int cb,
                                                 double pr, double pg, double pb, double pa, int closest[4]) {
    const int c2 = pal.argb[k];
    double err = pr * sqr((double) (c_red(c2) - cr));
    if (err >= closest[3]) return;
    err += pg * sqr((double) (c_green(c2) - cg));
    if (err >= closest[3]) return;
    err += pb * sqr((double) (c_blue(c2) - cb));
    if (err >= closest[3]) return;
    if (P.hasSemi) err += pa * sqr((double) (c_alpha(c2) - ca));
    if (err < closest[2]) {
        closest[1] = closest[0]; closest[3] = closest[2];
        closest[0] = k; closest[2] = j_d2i(err);
    } else if (err < closest[3]) {
        closest[1] = k; closest[3] = j_d2i(err);
    }
}
// RGB: NQ/PnnQuantizer.java:322-360
__device__ __forceinline__ void closest_tuple_rgb(const DevParams& P, const PalView& pal, int c, int closest[4],
                                                  const CellLists* lists = nullptr) {
    const int K = P.K;
    closest[0] = closest[1] = 0;
    closest[2] = closest[3] = 2147483647;
    double pr = P.PR, pg = P.PG, pb = P.PB, pa = P.PA;
    if (K < 3) pr = pg = pb = pa = 1;
    const int ca = c_alpha(c), cr = c_red(c), cg = c_green(c), cb = c_blue(c);
    bool done = false;
    if (lists && lists->closest) {
        const CellList cl = load_cell_list(lists->closest, lists->closestCount, cell_of(c));
        if (cl.n != NQ_LIST_FULLSCAN) {
            for (int t = 0; t < cl.n; ++t) closest_step_rgb(P, pal, cl.at(t), ca, cr, cg, cb, pr, pg, pb, pa, closest);
            done = true;
        }
    }
    if (!done) for (int k = 0; k < K; ++k) closest_step_rgb(P, pal, k, ca, cr, cg, cb, pr, pg, pb, pa, closest);
    if (closest[3] == 2147483647) closest[1] = closest[0];
}
// one palette entry of the LAB loop: NQ/PnnLABQuantizer.java:419-457
__device__ __forceinline__ void closest_step_lab(const DevParams& P, const PalView& pal, int k, int ca, int cr, int cg, int cb,
                                                 double wr, double wg, double wb, double ratio, int closest[4]) {
    const int c2 = pal.argb[k];
    const int dr = c_red(c2) - cr, dg = c_green(c2) - cg, db = c_blue(c2) - cb;
    if (ratio < 0) {
        // the ratio ladder of the reference can go negative (NQ/PnnLABQuantizer.java:259-264); the YUV terms then LOWER the sum
        // and the gates decide: literal evaluation (:419-457)
        double err = wr * sqr((double) dr);
        if (err >= closest[3]) return;
        err += wg * sqr((double) dg);
        if (err >= closest[3]) return;
        err += wb * sqr((double) db);
        if (err >= closest[3]) return;
        if (P.hasSemi) err += P.PA * sqr((double) (c_alpha(c2) - ca));
        for (int i = 0; i < 3; ++i) {
            err += ratio * sqr((double) (k_coeffs[i][0] * dr));
            if (err >= closest[3]) break;
            err += ratio * sqr((double) (k_coeffs[i][1] * dg));
            if (err >= closest[3]) break;
            err += ratio * sqr((double) (k_coeffs[i][2] * db));
            if (err >= closest[3]) break;
        }
        if (err < closest[2]) {
            closest[1] = closest[0]; closest[3] = closest[2];
            closest[0] = k; closest[2] = j_d2i(err);
        } else if (err < closest[3]) {
            closest[1] = k; closest[3] = j_d2i(err);
        }
        return;
    }
    // Every term is >= 0 and every gate of the reference (`if (err >= closest[3]) break`) only leaves early a candidate whose
    // final err would be >= closest[3] as well, i.e. one that neither branch below takes: the sum is evaluated straight through,
    // same operations in the same order, and compared once (no divergent exits between twelve short terms).
    double err = wr * sqr((double) dr);
    err += wg * sqr((double) dg);
    err += wb * sqr((double) db);
    if (P.hasSemi) err += P.PA * sqr((double) (c_alpha(c2) - ca));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        err += ratio * sqr((double) (k_coeffs[i][0] * dr));
        err += ratio * sqr((double) (k_coeffs[i][1] * dg));
        err += ratio * sqr((double) (k_coeffs[i][2] * db));
    }
    if (err < closest[2]) {
        closest[1] = closest[0]; closest[3] = closest[2];
        closest[0] = k; closest[2] = j_d2i(err);
    } else if (err < closest[3]) {
        closest[1] = k; closest[3] = j_d2i(err);
    }
}
// LAB: NQ/PnnLABQuantizer.java:415-461
__device__ __forceinline__ void closest_tuple_lab(const DevParams& P, const PalView& pal, int c, int closest[4],
                                                  const CellLists* lists = nullptr) {
    const int K = P.K;
    closest[0] = closest[1] = 0;
    closest[2] = closest[3] = 2147483647;
    const double ratio = P.ratio;
    const double wr = P.PR * (1 - ratio), wg = P.PG * (1 - ratio), wb = P.PB * (1 - ratio);
    const int ca = c_alpha(c), cr = c_red(c), cg = c_green(c), cb = c_blue(c);
    bool done = false;
    if (lists && lists->closest) {
        const CellList cl = load_cell_list(lists->closest, lists->closestCount, cell_of(c));
        if (cl.n != NQ_LIST_FULLSCAN) {
            for (int t = 0; t < cl.n; ++t) closest_step_lab(P, pal, cl.at(t), ca, cr, cg, cb, wr, wg, wb, ratio, closest);
            done = true;
        }
    }
    if (!done) for (int k = 0; k < K; ++k) closest_step_lab(P, pal, k, ca, cr, cg, cb, wr, wg, wb, ratio, closest);
    if (closest[3] == 2147483647) closest[1] = closest[0];
}

// ---- the Ditherable the quantizer hands to the ditherers ------------------------------------------
// (RGB NQ/PnnQuantizer.java:377-391, LAB NQ/PnnLABQuantizer.java:476-490).  `binCache` (nullable) is the
// reference's nearestMap when it is keyed by histogram bin (REFERENCE_SEQUENTIAL mode only; -1 = empty).
// REFERENCE_SEQUENTIAL + LAB: what getLab() adds to the reference's pixelMap while the gilbert pass runs (its size feeds the
// BlueNoise weight, NQ/PnnLABQuantizer.java:512): the colours handed to nearestColorIndex on a cache miss and the palette entries
// it touches.  One lane runs the chain, so a plain counter is enough.
struct SeqLog { int* colors; int* count; unsigned char* seen; int cap; };
__device__ __forceinline__ void seqlog_add(const SeqLog* l, int c) { const int i = (*l->count)++; if (i < l->cap) l->colors[i] = c; }
struct LookupCtx {
    const SeqLog* slog = nullptr;
    const DevParams* P;
    PalView pal;
    short* binCache;       // [65536] or nullptr (cache-miss semantics)
    long long rng;         // java.util.Random state of this chain
    int dither;
    const CellLists* lists; // candidate lists per colour cell, or nullptr
};

__device__ __forceinline__ int nearest_cached(LookupCtx& cx, int c) {
    const DevParams& P = *cx.P;
    unsigned char* seen = nullptr;
    if (cx.binCache != nullptr && P.binKeyed) {
        const int offset = getColorIndex(c, P.hasSemi != 0, P.hasAlpha != 0);
        short got = cx.binCache[offset];
        if (got >= 0) return got;
        if (cx.slog && P.kind == 1) { seqlog_add(cx.slog, c_alpha(c) <= 0xF ? P.transparentColor : c); seen = cx.slog->seen; }
        int k = nearest_any(P, cx.pal, c, cx.lists, seen);
        cx.binCache[offset] = (short) k;
        return k;
    }
    // a cache keyed by the full colour is transparent: nearest is pure in c (and a repeated colour is already in pixelMap)
    if (cx.slog && P.kind == 1) { seqlog_add(cx.slog, c_alpha(c) <= 0xF ? P.transparentColor : c); seen = cx.slog->seen; }
    return nearest_any(P, cx.pal, c, cx.lists, seen);
}

// RGB closestColorIndex: NQ/PnnQuantizer.java:313-375
__device__ __forceinline__ int closest_rgb(LookupCtx& cx, int c, int pos) {
    const DevParams& P = *cx.P;
    if (c_alpha(c) <= 0xF) return nearest_cached(cx, c);
    int closest[4];
    closest_tuple_rgb(P, cx.pal, c, closest, cx.lists);
    const int MAX_ERR = P.K << 2;
    int idx = (pos + 1) % 2;
    if (closest[3] * .67 < (closest[3] - closest[2])) idx = 0;
    else if (closest[0] > closest[1]) idx = pos % 2;
    if (closest[idx + 2] >= MAX_ERR || (P.hasAlpha && closest[idx] == 0)) return nearest_cached(cx, c);
    return closest[idx];
}
// LAB closestColorIndex: NQ/PnnLABQuantizer.java:407-474
__device__ __forceinline__ int closest_lab(LookupCtx& cx, int c) {
    const DevParams& P = *cx.P;
    if (c_alpha(c) <= 0xF) return nearest_cached(cx, c);
    int closest[4];
    closest_tuple_lab(P, cx.pal, c, closest, cx.lists);
    int idx = 1;
    if (closest[2] == 0 ||
        (jr_next_int_bound(cx.rng, 32767) % (int) ((unsigned) closest[3] + (unsigned) closest[2])) <= closest[3])
        idx = 0;
    const int MAX_ERR = P.K;
    if (closest[idx + 2] >= MAX_ERR || closest[idx] == 0 || c_alpha(cx.pal.argb[closest[idx]]) < c_alpha(c)) {
        return nearest_cached(cx, c);
    }
    return closest[idx];
}
// Ditherable.nearestColorIndex(palette, c, pos)
__device__ __forceinline__ int ditherable_lookup(LookupCtx& cx, int c, int pos) {
    const DevParams& P = *cx.P;
    if (P.kind == 0) {
        if (cx.dither) return nearest_cached(cx, c);
        return closest_rgb(cx, c, pos);
    }
    if (P.K <= 4) return nearest_cached(cx, c);
    return closest_lab(cx, c);
}

// stage the palette (and its Lab, LAB kind) into LDS; `smem` must hold K ints (+3K floats)
__device__ __forceinline__ PalView stage_palette(const DevParams& P, const int* __restrict__ g_palette, void* smem) {
    int* s_argb = (int*) smem;
    float* s_L = (float*) (s_argb + P.K);
    float* s_A = s_L + P.K;
    float* s_B = s_A + P.K;
    for (int i = threadIdx.x; i < P.K; i += blockDim.x) {
        int c2 = g_palette[i];
        s_argb[i] = c2;
        if (P.kind == 1) {
            Lab l2 = RGB2LAB(c2);
            s_L[i] = l2.L; s_A[i] = l2.A; s_B[i] = l2.B;
        }
    }
    __syncthreads();
    PalView v; v.argb = s_argb; v.L = s_L; v.A = s_A; v.B = s_B; v.gamma = g_tab.gamma; v.blue = g_tab.blue;
    return v;
}
__host__ __device__ __forceinline__ size_t palette_smem_bytes(int kind, int K) {
    return (size_t) K * sizeof(int) + (kind == 1 ? (size_t) 3 * K * sizeof(float) : 0);
}
// palette + LDS copies of the gamma and blue-noise tables (the dither chains read them with per-lane indices)
__host__ __device__ __forceinline__ size_t palette_tables_smem_bytes(int kind, int K) {
    return ((palette_smem_bytes(kind, K) + 15) & ~(size_t) 15) + 256 * sizeof(double) + 4096;
}
__device__ __forceinline__ PalView stage_palette_tables(const DevParams& P, const int* __restrict__ g_palette, void* smem) {
    unsigned char* base = (unsigned char*) smem + ((palette_smem_bytes(P.kind, P.K) + 15) & ~(size_t) 15);
    double* s_gamma = (double*) base;
    signed char* s_blue = (signed char*) (base + 256 * sizeof(double));
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_gamma[i] = g_tab.gamma[i];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ((int*) s_blue)[i] = ((const int*) g_tab.blue)[i];
    PalView v = stage_palette(P, g_palette, smem);     // ends with __syncthreads()
    v.gamma = s_gamma; v.blue = s_blue;
    return v;
}

} // namespace nq
