// fp64 sincos sized for what the flattener needs: a result that, rounded ONCE to f32, equals the correctly
// rounded f32 value except when the exact value lies within ~2^-55 (relative) of an f32 rounding boundary
// (~2^-28 of all arguments).  That is the same contract as "ocml fp64 value rounded once" (common.h, numeric
// rules) and as the CPU oracle's libm call, at 40 % of ocml's instruction count: ocml's sincos is ~190
// instructions because it carries Payne-Hanek reduction for arguments the flattener never produces; those
// arguments still go to ocml (cold branch in common.h).  (The same exercise for atan2 -- breakpoint reduction,
// one division, degree-11 polynomial -- came to 113 instructions against ocml's 122, so atan2 stays with ocml.)
//
// Every step is an IEEE fp64 operation spelled explicitly (fma where written, no contraction elsewhere), so
// a host (g++) build of this header computes the same bits as gfx950; tests/test_fp64_math.py does exactly that.
//
// The polynomials are the classical minimax sets for sin/cos on [-pi/4, pi/4] (published with Sun's fdlibm,
// 1993); tests/test_fp64_math.py measures them against libm.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace vk {
namespace f64 {

// sin(r), cos(r) for |r| <= pi/4 (a little beyond is harmless), |error| < 2^-57.
__device__ __forceinline__ void sincos_reduced(double r, double &s, double &c) {
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    s = fma(r * z, ps, r);
    c = fma(z * z, pc, fma(z, -0.5, 1.0));
}

constexpr double SINCOS_MAX_ARG = 512.0;

// sin(x) and cos(x) for |x| <= SINCOS_MAX_ARG; the caller routes larger / non-finite arguments elsewhere.
// Three-constant Cody-Waite reduction: k * PIO2_1 and k * PIO2_2 are exact for |k| < 2^19 (33-bit constants),
// so r carries the rounding of the last fma only.
__device__ __forceinline__ void sincos_medium(double x, double &s, double &c) {
    constexpr double TWO_OVER_PI = 6.36619772367581382433e-01;
    constexpr double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    constexpr double PIO2_2 = 6.07710050630396597660e-11;   // next 33 bits
    constexpr double PIO2_2T = 2.02226624879595063154e-21;  // pi/2 - (PIO2_1 + PIO2_2)
    const double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, PIO2_1, x);
    r = fma(-k, PIO2_2, r);
    r = fma(-k, PIO2_2T, r);
    double sr, cr;
    sincos_reduced(r, sr, cr);
    const int q = (int)k;
    const double sa = (q & 1) ? cr : sr;
    const double ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

// x^y for a finite x > 0 that came from an f32 (24 significant bits, any f32 exponent incl. denormals) and an f32-valued
// exponent with |y| <= 8: what the flattener's inverse-integral needs (|u|^(2/3) with the f32 constant 2/3) under the same
// contract as above -- the fp64 value, rounded ONCE to f32 by the caller, differs from libm's only when the exact value
// lies within ~2^-55 (relative) of an f32 rounding boundary.  ocml's pow is 253 instructions (double-double log and exp
// for any argument); this is ~115:
//   x = 2^e m, m in [sqrt(1/2), sqrt(2));  z = (m - 1) / (m + 1) as z_hi + z_lo (the residual by ONE exact fma: m - 1 and
//   m + 1 are exact for a 24-bit m);  ln m = 2 z_hi + [2 z_lo + z^3 P(z^2)], P the atanh series to z^22 (|z| <= 0.1716:
//   the bracket is < 0.006, so plain fp64 leaves it 2^-60 absolute);  log2 m as a double-double product with log2(e);
//   E = y (e + log2 m) with the rounding of the sum recovered (two-sum);  2^E = 2^k exp(f ln 2), |f| <= 1/2, degree-13.
__device__ inline __attribute__((noinline)) double pow_pos(double x, double y) {
    // exponent and mantissa (x is a normal double: an f32 denormal is one too)
    long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    if (m >= 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double num = m - 1.0, den = m + 1.0;
    const double rcp = 1.0 / den;
    const double z_hi = num * rcp;
    const double z_lo = fma(-z_hi, den, num) * rcp;
    const double w = z_hi * z_hi;
    double P = fma(w, 2.0 / 23.0, 2.0 / 21.0);
    P = fma(w, P, 2.0 / 19.0);
    P = fma(w, P, 2.0 / 17.0);
    P = fma(w, P, 2.0 / 15.0);
    P = fma(w, P, 2.0 / 13.0);
    P = fma(w, P, 2.0 / 11.0);
    P = fma(w, P, 2.0 / 9.0);
    P = fma(w, P, 2.0 / 7.0);
    P = fma(w, P, 2.0 / 5.0);
    P = fma(w, P, 2.0 / 3.0);
    const double H = 2.0 * z_hi;                          // ln m = H + T
    const double T = fma(z_hi * w, P, 2.0 * z_lo);
    constexpr double L2E_HI = 1.4426950408889634074, L2E_LO = 2.0355273740931033e-17;  // log2(e) = hi + lo
    const double p_hi = H * L2E_HI;
    const double p_lo = fma(H, L2E_HI, -p_hi) + fma(H, L2E_LO, T * L2E_HI);
    // E = y * (e + p_hi + p_lo): y * e is exact (24 x 11 bits)
    const double A = y * (double)e;
    const double B_hi = y * p_hi;
    const double B_lo = fma(y, p_hi, -B_hi) + y * p_lo;
    const double S = A + B_hi;
    const double bb = S - A;
    const double S_err = (A - (S - bb)) + (B_hi - bb);    // two-sum
    const double E_lo = S_err + B_lo;
    const double k = rint(S);
    const double f = (S - k) + E_lo;                      // |f| <= 1/2 (+ a hair)
    const double g = f * 6.93147180559945286227e-01;
    double q = fma(g, 1.6059043836821613e-10, 2.08767569878681e-09);   // 1/13!, 1/12!
    q = fma(g, q, 2.505210838544172e-08);
    q = fma(g, q, 2.755731922398589e-07);
    q = fma(g, q, 2.7557319223985893e-06);
    q = fma(g, q, 2.48015873015873e-05);
    q = fma(g, q, 1.984126984126984e-04);
    q = fma(g, q, 1.388888888888889e-03);
    q = fma(g, q, 8.333333333333333e-03);
    q = fma(g, q, 4.1666666666666664e-02);
    q = fma(g, q, 1.6666666666666666e-01);
    q = fma(g, q, 0.5);
    const double r = fma(g * g, q, g) + 1.0;              // exp(g) = 1 + g + g^2 q
    return ldexp(r, (int)k);
}

}  // namespace f64
}  // namespace vk
