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

}  // namespace f64
}  // namespace vk
