// TEST-ONLY: host build of vello_amd/csrc/engine/fp64_math.h (through the SIMT emulator's hip_runtime.h shim)
// swept against libm.  Prints: n  sin_mismatch  cos_mismatch  max_ulp_sin  max_ulp_cos, where a mismatch is a
// differing f32 after the single final rounding and the ulp figures are fp64 ulps of the libm value.
#include "fp64_math.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint64_t state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }
static inline float urand(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 40) * (1.0 / 16777216.0)); }
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static double ulps(double got, double want) { return got == want ? 0.0 : fabs(got - want) / (fabs(want) * 0x1p-52 + 1e-300); }

// pow mode: n  mismatches  max_ulp  -- vk::f64::pow_pos against libm's pow, both rounded once to f32
static int pow_mode(long n) {
    const float ys[] = {2.0f / 3.0f, 2.0f / 3.0f, 2.0f / 3.0f, 0.5f, 1.5f, 1.0f / 3.0f, 2.0f, 7.3f, -2.0f / 3.0f, -8.0f};
    long mis = 0;
    double worst = 0;
    for (long i = 0; i < n; i++) {
        float x;
        switch (i & 3) {
            case 0: x = expf(urand(-14.0f, 7.0f)); break;       // the inverse integral's arguments, log-uniform 1e-6 ... 1e3
            case 1: x = urand(0.5f, 2.0f); break;               // around 1 (cancellation in the logarithm)
            case 2: x = urand(0.0f, 100.0f); break;
            default: x = from_bits((uint32_t)rnd() & 0x7fffffffu);  // any positive f32 incl. denormals
                     if (!(x > 0.0f && x < INFINITY)) x = 1.0f;
        }
        if (!(x > 0.0f)) x = 1e-30f;
        const float y = ys[(i >> 2) % (sizeof ys / sizeof ys[0])];
        const double got = vk::f64::pow_pos((double)x, (double)y), want = pow((double)x, (double)y);
        const float fg = (float)got, fw = (float)want;
        mis += memcmp(&fg, &fw, 4) != 0;
        if (want > 1e-300 && want < 1e300) worst = fmax(worst, ulps(got, want));
    }
    const float special[] = {1.0f, 1e-45f, 1.1754944e-38f, 3.4028235e38f, 0.70710677f, 0.70710683f, 1.4142135f, 1.4142137f, 8.0f, 0.125f};
    for (float x : special)
        for (float y : ys) {
            const float fg = (float)vk::f64::pow_pos((double)x, (double)y), fw = (float)pow((double)x, (double)y);
            mis += memcmp(&fg, &fw, 4) != 0;
        }
    printf("%ld %ld %.4f\n", n, mis, worst);
    return 0;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    if (argc > 2 && !strcmp(argv[2], "pow")) return pow_mode(n);
    long mis_s = 0, mis_c = 0;
    double worst_s = 0, worst_c = 0;
    for (long i = 0; i < n; i++) {
        float x;
        switch (i & 3) {
            case 0: x = urand(-3.2f, 3.2f); break;             // the angles flatten produces
            case 1: x = urand(-512.0f, 512.0f); break;         // the whole supported range
            case 2: x = urand(-0.1f, 0.1f); break;             // nearly straight segments
            default: x = from_bits((uint32_t)rnd());           // any exponent, incl. denormals
                     if (!(fabsf(x) <= (float)vk::f64::SINCOS_MAX_ARG)) x = urand(-8.0f, 8.0f);
        }
        double s, c;
        vk::f64::sincos_medium((double)x, s, c);
        const double ws = sin((double)x), wc = cos((double)x);
        mis_s += (float)s != (float)ws;
        mis_c += (float)c != (float)wc;
        worst_s = fmax(worst_s, ulps(s, ws));
        worst_c = fmax(worst_c, ulps(c, wc));
    }
    // exact multiples of pi/2 as f32, signed zeros, range ends
    const float special[] = {0.0f, -0.0f, 1.5707964f, -1.5707964f, 3.1415927f, -3.1415927f, 6.2831855f, 512.0f, -512.0f, 1e-45f, 1e-38f};
    for (float x : special) {
        double s, c;
        vk::f64::sincos_medium((double)x, s, c);
        const float fs = (float)s, fc = (float)c, ws = (float)sin((double)x), wc = (float)cos((double)x);
        mis_s += memcmp(&fs, &ws, 4) != 0;
        mis_c += memcmp(&fc, &wc, 4) != 0;
    }
    printf("%ld %ld %ld %.4f %.4f\n", n, mis_s, mis_c, worst_s, worst_c);
    return 0;
}
