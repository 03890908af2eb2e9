/* Host build of smash_amd/csrc/sx_math.h for tests/test_sx_math.py (g++ -O2 -ffp-contract=off -mfma). */
#include <stdint.h>
#include <stdlib.h>
#include "../../smash_amd/csrc/sx_math.h"
#include "../../smash_amd/csrc/sx_libm.h"

extern "C" {

/* number of floats in [lo_bits, hi_bits) on which sx_tanhf differs from glibc tanhf */
long sxt_tanh_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    long bad = 0;
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
        float x = sx_u2f((uint32_t)u);
        if (sx_f2u(tanhf(x)) != sx_f2u(sx_tanhf(x))) bad++;
    }
    return bad;
}

static float rnd(float lo, float hi) { return lo + (hi - lo) * (float)(rand() / (double)RAND_MAX); }

/* mismatch counts of the pow helpers against glibc powf on n random arguments from the model's ranges */
void sxt_pow_mismatches(long n, long* out) {
    srand(12345);
    for (int i = 0; i < 6; ++i) out[i] = 0;
    for (long i = 0; i < n; ++i) {
        float x = rnd(3.f, 900.f), m4, m5, y, r, r5, h = rnd(1e-6f, 1.f), p35, p25;
        sx_pow_m4_m5(x, &m4, &m5);
        if (m4 != powf(x, -4.f)) out[0]++;
        if (m5 != powf(x, -5.f)) out[1]++;
        y = powf(x, -4.f) * rnd(1.f, 31.f);
        sx_pow_m025_m125(y, &r, &r5);
        if (r != powf(y, -0.25f)) out[2]++;
        if (r5 != powf(y, -1.25f)) out[3]++;
        sx_pow_3p5_2p5(h, &p35, &p25);
        if (p35 != powf(h, 3.5f)) out[4]++;
        if (p25 != powf(h, 2.5f)) out[5]++;
    }
}

/* mismatches of the reciprocal + 2 FMA division against a / d */
long sxt_div_mismatches(long n) {
    srand(777);
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        uint32_t ua = ((uint32_t)rand() << 8) ^ (uint32_t)rand(), ub = ((uint32_t)rand() << 8) ^ (uint32_t)rand();
        ua = (ua & 0x807fffffu) | ((uint32_t)(100 + rand() % 56) << 23);
        ub = (ub & 0x007fffffu) | ((uint32_t)(100 + rand() % 56) << 23);
        if ((i & 255) == 0) ub |= 0x007fffffu;   /* all-ones significand: the corner of Markstein's theorem */
        float a = sx_u2f(ua), d = sx_u2f(ub);
        SxDiv D = sx_mkdiv(d);
        if (sx_div(a, D) != a / d) bad++;
    }
    return bad;
}

/* the scaled form of that division (exact-libm build, tiny quotients): numerators from the smallest subnormal up, quotients between
 * 2^-160 and 2^-95.  out[0] = results flagged ok that differ from a / d (must be 0), out[1] = flagged ok, out[2] = not ok although
 * a / d is a normal number (must be 0: only subnormal and zero results may be handed back) */
void sxt_div_scaled_check(long n, long* out) {
    srand(991);
    out[0] = out[1] = out[2] = 0;
    for (long i = 0; i < n; ++i) {
        uint32_t ua = ((uint32_t)rand() << 8) ^ (uint32_t)rand(), ub = ((uint32_t)rand() << 8) ^ (uint32_t)rand();
        const int ed = 90 + rand() % 76;                         /* d in 2^-37 .. 2^38 */
        const int eq = 127 - 160 + rand() % 66;                  /* target quotient exponent -160 .. -95 */
        int ea = eq + ed - 127;                                  /* biased exponent of a (may be <= 0: subnormal a) */
        ub = (ub & 0x007fffffu) | ((uint32_t)ed << 23);
        if ((i & 255) == 0) ub |= 0x007fffffu;
        float a;
        if (ea >= 1) a = sx_u2f((ua & 0x807fffffu) | ((uint32_t)ea << 23));
        else { const int sh = 1 - ea; a = sh < 24 ? sx_u2f((ua & 0x80000000u) | (((ua & 0x007fffffu) | 0x00800000u) >> sh)) : sx_u2f((ua & 0x80000000u) | 1u); }
        if (a == 0.f) continue;
        const float d = sx_u2f(ub);
        const SxDiv D = sx_mkdiv(d);
        bool ok;
        const float q = sx_div_scaled(a, D, &ok), ref = a / d;
        if (ok) { out[1]++; if (q != ref) out[0]++; }
        else if (fabsf(ref) >= 0x1p-126f) out[2]++;
    }
}

float sxt_expf(float x) { return sx_expf(x); }

/* sx_powf / sx_logf (fp64 log2 / exp2 evaluation) on n random arguments of the vic-a kind: out[0] = fp32 mismatches against
 * glibc powf, out[1] = against the correctly rounded logarithm (float)log((double)x), dout[0] / dout[1] = largest relative error of the fp64 values against the
 * double-precision library pow / log */
void sxt_powlog_check(long n, long* out, double* dout) {
    srand(4242);
    out[0] = out[1] = 0; dout[0] = dout[1] = 0.0;
    for (long i = 0; i < n; ++i) {
        float x, y;
        switch (i & 3) {
            case 0: x = rnd(1e-7f, 1.f); y = rnd(0.05f, 3.f); break;         /* 1 - w/c in (0,1), exponents 1/(b+1), b+1, b */
            case 1: x = rnd(1e-3f, 50.f); y = rnd(-5.f, 5.f); break;
            case 2: x = sx_u2f(0x3f800000u + (uint32_t)(rand() % 2000000) - 1000000u); y = rnd(-3.f, 3.f); break;   /* x ~ 1 */
            default: x = sx_u2f((uint32_t)rand() % 0x7f000000u + 0x00100000u); y = rnd(-1.5f, 1.5f); break;        /* any magnitude */
        }
        const SxLog2 L = sx_log2_d(x);
        const double pd = sx_exp2_d((double)y * L.l2), pr = pow((double)x, (double)y);
        if (pr > 1e-300 && pr < 1e300) { const double e = fabs(pd - pr) / pr; if (e > dout[0]) dout[0] = e; }
        const double ld = L.l2 * 0.6931471805599453, lr = log((double)x);
        if (fabs(lr) > 1e-300) { const double e = fabs(ld - lr) / fabs(lr); if (e > dout[1]) dout[1] = e; }
        if (sx_f2u(sx_powf(x, y)) != sx_f2u(powf(x, y))) out[0]++;
        if (sx_f2u(sx_logf(x)) != sx_f2u((float)log((double)x))) out[1]++;   /* glibc's logf itself is ~1.7 % away from correct rounding */
    }
}
/* special values: returns the number of disagreements with glibc */
long sxt_pow_specials(void) {
    const float xs[] = {0.f, 1.f, 2.f, 1e-45f, 1e-40f, 3.4e38f, 0.5f}, ys[] = {0.f, 1.f, -1.f, 0.5f, 2.f, -2.5f, 30.f, -30.f};
    long bad = 0;
    for (float x : xs) for (float y : ys) {
        const float a = sx_powf(x, y), b = powf(x, y);
        if (!(a == b || (a != a && b != b))) bad++;
    }
    if (sx_logf(0.f) != logf(0.f)) bad++;
    if (sx_logf(-1.f) == sx_logf(-1.f)) bad++;
    return bad;
}

/* ---- exact-libm build (sx_libm.h): glibc's expf / logf / powf restated; counts of results whose BITS differ from the C library's ---- */
long sxt_g_expf_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    long bad = 0;
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
        const float x = sx_u2f((uint32_t)u), a = sx_g_expf(x), b = expf(x);
        if (sx_f2u(a) != sx_f2u(b) && !(a != a && b != b)) bad++;
    }
    return bad;
}
long sxt_g_logf_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    long bad = 0;
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
        const float x = sx_u2f((uint32_t)u), a = sx_g_logf(x), b = logf(x);
        if (sx_f2u(a) != sx_f2u(b) && !(a != a && b != b)) bad++;
    }
    return bad;
}
/* powf: the six fixed exponents of the GR operators on every float of a base range, then n random (x, y) pairs of the vic-a kind
 * and of any magnitude, then the special values */
long sxt_g_powf_fixed_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    const float ys[6] = {-4.f, -5.f, -0.25f, -1.25f, 3.5f, 2.5f};
    long bad = 0;
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
        const float x = sx_u2f((uint32_t)u);
        for (int j = 0; j < 6; ++j) if (sx_f2u(sx_g_powf(x, ys[j])) != sx_f2u(powf(x, ys[j]))) bad++;
    }
    return bad;
}
long sxt_g_powf_random_mismatches(long n, unsigned seed) {
    srand(seed);
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        float x, y;
        switch (i & 3) {
            case 0: x = rnd(1e-7f, 1.f); y = rnd(0.05f, 3.f); break;
            case 1: x = rnd(1e-3f, 50.f); y = rnd(-5.f, 5.f); break;
            case 2: x = sx_u2f(0x3f800000u + (uint32_t)(rand() % 2000000) - 1000000u); y = rnd(-3.f, 3.f); break;
            default: x = sx_u2f((((uint32_t)rand() << 16) ^ (uint32_t)rand()) & 0x7fffffffu); y = sx_u2f(((uint32_t)rand() << 16) ^ (uint32_t)rand()); break;
        }
        const float a = sx_g_powf(x, y), b = powf(x, y);
        if (sx_f2u(a) != sx_f2u(b) && !(a != a && b != b)) bad++;
    }
    return bad;
}
/* sx_g_powf2 (one log2_inline for two exponents of a base) against two separate library calls */
long sxt_g_powf2_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    const float ys[3][2] = {{-4.f, -5.f}, {-0.25f, -1.25f}, {3.5f, 2.5f}};
    long bad = 0;
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
        const float x = sx_u2f((uint32_t)u);
        for (int j = 0; j < 3; ++j) {
            float a, b;
            sx_g_powf2(x, ys[j][0], ys[j][1], &a, &b);
            if (sx_f2u(a) != sx_f2u(powf(x, ys[j][0])) || sx_f2u(b) != sx_f2u(powf(x, ys[j][1]))) bad++;
        }
    }
    return bad;
}
long sxt_g_specials(void) {
    const float inf = sx_inff(), nan = sx_nanf();
    const float xs[] = {0.f, -0.f, 1.f, -1.f, 2.f, -2.f, 1e-45f, 1e-40f, -1e-40f, 3.4e38f, 0.5f, -0.5f, inf, -inf, nan};
    const float ys[] = {0.f, -0.f, 1.f, -1.f, 0.5f, 2.f, 3.f, -3.f, -2.5f, 30.f, -30.f, 1e30f, -1e30f, inf, -inf, nan};
    long bad = 0;
    for (float x : xs) for (float y : ys) {
        const float a = sx_g_powf(x, y), b = powf(x, y);
        if (!(sx_f2u(a) == sx_f2u(b) || (a != a && b != b))) bad++;
    }
    for (float x : xs) {
        const float a = sx_g_logf(x), b = logf(x), c = sx_g_expf(x), d = expf(x);
        if (!(sx_f2u(a) == sx_f2u(b) || (a != a && b != b))) bad++;
        if (!(sx_f2u(c) == sx_f2u(d) || (c != c && d != d))) bad++;
    }
    return bad;
}

}  /* extern "C" */
