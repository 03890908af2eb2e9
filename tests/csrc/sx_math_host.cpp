/* Host build of smash_amd/csrc/sx_math.h for tests/test_sx_math.py (g++ -O2 -ffp-contract=off -mfma). */
#include <stdint.h>
#include <stdlib.h>
#include "../../smash_amd/csrc/sx_math.h"

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

float sxt_expf(float x) { return sx_expf(x); }

}  /* extern "C" */
