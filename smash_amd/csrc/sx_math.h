// sx_math.h -- device math for the smashx kernels (gfx950).
//
// Parity design (DESIGN.md "Numerics"): the reference is fp32 Fortran calling glibc's libm.  To stay
// within 1e-6 of it over thousands of recurrent steps the kernels must reproduce libm's *rounded
// results*, not just its accuracy class:
//   * glibc 2.35 powf/expf/logf are correctly rounded in 99.94 % of calls (measured), so the
//     kernels compute the few fixed powers the model needs (x^-4, x^-5, y^-1/4, y^-5/4, h^3.5,
//     h^2.5) and exp/log as "fp32 hardware seed + one fp64 Newton step, round once" -- correctly
//     rounded except in ~1e-7 of cases, no table, a handful of fp64 FMAs.
//   * glibc 2.35 tanhf is the fdlibm float algorithm (expm1f based) and is NOT correctly rounded
//     (36 % of results differ from the correctly rounded value), so sx_tanhf restates that published
//     fdlibm algorithm operation by operation; it is bit-identical to glibc on every float in
//     [2^-63, 24] (checked exhaustively on the host build of this header, tests/test_sx_math.py).
// Everything is compiled with -ffp-contract=off; the only fused operations are the explicit fma().
#pragma once

#include <stdint.h>

// SX_EXACT_LIBM=1 (the libsmashx_exact.so build): every power, exponential and logarithm goes through glibc 2.35's own float
// algorithms restated in sx_libm.h, every division is the IEEE division.  Slower; exists to show that the default build's
// distance from the reference is libm rounding and nothing else (tests/test_gpu_exact.py, DESIGN.md "Numerics").
#ifndef SX_EXACT_LIBM
#define SX_EXACT_LIBM 0
#endif
#ifndef SX_EXACT_DIV
#define SX_EXACT_DIV 2
#endif
#ifndef SX_TANH_FAST
#define SX_TANH_FAST 1      // wave-uniform straight-line path of sx_tanhf for small arguments (0: the branchy restatement only)
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define SX_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define SX_HD static inline
#endif

SX_HD uint32_t sx_f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
SX_HD float sx_u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// fast fp32 seeds (~1 ulp); only their fp64-refined results are ever used
SX_HD float sx_seed_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
SX_HD float sx_seed_rsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
SX_HD float sx_seed_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

// Division by a loop-invariant denominator: r = RN(1/d) once (one IEEE division), then
//   q = RN(a*r);  e = a - d*q (exact, fma);  RN(q + e*r)
// which IS the correctly rounded quotient RN(a/d) for normal-range operands (Markstein's theorem;
// 0 mismatches against a/d on 2e8 random pairs, all-ones-significand denominators included:
// tests/test_sx_math.py).  3 instructions instead of the ~10 of the IEEE division expansion.
struct SxDiv { float d, r; };
SX_HD SxDiv sx_mkdiv(float d) { SxDiv D; D.d = d; D.r = 1.0f / d; return D; }
// The same division for a tiny numerator (|a / d| < 2^-100, subnormal a included): the three operations on a * 2^64, scaled back.
// Power-of-two scalings are exact, the residual of the scaled problem does not underflow, so the result is RN(a / d) whenever that
// is a normal number: *ok says so (false: subnormal or zero result -- scaling back rounded a second time; the caller divides).
// Checked against a / d on the host, subnormal numerators and results included (tests/test_sx_math.py).
SX_HD float sx_div_scaled(float a, const SxDiv& D, bool* ok) {
    const float as = a * 0x1p64f;
    const float qs = as * D.r;
    const float q3 = fmaf(fmaf(-D.d, qs, as), D.r, qs) * 0x1p-64f;
    *ok = fabsf(q3) >= 0x1p-126f;
    return q3;
}
SX_HD float sx_div(float a, const SxDiv& D) {
    const float q = a * D.r;
    const float e = fmaf(-D.d, q, a);
    const float q2 = fmaf(e, D.r, q);
#if SX_EXACT_LIBM
    // The theorem needs the residual to be exact: outside the comfortable exponent range (tiny quotients on their way to the
    // subnormals -- the fringe of a decaying adjoint field is full of them -- and huge ones) the exact build takes the IEEE division.
    // SX_EXACT_DIV: 0 = per-lane branch (round 2), 1 = the test is made for the whole wavefront (scalar branch, the IEEE expansion
    // only runs in wavefronts that hold such a quotient), 2 = as 1, and quotients below 2^-100 whose IEEE result is still a normal
    // number are obtained exactly from the same three operations on a * 2^64 (a power-of-two scaling commutes with the division's
    // single rounding as long as the result stays normal): only subnormal, zero, huge and non-finite results take the expansion.
    const float m = fabsf(q2);
#if defined(__HIP_DEVICE_COMPILE__) && SX_EXACT_DIV >= 1
#if SX_EXACT_DIV >= 2
    const bool tiny = !(m > 0x1p-100f) && a != 0.f;
    if (__builtin_amdgcn_ballot_w64(tiny || !(m < 0x1p100f)) == 0ull) return q2;
    bool ok3;
    const float q3 = sx_div_scaled(a, D, &ok3);
    const bool bad = (tiny && !ok3) || (!(m < 0x1p100f) && a != 0.f);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) return bad ? a / D.d : (tiny ? q3 : q2);
    return tiny ? q3 : q2;
#else
    const bool bad = !(m > 0x1p-100f && m < 0x1p100f) && a != 0.f;
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) return bad ? a / D.d : q2;
#endif
#else
    if (!(m > 0x1p-100f && m < 0x1p100f) && a != 0.f) return a / D.d;
#endif
#endif
    return q2;
}

// Four quotients by one loop-invariant denominator (the routing kernels' time blocks).  Default build: four sx_div.  Exact-libm build:
// the three operations for all four, then ONE range test for the batch -- sx_div's wave-uniform guard costs a ballot and a scalar branch
// per quotient, and in a routing super-step (8 quotients forward, 12 reverse) that was a third of the instructions -- and only a
// wavefront that holds an out-of-range quotient goes through the guarded form.  Same results as four sx_div by construction.
SX_HD void sx_div4(float* q, const float* a, const SxDiv& D) {
#if SX_EXACT_LIBM && defined(__HIP_DEVICE_COMPILE__) && SX_EXACT_DIV >= 1
    bool odd = false;
    for (int i = 0; i < 4; ++i) {
        const float q1 = a[i] * D.r;
        q[i] = fmaf(fmaf(-D.d, q1, a[i]), D.r, q1);
        const float m = fabsf(q[i]);
        odd = odd || (!(m > 0x1p-100f && m < 0x1p100f) && a[i] != 0.f);
    }
    if (__builtin_amdgcn_ballot_w64(odd) == 0ull) return;
#endif
    for (int i = 0; i < 4; ++i) q[i] = sx_div(a[i], D);
}

// Division by a denominator that changes every step (1 + hp*tanh, the two quotients inside tanh).  hipcc expands
// a/b into the 11-instruction IEEE sequence (two v_div_scale, v_rcp, four fma, v_div_fmas, v_div_fixup); the
// operands here are always in the normal range, so the same Markstein correction as sx_div works on a reciprocal
// refined by one Newton step from the 1-ulp hardware seed: r = RN(1/b) unless 1/b lies within 2^-46 of a rounding
// boundary, and then q is the correctly rounded quotient.  6 instructions; mismatches against a/b are counted on
// the device by smashx_selftest_math (tests/test_gpu_parity.py: < 1e-6 of calls, 1 ulp).
SX_HD float sx_fdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && !SX_EXACT_LIBM
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
#else
    return a / b;
#endif
}
SX_HD SxDiv sx_mkdiv_fast(float d) {
    SxDiv D; D.d = d;
#if defined(__HIP_DEVICE_COMPILE__) && !SX_EXACT_LIBM
    const float r = __builtin_amdgcn_rcpf(d);
    D.r = fmaf(fmaf(-d, r, 1.0f), r, r);
#else
    D.r = 1.0f / d;
#endif
    return D;
}

SX_HD float sx_nanf() { return sx_u2f(0x7fc00000u); }
SX_HD float sx_inff() { return sx_u2f(0x7f800000u); }
#if SX_EXACT_LIBM
#include "sx_libm.h"
// the model's powers, exponential and logarithm exactly as the reference's compiler emits them: calls of powf / expf / logf
SX_HD float sx_pow_m4(float x) { return sx_g_powf(x, -4.0f); }
SX_HD void sx_pow_m4_m5(float x, float* m4, float* m5) { sx_g_powf2(x, -4.0f, -5.0f, m4, m5); }      // one log2_inline per base
SX_HD float sx_pow_m025(float y) { return sx_g_powf(y, -0.25f); }
SX_HD void sx_pow_m025_m125(float y, float* m025, float* m125) { sx_g_powf2(y, -0.25f, -1.25f, m025, m125); }
SX_HD float sx_pow_3p5(float h) { return sx_g_powf(h, 3.5f); }
SX_HD void sx_pow_3p5_2p5(float h, float* p35, float* p25) { sx_g_powf2(h, 3.5f, 2.5f, p35, p25); }
SX_HD float sx_expf(float x) { return sx_g_expf(x); }
SX_HD float sx_logf(float x) { return sx_g_logf(x); }
SX_HD float sx_powf(float x, float y) { return sx_g_powf(x, y); }
#else
// 1/x in fp64 to ~2^-45 from an fp32 seed: one Newton step  u <- u + u(1 - x u)
SX_HD double sx_rcp_d(float x) {
    const double d = (double)x;
    double u = (double)sx_seed_rcp(x);
    const double e = fma(-d, u, 1.0);
    return fma(u, e, u);
}
// y^(-1/4) in fp64 to ~2^-43: Newton on r^-4 = y:  r <- r + r(1 - y r^4)/4
SX_HD double sx_rquart_d(float y) {
    const double d = (double)y;
    double r = (double)sx_seed_rsq(sx_seed_sqrt(y));
    const double r2 = r * r;
    const double e = fma(-d, r2 * r2, 1.0);
    return fma(r, 0.25 * e, r);
}
// sqrt(h) in fp64 to ~2^-44:  s = h*rs;  s <- s + (h - s^2) * rs/2
SX_HD double sx_sqrt_d(float h) {
    const double d = (double)h;
    const double r = (double)sx_seed_rsq(h);
    double s = d * r;
    const double e = fma(-s, s, d);
    return fma(e, 0.5 * r, s);
}

// powf(x, -4), powf(x, -5)   [x > 0]   (gr_transfer, md_gr_operator.f90:94-106; GR_TRANSFER_B forward_db.f90:6349-6368)
SX_HD float sx_pow_m4(float x) { const double u = sx_rcp_d(x); const double u2 = u * u; return (float)(u2 * u2); }
SX_HD void sx_pow_m4_m5(float x, float* m4, float* m5) {
    const double u = sx_rcp_d(x); const double u2 = u * u; const double u4 = u2 * u2;
    *m4 = (float)u4; *m5 = (float)(u4 * u);
}
// powf(y, -0.25), powf(y, -1.25)   [y > 0]
SX_HD float sx_pow_m025(float y) { return (float)sx_rquart_d(y); }
SX_HD void sx_pow_m025_m125(float y, float* m025, float* m125) {
    const double r = sx_rquart_d(y); const double r2 = r * r;
    *m025 = (float)r; *m125 = (float)((r2 * r2) * r);
}
// powf(h, 3.5), powf(h, 2.5)   [h >= 0]   (gr_exchange md_gr_operator.f90:77; GR_EXCHANGE_B forward_db.f90:6155-6156)
SX_HD float sx_pow_3p5(float h) {
    if (!(h > 0.f)) return 0.f;
    const double d = (double)h; return (float)(((d * d) * d) * sx_sqrt_d(h));
}
SX_HD void sx_pow_3p5_2p5(float h, float* p35, float* p25) {
    if (!(h > 0.f)) { *p35 = 0.f; *p25 = 0.f; return; }
    const double d = (double)h, s = sx_sqrt_d(h), d2 = d * d;
    *p35 = (float)((d2 * d) * s); *p25 = (float)(d2 * s);
}

// expf: fp64 evaluation, one rounding (glibc's float version is correctly rounded in 99.94 %)
SX_HD float sx_expf(float x) { return (float)exp((double)x); }

// logf / powf with run-time arguments (vic-a exponents 1/(b+1), b+1, ...; the logarithmic criterion; the gap branch of
// gr_transfer).  Same contract as above -- an fp64 value within ~2^-45 of the true one, rounded once to fp32, i.e. the
// correctly rounded result except when the true value lies within 2^-45 of a rounding boundary (glibc's own powf / logf
// miss 6e-4 of the time) -- but evaluated by ~45 fp64 FMAs instead of the ~200 instructions of the double-precision
// library pow:  log2 x = e + 2 atanh(s) / ln 2 with x = m 2^e, m in [sqrt(1/2), sqrt 2), s = (m-1)/(m+1);
// x^y = 2^(y log2 x) with 2^t = 2^n exp(r ln 2), r = t - n in [-1/2, 1/2].   tests/test_sx_math.py checks both against the
// double-precision library on a few million arguments (|error| < 2^-44 relative) and the fp32 results against glibc's.
struct SxLog2 { double l2; int special; };   // special: 0 = finite x > 0, 1 = x == 0, 2 = x < 0 or NaN, 3 = +inf
SX_HD SxLog2 sx_log2_d(float x, const bool fast = true) {
    SxLog2 R; R.special = 0; R.l2 = 0.0;
    uint32_t u = sx_f2u(x);
    int e = 0;
#if defined(__HIP_DEVICE_COMPILE__) && SX_TANH_FAST
    // every lane a positive normal number (always, for the bases the model raises): no branch at all -- the same operations
    if (!fast || __builtin_amdgcn_ballot_w64(!(u - 0x00800000u < 0x7f000000u)) != 0ull)
#endif
    {
        if (!(x > 0.f)) { R.special = (x == 0.f) ? 1 : 2; return R; }
        if (u >= 0x7f800000u) { R.special = 3; return R; }
        if (u < 0x00800000u) { u = sx_f2u(x * 16777216.0f); e = -24; }      // subnormal
    }
    e += (int)(u >> 23) - 127;
    float m = sx_u2f((u & 0x007fffffu) | 0x3f800000u);                  // [1, 2)
    { const bool up = m > 1.41421356f; m = up ? m * 0.5f : m; e = up ? e + 1 : e; }   // [sqrt(1/2), sqrt 2)
    const double f = (double)m - 1.0;                                    // exact
    const double d = 2.0 + f;
    double r = (double)sx_seed_rcp((float)d);
    double t = fma(-d, r, 1.0); r = fma(r, t, r);
    t = fma(-d, r, 1.0); r = fma(r, t, r);
    const double s2 = f * r, z = s2 * s2;                                // s = f / (2 + f), |s| <= 0.1716
    double p = 1.0 / 17.0;
    p = fma(p, z, 1.0 / 15.0); p = fma(p, z, 1.0 / 13.0); p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0); p = fma(p, z, 1.0 / 5.0); p = fma(p, z, 1.0 / 3.0);
    const double lm = fma(s2 * z, p, s2);                                // atanh(s) = s + s^3 (1/3 + s^2/5 + ...)
    R.l2 = fma(lm, 2.8853900817779268, (double)e);                      // 2 / ln 2
    return R;
}
// 2^t for |t| < 1100 (beyond: the ldexp saturates to 0 / inf)
SX_HD double sx_exp2_d(double t) {
    t = !(t > -1100.0) ? -1100.0 : t;
    t = !(t < 1100.0) ? 1100.0 : t;
    const double n = rint(t);
    const double u = (t - n) * 0.6931471805599453;                        // r ln 2, |u| <= 0.3466
    double p = 2.505210838544172e-08;                                     // 1/11!
    p = fma(p, u, 2.755731922398589e-07); p = fma(p, u, 2.755731922398589e-06); p = fma(p, u, 2.48015873015873e-05);
    p = fma(p, u, 1.984126984126984e-04); p = fma(p, u, 1.388888888888889e-03); p = fma(p, u, 8.333333333333333e-03);
    p = fma(p, u, 4.166666666666666e-02); p = fma(p, u, 1.666666666666667e-01); p = fma(p, u, 0.5);
    p = fma(p, u, 1.0); p = fma(p, u, 1.0);
    return ldexp(p, (int)n);
}
SX_HD float sx_logf(float x) {
    const SxLog2 L = sx_log2_d(x);
    if (L.special) return L.special == 1 ? -sx_inff() : L.special == 3 ? sx_inff() : sx_nanf();
    return (float)(L.l2 * 0.6931471805599453);
}
// x^y given log2 x (several powers of one base share the logarithm: the vic-a adjoints need x^y, x^(y-1) and ln x)
SX_HD float sx_pow_from(const SxLog2& L, float x, float y, const bool fast = true) {
#if defined(__HIP_DEVICE_COMPILE__) && SX_TANH_FAST
    // the ordinary case on every lane of the wavefront: straight to the exponential
    if (fast && __builtin_amdgcn_ballot_w64(L.special != 0 || y == 0.f || x == 1.f) == 0ull) return (float)sx_exp2_d((double)y * L.l2);
#endif
    if (y == 0.f || x == 1.f) return 1.f;
    if (L.special == 1) return y > 0.f ? 0.f : sx_inff();
    if (L.special == 3) return y > 0.f ? sx_inff() : 0.f;
    if (L.special == 2) return sx_nanf();         // negative base: the model never raises one (the reference would give NaN or +-|x|^y)
    return (float)sx_exp2_d((double)y * L.l2);
}
SX_HD float sx_powf(float x, float y) { return sx_pow_from(sx_log2_d(x), x, y); }

#endif   // SX_EXACT_LIBM
#ifndef SX_LIBM_INIT
#define SX_LIBM_INIT()       // default build: nothing to stage (the exact-libm build copies its tables into LDS here, sx_libm.h)
#endif

// Several powers and the logarithm of ONE base (vic-a: x**y, x**(y-1) and ln x of the same x in the adjoint of the infiltration
// curve, forward_db.f90:6808-6990).  Default build: the base-2 logarithm is evaluated once -- sx_powb / sx_logb return exactly what
// sx_powf / sx_logf return, by construction.  Exact-libm build: every call is the C library's own algorithm, nothing is shared.
struct SxPowBase {
    float x;
#if !SX_EXACT_LIBM
    SxLog2 L;
#endif
};
SX_HD SxPowBase sx_powbase(float x) {
    SxPowBase B; B.x = x;
#if !SX_EXACT_LIBM
    B.L = sx_log2_d(x);
#endif
    return B;
}
SX_HD SxPowBase sx_powbase_one() {       // the base 1 (log2 = 0) without evaluating anything: a placeholder for a base that is not raised
    SxPowBase B; B.x = 1.f;
#if !SX_EXACT_LIBM
    B.L.l2 = 0.0; B.L.special = 0;
#endif
    return B;
}
SX_HD float sx_powb(const SxPowBase& B, float y) {
#if SX_EXACT_LIBM
    return sx_powf(B.x, y);
#else
    return sx_pow_from(B.L, B.x, y);
#endif
}
SX_HD float sx_logb(const SxPowBase& B) {
#if SX_EXACT_LIBM
    return sx_logf(B.x);
#else
    if (B.L.special) return B.L.special == 1 ? -sx_inff() : B.L.special == 3 ? sx_inff() : sx_nanf();
    return (float)(B.L.l2 * 0.6931471805599453);
#endif
}

// ---- fdlibm float expm1 / tanh (Sun Microsystems 1993, public algorithm; the float port is what
// ---- glibc 2.35 ships as expm1f/tanhf).  Restated for arguments the model can produce:
// ---- expm1 for x <= 44 (tanh passes -2|x| in [-2,0) or 2|x| in [2,44)).
SX_HD float sx_expm1f(float x) {
    const float one = 1.0f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
    const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
                Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
    float y, hi, lo, c = 0.f, t, e, hxs, hfx, r1;
    int32_t k;
    uint32_t hx = sx_f2u(x);
    const uint32_t xsb = hx & 0x80000000u;
    hx &= 0x7fffffffu;
    if (hx >= 0x4195b844u) {            // |x| >= 27 ln2
        if (xsb) return -1.0f;          // fdlibm: tiny - one
        return sx_expf(x) - one;        // never reached by tanh for |x| < 22 (2|x| < 44 < 88.7): fdlibm falls through
    }
    if (hx > 0x3eb17218u) {             // |x| > 0.5 ln2
        if (hx < 0x3F851592u) {         // |x| < 1.5 ln2
            if (!xsb) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else      { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5f : 0.5f));
            t = (float)k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000u) {      // |x| < 2^-25
        return x;
    } else {
        k = 0;
    }
    hfx = 0.5f * x;
    hxs = x * hfx;
    r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    t = 3.0f - r1 * hfx;
    e = hxs * sx_fdiv(r1 - t, 6.0f - x * t);
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) {
        if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
        return one + 2.0f * (x - e);
    }
    if (k <= -2 || k > 56) {
        y = one - (e - x);
        y = sx_u2f(sx_f2u(y) + ((uint32_t)k << 23));
        return y - one;
    }
    if (k < 23) {
        t = sx_u2f(0x3f800000u - (0x1000000u >> k));
        y = t - (e - x);
        y = sx_u2f(sx_f2u(y) + ((uint32_t)k << 23));
    } else {
        t = sx_u2f((uint32_t)(0x7f - k) << 23);
        y = x - (e + t);
        y += one;
        y = sx_u2f(sx_f2u(y) + ((uint32_t)k << 23));
    }
    return y;
}

SX_HD float sx_tanhf(float x, const bool fast = true) {      // fast = false: the branchy restatement only (device self-test)
    const uint32_t jx = sx_f2u(x), ix = jx & 0x7fffffffu;
    float t, z;
#if defined(__HIP_DEVICE_COMPILE__) && SX_TANH_FAST
    // Small arguments on every lane of the wavefront (evaporation or net rain against the store's capacity: en / cp ~ 1e-3): the same
    // operations as below along the one path such arguments take -- |x| < 2^-55: x (1 + x); else expm1f(-2|x|) with k = 0 (|2x| <
    // ln2 / 2), its |2x| < 2^-25 shortcut as a select -- without any branch.  Same bits by construction.
    if (fast && __builtin_amdgcn_ballot_w64(!(ix < 0x3e317218u)) == 0ull) {      // |x| < ln2 / 4: |2x| <= 0x3eb17218 (ln2 / 2), expm1f's k = 0 range
        const float ax = sx_u2f(ix);
        const float x2 = -2.0f * ax;
        const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f, Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
        const float hfx = 0.5f * x2, hxs = x2 * hfx;
        const float r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
        const float tt = 3.0f - r1 * hfx;
        const float e = hxs * sx_fdiv(r1 - tt, 6.0f - x2 * tt);
        float em = x2 - (x2 * e - hxs);
        em = ((sx_f2u(x2) & 0x7fffffffu) < 0x33000000u) ? x2 : em;           // expm1f: |x| < 2^-25 returns x
        z = sx_fdiv(-em, em + 2.0f);
        float res = (jx >> 31) ? -z : z;
        res = (ix < 0x24000000u) ? x * (1.0f + x) : res;                      // tanhf: |x| < 2^-55
        return ix == 0u ? x : res;
    }
#endif
    if (ix < 0x41b00000u) {             // |x| < 22
        if (ix == 0) return x;
        if (ix < 0x24000000u) return x * (1.0f + x);
        const float ax = sx_u2f(ix);
        if (ix >= 0x3f800000u) { t = sx_expm1f(2.0f * ax); z = 1.0f - sx_fdiv(2.0f, t + 2.0f); }
        else                   { t = sx_expm1f(-2.0f * ax); z = sx_fdiv(-t, t + 2.0f); }
    } else {
        z = 1.0f - 1e-30f;
    }
    return (jx >> 31) ? -z : z;
}
