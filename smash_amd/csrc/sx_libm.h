// sx_libm.h -- glibc 2.35's expf / logf / powf restated operation by operation (exact-libm build, -DSX_EXACT_LIBM=1).
//
// Why.  The reference is fp32 Fortran calling the C library (md_gr_operator.f90:63,77,96,106: tanh, **; md_routing_operator.f90:75:
// exp; mwd_cost.f90: log; md_vic_operator.f90: ** with run-time exponents).  The default build of the kernels evaluates those powers
// as "fp32 hardware seed + fp64 Newton step, one rounding" -- the correctly rounded result, which glibc's own float functions miss
// in ~6e-4 of calls.  To show that this is the ONLY difference between the HIP path and the reference, the exact-libm build
// evaluates the very algorithms glibc 2.35 ships (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_powf.c: the table-driven
// double-precision kernels of ARM's optimized-routines by Szabolcs Nagy, MIT licence), with its constants, and IEEE division
// everywhere; tanhf is the fdlibm algorithm already restated in sx_math.h.  The forward sweep is then bit-identical to the
// reference on every golden vector and the gradients agree to the last bit or to the reference's own summation noise
// (tests/test_gpu_exact.py).  The tables below are the constants of that published algorithm (2^(i/32) as IEEE bit patterns
// minus the exponent offset; 1/c and log(c) or log2(c) at 16 centres c of [sqrt(1/2), sqrt 2)); tests/test_sx_math.py checks the
// host build of this header bit for bit against the C library of the build container on ~10^9 arguments.
//
// The double-precision evaluation order follows the C sources; the library itself runs an FMA-contracted build of them on
// x86-64 CPUs with FMA (sysdeps/x86_64/fpu/multiarch), i.e. its fp64 intermediate can differ from an uncontracted evaluation by
// one fp64 ulp -- visible after the final rounding to float in ~1e-9 of calls.  SX_LIBM_FMA (default 1) writes the contractions
// GCC performs there as explicit fma() calls; either way the fp32 results are the library's in all but that 1e-9.
#pragma once

#include "sx_math.h"

#ifndef SX_LIBM_FMA
#define SX_LIBM_FMA 1
#endif

SX_HD double sx_lm_fma(double a, double b, double c) {   // a*b + c the way the library's build evaluates it
#if SX_LIBM_FMA
    return fma(a, b, c);
#else
    return a * b + c;
#endif
}
SX_HD uint64_t sx_d2u(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u; memcpy(&u, &d, 8); return u;
#endif
}
SX_HD double sx_u2d(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; memcpy(&d, &u, 8); return d;
#endif
}

// Where the tables live.  Host builds and the default device build: function-local constant arrays.  Exact-libm device build:
// one copy in LDS per workgroup (SX_LIBM_INIT() at the top of every kernel that can reach these functions) -- a look-up from
// global memory sits on vmcnt, i.e. in the queue of the forcing / tape prefetch of the time loops, and each one made the step
// wait for the rows of the NEXT step (vert_adj 196 ms with global tables).  Layout of sx_lm_lds: [0, 32) the exp2 table as
// doubles with the table's bit patterns, [32, 64) logf's (1/c, log c) pairs, [64, 96) powf's (1/c, log2 c) pairs.
#if defined(__HIP_DEVICE_COMPILE__) && SX_EXACT_LIBM
#define SX_LM_LDS 1
#else
#define SX_LM_LDS 0
#endif

// __exp2f_data.tab (EXP2F_TABLE_BITS = 5): bits(2^(i/32)) - (i << 47)
#define SX_EXP2F_TAB { \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, \
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, \
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, \
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull }

// logf's and powf's tables: (1/c, log c) and (1/c, log2 c) at 16 centres c of [sqrt(1/2), sqrt 2)
#define SX_LOGF_TAB { \
        0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2, \
        0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3, \
        0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, \
        0x1p+0, 0x0p+0, 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4, \
        0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3, 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2, \
        0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 }
#define SX_POWLOG2_TAB { \
        0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2, 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2, \
        0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2, 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3, \
        0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4, 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, \
        0x1p+0, 0x0p+0, 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4, 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3, \
        0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3, 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2, 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2, \
        0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 }

#if SX_LM_LDS
__shared__ double sx_lm_lds[96];
__device__ const uint64_t sx_lm_src_e[32] = SX_EXP2F_TAB;
__device__ const double sx_lm_src_l[32] = SX_LOGF_TAB;
__device__ const double sx_lm_src_p[32] = SX_POWLOG2_TAB;
__device__ __forceinline__ void sx_libm_init() {       // every thread of the workgroup must call it, before any early return
    for (int i = threadIdx.x; i < 96; i += blockDim.x)
        sx_lm_lds[i] = i < 32 ? __longlong_as_double((long long)sx_lm_src_e[i]) : i < 64 ? sx_lm_src_l[i - 32] : sx_lm_src_p[i - 64];
    __syncthreads();
}
#define SX_LIBM_INIT() sx_libm_init()
#define SX_LM_EXP2(T, i) ((uint64_t)__double_as_longlong(sx_lm_lds[(i)]))
#define SX_LM_LOGF(T, i, j) (sx_lm_lds[32 + 2 * (i) + (j)])
#define SX_LM_POWL(T, i, j) (sx_lm_lds[64 + 2 * (i) + (j)])
#define SX_LM_DECL_E
#define SX_LM_DECL_L
#define SX_LM_DECL_P
#else
#define SX_LIBM_INIT()
#define SX_LM_EXP2(T, i) (T##e[(i)])
#define SX_LM_LOGF(T, i, j) (T##l[2 * (i) + (j)])
#define SX_LM_POWL(T, i, j) (T##p[2 * (i) + (j)])
#define SX_LM_DECL_E const uint64_t Te[32] = SX_EXP2F_TAB;
#define SX_LM_DECL_L const double Tl[32] = SX_LOGF_TAB;
#define SX_LM_DECL_P const double Tp[32] = SX_POWLOG2_TAB;
#endif

// expf (e_expf.c): x N/ln2 = k + r, exp(x) = 2^(k/N) 2^(r/N), degree-3 polynomial for 2^(r/N), N = 32
SX_HD float sx_g_expf(float x) {
    SX_LM_DECL_E
    const double InvLn2N = 0x1.71547652b82fep+5;                  // N / ln 2
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;   // poly_scaled
    const double xd = (double)x;
    const uint32_t abstop = (sx_f2u(x) >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                       // |x| >= 88 or NaN
        if (sx_f2u(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return sx_inff();                  // x > log(0x1p128)
        if (x < -0x1.9fe368p6f) return 0.0f;                      // x < log(0x1p-150)
    }
    const double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const uint64_t ki = sx_d2u(kd);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t = SX_LM_EXP2(T, ki % 32);
    t += ki << (52 - 5);
    const double s = sx_u2d(t);
    const double zz = sx_lm_fma(C0, r, C1);
    const double r2 = r * r;
    double y = sx_lm_fma(C2, r, 1.0);
    y = sx_lm_fma(zz, r2, y);
    y = y * s;
    return (float)y;
}

// logf (e_logf.c): x = 2^k z, z in [OFF, 2 OFF), table of 1/c and log c at 16 centres, degree-3 polynomial in r = z/c - 1
SX_HD float sx_g_logf(float x) {
    SX_LM_DECL_L
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = sx_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {          // x < 0x1p-126, inf or NaN
        if (ix * 2u == 0u) return -sx_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return sx_nanf();
        ix = sx_f2u(x * 0x1p23f);                                 // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = SX_LM_LOGF(T, i, 0), logc = SX_LM_LOGF(T, i, 1);
    const double z = (double)sx_u2f(iz);
    const double r = sx_lm_fma(z, invc, -1.0);
    const double y0 = sx_lm_fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = sx_lm_fma(A1, r, A2);
    y = sx_lm_fma(A0, r2, y);
    y = sx_lm_fma(y, r2, y0 + r);
    return (float)y;
}

// powf (e_powf.c): x^y = 2^(y log2 x), log2 with a 16-entry table and a degree-5 polynomial (relative error 1.3 2^-68 before
// rounding), 2^t as in exp2f.  Sign handling for negative bases with integer exponents is kept (the model never needs it).
SX_HD int sx_g_checkint(uint32_t iy) {      // 0: not an integer, 1: odd, 2: even
    const int e = (int)(iy >> 23 & 0xff);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
// log2_inline of e_powf.c for a positive normal x (ix = its bits)
SX_HD double sx_g_log2_core(uint32_t ix) {
    SX_LM_DECL_P
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                 A4 = 0x1.71547652ab82bp0;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = SX_LM_POWL(T, i, 0), logc = SX_LM_POWL(T, i, 1);
    const double z = (double)sx_u2f(iz);
    const double r = sx_lm_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = sx_lm_fma(A0, r, A1);
    const double p = sx_lm_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = sx_lm_fma(A4, r, y0);
    q = sx_lm_fma(p, r2, q);
    yy = sx_lm_fma(yy, r4, q);
    return yy;
}
// the tail of powf from ylogx = y * log2(x) on: range checks, then exp2_inline
SX_HD float sx_g_exp2_core(double ylogx, uint32_t sign_bias) {
    SX_LM_DECL_E
    const double SHIFT = 0x1.8p+47;                               // shift_scaled = 0x1.8p52 / N
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    if (((sx_d2u(ylogx) >> 47) & 0xffffu) >= (sx_d2u(126.0) >> 47)) {   // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -sx_inff() : sx_inff();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    double kd = ylogx + SHIFT;
    const uint64_t ki = sx_d2u(kd);
    kd -= SHIFT;                                                   // k/N
    const double rr = ylogx - kd;
    uint64_t t = SX_LM_EXP2(T, ki % 32);
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = sx_u2d(t);
    const double zz = sx_lm_fma(C0, rr, C1);
    const double rr2 = rr * rr;
    double w = sx_lm_fma(C2, rr, 1.0);
    w = sx_lm_fma(zz, rr2, w);
    w = w * s;
    return (float)w;
}
SX_HD float sx_g_powf(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = sx_f2u(x);
    const uint32_t iy = sx_f2u(y);
    const bool y_special = 2u * iy - 1u >= 2u * 0x7f800000u - 1u;         // zeroinfnan(iy)
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || y_special) {
        if (y_special) {
            if (2u * iy == 0u) return 1.0f;                        // (issignaling is not modelled)
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;   // |x| < 1 && y == inf or |x| > 1 && y == -inf
            return y * y;
        }
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {               // x is 0, inf or NaN
            float x2 = x * x;
            if ((ix & 0x80000000u) && sx_g_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {                                    // x < 0: finite only for integer y
            const int yint = sx_g_checkint(iy);
            if (yint == 0) return sx_nanf();
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                                    // subnormal x: normalise
            ix = sx_f2u(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double ylogx = (double)y * sx_g_log2_core(ix);           // cannot overflow: y is single precision
    return sx_g_exp2_core(ylogx, sign_bias);
}
// Two powers of one base: powf(x, y1) and powf(x, y2) evaluate the same log2_inline(x), so it is computed once -- bit for bit what
// two separate calls give.  Fast path for a positive normal x and ordinary exponents (every call site of the GR operators).
SX_HD void sx_g_powf2(float x, float y1, float y2, float* r1, float* r2) {
    const uint32_t ix = sx_f2u(x), i1 = sx_f2u(y1), i2 = sx_f2u(y2);
    const bool plain = ix - 0x00800000u < 0x7f800000u - 0x00800000u && !(2u * i1 - 1u >= 2u * 0x7f800000u - 1u) &&
                       !(2u * i2 - 1u >= 2u * 0x7f800000u - 1u);
    if (!plain) { *r1 = sx_g_powf(x, y1); *r2 = sx_g_powf(x, y2); return; }
    const double l2 = sx_g_log2_core(ix);
    *r1 = sx_g_exp2_core((double)y1 * l2, 0u);
    *r2 = sx_g_exp2_core((double)y2 * l2, 0u);
}
