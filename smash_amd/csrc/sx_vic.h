// sx_vic.h -- operators of the vic-a structure as device functions (gfx950), forward and adjoint.
//
// Forward: smash/solver/operator/md_vic_operator.f90:22-202 statement by statement (vic_infiltration,
// vic_vertical_transfer, vic_interflow, vic_baseflow, brooks_and_corey_flow, linear_evapotranspiration).
// Adjoint: the local adjoint expressions of forward_db.f90 in their order (VIC_INFILTRATION_B :6808-6973,
// VIC_VERTICAL_TRANSFER_B :7102-7197, VIC_INTERFLOW_B :7300-7368, VIC_BASEFLOW_B :7442-7503,
// BROOKS_AND_COREY_FLOW_B :7582-7643, LINEAR_EVAPOTRANSPIRATION_B :7700-7723), each given the levels its operator saw
// on entry.  libm: x**y with run-time exponents (1/(b+1), b+1, b, ...) -> sx_powf / sx_logf (fp64 evaluation, one
// rounding: correctly rounded like glibc's in 99.94 % of calls); the fixed powers of the interflow store (n = 5) use
// the same routines as gr_transfer; lambda = 1 in brooks_and_corey_flow is the identity.  Divisions go through sx_fdiv
// (the correctly rounded quotient for the normal-range operands the operators produce, 6 instructions instead of 11).
#pragma once

#include "sx_math.h"

#ifndef SX_DEV
#define SX_DEV __device__ __forceinline__
#endif

struct SxVicParams {
    float b, cusl1, cusl2, clsl, ks, ds, dsm, ws;
    float qs, rdw, tw;      // per-cell invariants of vic_baseflow, set by sx_vic_derive: (ds dsm) / ws, ds / ws, dsm / (1 - ws)
    SxDiv d1, d2, dl, d11, d22, dll;   // exact division by cusl1, cusl2, clsl and their squares (sx_div: 3 operations instead of 6)
};
SX_DEV void sx_vic_derive(SxVicParams& P) {
    P.qs = sx_fdiv((P.ds * P.dsm), P.ws);
    P.rdw = sx_fdiv(P.ds, P.ws);
    P.tw = sx_fdiv(P.dsm, -P.ws + 1.f);
    P.d1 = sx_mkdiv(P.cusl1); P.d2 = sx_mkdiv(P.cusl2); P.dl = sx_mkdiv(P.clsl);
    P.d11 = sx_mkdiv(P.cusl1 * P.cusl1); P.d22 = sx_mkdiv(P.cusl2 * P.cusl2); P.dll = sx_mkdiv(P.clsl * P.clsl);
}
struct SxVicGrads { float b_b, cusl1_b, cusl2_b, clsl_b, ks_b, ds_b, dsm_b, ws_b, husl1_b, husl2_b, hlsl_b; };

// ---------------------------------------------------------------- forward
// What the forward infiltration of a reverse step hands to its adjoint (round 3): the two bases' logarithms and the two powers.  The
// adjoint used to evaluate all of them again (and x**y a second time inside d/dy): 4 log2 + 8 exp2 in fp64 per rain step instead of
// 2 + 4.  Same values by construction -- sx_powb is a pure function of (base, exponent).
struct SxVicInfKeep { SxPowBase B1, B2; float pwr1a, pwr1b; };
template <bool KEEP>
SX_DEV void sx_vic_infiltration_k(const SxVicParams& P, float prcp, float cusl1, float cusl2, float b, float& husl1, float& husl2, float& runoff,
                                  SxVicInfKeep& K) {
    const float bp1 = b + 1.f;
    float ifl;
    if (prcp <= 0.f) ifl = 0.f;
    else {
        const float cusl = cusl1 + cusl2;
        float wusl = husl1 * cusl1 + husl2 * cusl2;
        wusl = fmaxf(1.e-6f, wusl);
        wusl = fminf(cusl - 1e-6f, wusl);
        const float iflm = cusl * bp1;
        const SxPowBase B1 = sx_powbase(1.f - (sx_fdiv(wusl, cusl)));
        const float pa = sx_powb(B1, sx_fdiv(1.f, bp1));
        if (KEEP) { K.B1 = B1; K.pwr1a = pa; }
        const float iflc = iflm * (1.f - pa);
        if (iflc + prcp >= iflm) ifl = cusl - wusl;
        else {
            const SxPowBase B2 = sx_powbase(1.f - (sx_fdiv((iflc + prcp), iflm)));
            const float pb = sx_powb(B2, bp1);
            if (KEEP) { K.B2 = B2; K.pwr1b = pb; }
            ifl = (cusl - wusl) - cusl * pb;
        }
        ifl = fminf(prcp, ifl);
    }
    const float ifl_usl1 = fminf((1.f - husl1) * cusl1, ifl);
    ifl = ifl - ifl_usl1;
    const float ifl_usl2 = fminf((1.f - husl2) * cusl2, ifl);
    ifl = ifl - ifl_usl2;
    husl1 = husl1 + sx_div(ifl_usl1, P.d1);
    husl2 = husl2 + sx_div(ifl_usl2, P.d2);
    runoff = prcp - (ifl_usl1 + ifl_usl2);
}
SX_DEV void sx_vic_infiltration(const SxVicParams& P, float prcp, float cusl1, float cusl2, float b, float& husl1, float& husl2, float& runoff) {
    SxVicInfKeep K;
    sx_vic_infiltration_k<false>(P, prcp, cusl1, cusl2, b, husl1, husl2, runoff, K);
}

// residual = 0, porosity = 1, lambda = 1 at both call sites (md_vic_operator.f90:94,99): pwx1 = h_upper / 1, pwx1**1 = pwx1
SX_DEV float sx_brooks_corey(float ks, float c_upper, float c_lower, float h_upper, float h_lower) {
    const float flow = ks * (h_upper - 0.f);           // (h_upper - residual) / (porosity - residual): x / 1 is x
    const float w_upper = h_upper * c_upper * 1.f;
    const float w_lower = h_lower * c_lower * 1.f;
    const float max_flow = fminf(w_upper, c_lower - w_lower);
    return fminf(max_flow, flow);
}
SX_DEV float sx_linear_evap(float e, float c, float h) { return fminf(c * h, e * h); }

// Steps without evaporative demand (pet = 0: every night hour of the reference's hourly PET disaggregation, core/_constant.py:47-75)
// leave the three linear_evapotranspiration stages with nothing to do when the layers hold water: fe = min(c h, 0 h) = +0, the level
// keeps its value (h - (+0) / c), pet_remain stays 0.  When that holds on EVERY lane of the wavefront (a ballot, a scalar branch,
// nothing is predicated) the stages are skipped, forward and reverse -- the general code with the zeros carried through by hand: in
// VIC_VERTICAL_TRANSFER_B every term of the three stages is then a product with fe = 0 or with w_b = 0 (c h > 0 = e h), br1 = br2 = 1
// zero pet_remain_b: the gradients keep their values (only the sign of a zero gradient, or a NaN the general code would have spread
// from an already infinite adjoint, can differ -- as for the still steps of the GR structures, sx_ops.h).
//   needs, per lane: pet == 0 and husl1, husl2, hlsl > 0 after the two brooks_and_corey exchanges (h = 0 takes the general code: there
//   c h > e h fails and linear_evapotranspiration_b routes the adjoint through w = c h).
#ifndef SX_VIC_NIGHT
#define SX_VIC_NIGHT 1
#endif
// what the forward vertical transfer of a reverse step hands to its adjoint (the levels each operator saw, the flows, the branches)
struct SxVicVT { float h1_0, h2_0, hl_0, fbc1, h2_1, fbc2, h1_2, h2_2, hl_2, fe1, fe2, fe3, pr1, pr2; bool br1, br2, night; };
template <bool KEEP>
SX_DEV void sx_vic_vertical_transfer_k(const SxVicParams& P, float pet, float cusl1, float cusl2, float clsl, float ks, float& husl1, float& husl2,
                                       float& hlsl, SxVicVT& K) {
    if (KEEP) { K.h1_0 = husl1; K.h2_0 = husl2; K.hl_0 = hlsl; }
    float fbc = sx_brooks_corey(ks, cusl1, cusl2, husl1, husl2);
    if (KEEP) K.fbc1 = fbc;
    husl1 = husl1 - sx_div(fbc, P.d1);
    husl2 = husl2 + sx_div(fbc, P.d2);
    if (KEEP) K.h2_1 = husl2;
    fbc = sx_brooks_corey(ks, cusl2, clsl, husl2, hlsl);
    if (KEEP) K.fbc2 = fbc;
    husl2 = husl2 - sx_div(fbc, P.d2);
    hlsl = hlsl + sx_div(fbc, P.dl);
    if (KEEP) { K.h1_2 = husl1; K.h2_2 = husl2; K.hl_2 = hlsl; K.fe1 = 0.f; K.fe2 = 0.f; K.fe3 = 0.f; K.pr1 = 0.f; K.pr2 = 0.f; K.br1 = true; K.br2 = true; }
    // (the reverse step only: in the forward kernel the three stages are ~25 instructions and the test costs what it saves --
    // measured 47.0 -> 47.5 ms with it; the two forms give the same bits, so the kernels need not agree on which one runs)
    const bool night = SX_VIC_NIGHT && KEEP && __builtin_amdgcn_ballot_w64(!(pet == 0.f && husl1 > 0.f && husl2 > 0.f && hlsl > 0.f)) == 0ull;   // over the active lanes
    if (KEEP) K.night = night;
    if (night) return;
    float fe = sx_linear_evap(pet, cusl1, husl1);
    husl1 = husl1 - sx_div(fe, P.d1);
    float pet_remain;
    if (KEEP) { K.fe1 = fe; K.br1 = !(0.f < pet - fe); }
    pet_remain = fmaxf(0.f, pet - fe);
    if (KEEP) K.pr1 = (0.f < pet - fe) ? pet - fe : 0.f;
    fe = sx_linear_evap(pet_remain, cusl2, husl2);
    husl2 = husl2 - sx_div(fe, P.d2);
    if (KEEP) { K.fe2 = fe; K.br2 = !(0.f < pet_remain - fe); K.pr2 = (0.f < pet_remain - fe) ? pet_remain - fe : 0.f; }
    pet_remain = fmaxf(0.f, pet_remain - fe);
    fe = sx_linear_evap(pet_remain, clsl, hlsl);
    if (KEEP) K.fe3 = fe;
    hlsl = hlsl - sx_div(fe, P.dl);
}
SX_DEV void sx_vic_vertical_transfer(const SxVicParams& P, float pet, float cusl1, float cusl2, float clsl, float ks, float& husl1, float& husl2, float& hlsl) {
    SxVicVT K;
    sx_vic_vertical_transfer_k<false>(P, pet, cusl1, cusl2, clsl, ks, husl1, husl2, hlsl, K);
}

SX_DEV void sx_vic_interflow(const SxVicParams& P, float cusl2, float cusl2_m4, float& husl2, float& qi) {   // n = 5
    const float husl2_imd = husl2;
    husl2 = sx_div(sx_pow_m025(sx_pow_m4(husl2_imd * cusl2) + cusl2_m4), P.d2);
    qi = (husl2_imd - husl2) * cusl2;
}

SX_DEV void sx_vic_baseflow(const SxVicParams& P, float& hlsl, float& qb) {
    const float clsl = P.clsl, dsm = P.dsm, ws = P.ws;
    float q;
    if (hlsl <= ws) q = P.qs * hlsl;
    else q = sx_fdiv(dsm * (1.f - P.rdw) * (hlsl - ws), 1.f - ws);
    const float wlsl = clsl * hlsl;
    q = fminf(wlsl, q);
    hlsl = hlsl - sx_div(q, P.dl);
    qb = q;
}

// one vertical cell-step of vic_a_forward (md_forward_structure.f90:804-829); returns qt
SX_DEV float sx_vic_step(const SxVicParams& P, float cusl2_m4, float prcp, float pet, float& husl1, float& husl2, float& hlsl) {
    float runoff = 0.f, qi, qb;
    if (prcp >= 0.f && pet >= 0.f) {
        sx_vic_infiltration(P, prcp, P.cusl1, P.cusl2, P.b, husl1, husl2, runoff);
        sx_vic_vertical_transfer(P, pet, P.cusl1, P.cusl2, P.clsl, P.ks, husl1, husl2, hlsl);
    }
    sx_vic_interflow(P, P.cusl2, cusl2_m4, husl2, qi);
    sx_vic_baseflow(P, hlsl, qb);
    return (runoff + qi + qb);
}

// ---------------------------------------------------------------- adjoint
SX_DEV float sx_pow_guard_b(float x, float y, float r_b) {
    if (x <= 0.f && (y == 0.f || y != (float)(int)y)) return 0.f;
    return y * sx_powf(x, y - 1.f) * r_b;
}

// K: from sx_vic_infiltration_k<true> on the same (prcp, husl1, husl2)
SX_DEV void sx_vic_infiltration_b(const SxVicParams& P, const SxVicInfKeep& K, float prcp, float cusl1, float cusl2, float b, float husl1, float husl2, float runoff_b, SxVicGrads& G) {
    float bp1 = b + 1.f, ifl, cusl = 0.f, wusl = 0.f, iflm = 0.f, iflc = 0.f, pwx1 = 0.f, pwy1 = 0.f, pwr1 = 0.f, pwr1_first = 0.f, pwr1_second = 0.f;
    int c_prcp, c_w1 = 0, c_w2 = 0, c_full = 0, c_min;
    if (prcp <= 0.f) { ifl = 0.f; c_prcp = 0; c_min = 0; }
    else {
        c_prcp = 1;
        cusl = cusl1 + cusl2;
        wusl = husl1 * cusl1 + husl2 * cusl2;
        if (1.e-6f < wusl) c_w1 = 0; else { wusl = 1.e-6f; c_w1 = 1; }
        if (cusl - 1e-6f > wusl) c_w2 = 0; else { wusl = cusl - 1e-6f; c_w2 = 1; }
        iflm = cusl * bp1;
        pwx1 = 1.f - sx_fdiv(wusl, cusl);
        pwy1 = sx_fdiv(1.f, bp1);
        pwr1 = K.pwr1a;                            // sx_powb(K.B1, pwy1), K.B1 = sx_powbase(pwx1): evaluated by the forward part of this step
        iflc = iflm * (1.f - pwr1);
        if (iflc + prcp >= iflm) { ifl = cusl - wusl; c_full = 1; }
        else {
            pwx1 = 1.f - sx_fdiv((iflc + prcp), iflm);
            pwr1_first = pwr1;
            pwr1 = K.pwr1b;                        // sx_powb(K.B2, bp1), K.B2 = sx_powbase(pwx1)
            pwr1_second = pwr1;
            ifl = cusl - wusl - cusl * pwr1;
            c_full = 0;
        }
        if (prcp > ifl) c_min = 2; else { ifl = prcp; c_min = 1; }
    }
    float ifl_usl1, ifl_usl2;
    int c_u1, c_u2;
    if ((1.f - husl1) * cusl1 > ifl) { ifl_usl1 = ifl; c_u1 = 0; } else { ifl_usl1 = (1.f - husl1) * cusl1; c_u1 = 1; }
    ifl = ifl - ifl_usl1;
    if ((1.f - husl2) * cusl2 > ifl) { ifl_usl2 = ifl; c_u2 = 0; } else { ifl_usl2 = (1.f - husl2) * cusl2; c_u2 = 1; }
    float ifl_usl1_b = sx_div(G.husl1_b, P.d1) - runoff_b;
    float ifl_usl2_b = sx_div(G.husl2_b, P.d2) - runoff_b;
    G.cusl2_b = G.cusl2_b - sx_div(ifl_usl2 * G.husl2_b, P.d22);
    G.cusl1_b = G.cusl1_b - sx_div(ifl_usl1 * G.husl1_b, P.d11);
    float ifl_b;
    if (c_u2 == 0) ifl_b = ifl_usl2_b;
    else {
        G.husl2_b = G.husl2_b - cusl2 * ifl_usl2_b;
        G.cusl2_b = G.cusl2_b + (1.f - husl2) * ifl_usl2_b;
        ifl_b = 0.f;
    }
    ifl_usl1_b = ifl_usl1_b - ifl_b;
    if (c_u1 == 0) ifl_b = ifl_b + ifl_usl1_b;
    else {
        G.husl1_b = G.husl1_b - cusl1 * ifl_usl1_b;
        G.cusl1_b = G.cusl1_b + (1.f - husl1) * ifl_usl1_b;
    }
    float bp1_b;
    if (c_prcp == 0) bp1_b = 0.f;
    else {
        if (c_min == 1) ifl_b = 0.f;
        float cusl_b, wusl_b, iflc_b, iflm_b, pwr1_b, pwx1_b, pwy1_b;
        if (c_full == 0) {
            cusl_b = (1.0f - pwr1) * ifl_b;
            wusl_b = -ifl_b;
            pwr1_b = -(cusl * ifl_b);
            pwr1 = pwr1_first;
            if (pwx1 <= 0.0f && (bp1 == 0.0f || bp1 != (float)(int)bp1)) pwx1_b = 0.f;
            else pwx1_b = bp1 * sx_powb(K.B2, bp1 - 1.f) * pwr1_b;
            if (pwx1 <= 0.0f) bp1_b = 0.f;
            else bp1_b = pwr1_second * sx_logb(K.B2) * pwr1_b;          // pwx1**bp1 is the forward value
            iflc_b = -(sx_fdiv(pwx1_b, iflm));
            iflm_b = sx_fdiv((prcp + iflc) * pwx1_b, iflm * iflm);
            pwy1 = sx_fdiv(1.f, bp1);
            pwx1 = 1.f - sx_fdiv(wusl, cusl);
        } else {
            cusl_b = ifl_b;
            wusl_b = -ifl_b;
            iflc_b = 0.f;
            iflm_b = 0.f;
            bp1_b = 0.f;
        }
        iflm_b = iflm_b + (1.f - pwr1) * iflc_b;
        pwr1_b = -(iflm * iflc_b);
        if (pwx1 <= 0.0f && (pwy1 == 0.0f || pwy1 != (float)(int)pwy1)) pwx1_b = 0.f;
        else pwx1_b = pwy1 * sx_powb(K.B1, pwy1 - 1.f) * pwr1_b;
        if (pwx1 <= 0.0f) pwy1_b = 0.f;
        else pwy1_b = pwr1 * sx_logb(K.B1) * pwr1_b;                      // pwx1**pwy1 is the forward value (pwr1 holds it here)
        bp1_b = bp1_b + cusl * iflm_b - sx_fdiv(pwy1_b, bp1 * bp1);
        wusl_b = wusl_b - sx_fdiv(pwx1_b, cusl);
        cusl_b = cusl_b + sx_fdiv(wusl * pwx1_b, cusl * cusl) + bp1 * iflm_b;
        if (c_w2 != 0) { cusl_b = cusl_b + wusl_b; wusl_b = 0.f; }
        if (c_w1 != 0) wusl_b = 0.f;
        G.husl1_b = G.husl1_b + cusl1 * wusl_b;
        G.cusl1_b = G.cusl1_b + husl1 * wusl_b + cusl_b;
        G.husl2_b = G.husl2_b + cusl2 * wusl_b;
        G.cusl2_b = G.cusl2_b + husl2 * wusl_b + cusl_b;
    }
    G.b_b = G.b_b + bp1_b;
}

// residual = 0, porosity = 1, lambda = 1: pwr1 = pwx1, d(pwx1**1) = pwr1_b
SX_DEV void sx_brooks_corey_b(float ks, float& ks_b, float c_upper, float& c_upper_b, float c_lower, float& c_lower_b, float h_upper,
                              float& h_upper_b, float h_lower, float& h_lower_b, float flow_b) {
    const float pwx1 = (h_upper - 0.f);                // / (1 - 0): x / 1 is x
    const float pwr1 = pwx1;
    const float flow = ks * pwr1;
    const float w_upper = h_upper * c_upper * 1.f;
    const float w_lower = h_lower * c_lower * 1.f;
    float max_flow, max_flow_b, w_lower_b, w_upper_b;
    int br;
    if (w_upper > c_lower - w_lower) { max_flow = c_lower - w_lower; br = 0; } else { max_flow = w_upper; br = 1; }
    if (max_flow > flow) max_flow_b = 0.f;
    else { max_flow_b = flow_b; flow_b = 0.f; }
    if (br == 0) { c_lower_b = c_lower_b + max_flow_b; w_lower_b = -max_flow_b; w_upper_b = 0.f; }
    else { w_upper_b = max_flow_b; w_lower_b = 0.f; }
    const float pwr1_b = ks * flow_b;
    const float pwx1_b = 1.f * 1.f * pwr1_b;          // lambda * pwx1**(lambda - 1) * pwr1_b
    h_lower_b = h_lower_b + c_lower * 1.f * w_lower_b;
    c_lower_b = c_lower_b + h_lower * 1.f * w_lower_b;
    h_upper_b = h_upper_b + c_upper * 1.f * w_upper_b + pwx1_b;
    c_upper_b = c_upper_b + h_upper * 1.f * w_upper_b;
    ks_b = ks_b + pwr1 * flow_b;
}

SX_DEV void sx_linear_evap_b(float e, float& e_b, float c, float& c_b, float h, float& h_b, float flow_b) {
    const float flow = e * h, w = c * h;
    float w_b;
    if (w > flow) w_b = 0.f;
    else { w_b = flow_b; flow_b = 0.f; }
    c_b = c_b + h * w_b;
    h_b = h_b + c * w_b + e * flow_b;
    e_b = e_b + h * flow_b;
}

// K: from sx_vic_vertical_transfer_k<true> on the levels on entry of vic_vertical_transfer
SX_DEV void sx_vic_vertical_transfer_b(const SxVicParams& P, const SxVicVT& K, float pet, float cusl1, float cusl2, float clsl, float ks, SxVicGrads& G) {
    const float h1_0 = K.h1_0, h2_0 = K.h2_0, hl_1 = K.hl_0, fbc1 = K.fbc1, h2_1 = K.h2_1, fbc2 = K.fbc2;
    if (!K.night) {        // wave-uniform
        const float h1_2 = K.h1_2, h2_2 = K.h2_2, hl_2 = K.hl_2, fe1 = K.fe1, fe2 = K.fe2, fe3 = K.fe3;
        const float pet_remain1 = K.pr1, pet_remain2 = K.pr2;
        float fe_b = -(sx_div(G.hlsl_b, P.dl));
        G.clsl_b = G.clsl_b + sx_div(fe3 * G.hlsl_b, P.dll);
        float pet_remain_b = 0.f;
        sx_linear_evap_b(pet_remain2, pet_remain_b, clsl, G.clsl_b, hl_2, G.hlsl_b, fe_b);
        if (!K.br2) fe_b = -pet_remain_b;
        else { pet_remain_b = 0.f; fe_b = 0.f; }
        fe_b = fe_b - sx_div(G.husl2_b, P.d2);
        G.cusl2_b = G.cusl2_b + sx_div(fe2 * G.husl2_b, P.d22);
        sx_linear_evap_b(pet_remain1, pet_remain_b, cusl2, G.cusl2_b, h2_2, G.husl2_b, fe_b);
        if (!K.br1) fe_b = -pet_remain_b;
        else fe_b = 0.f;
        fe_b = fe_b - sx_div(G.husl1_b, P.d1);
        G.cusl1_b = G.cusl1_b + sx_div(fe1 * G.husl1_b, P.d11);
        float pet_b = 0.f;
        sx_linear_evap_b(pet, pet_b, cusl1, G.cusl1_b, h1_2, G.husl1_b, fe_b);
    }
    float fbc_b = sx_div(G.hlsl_b, P.dl) - sx_div(G.husl2_b, P.d2);
    G.clsl_b = G.clsl_b - sx_div(fbc2 * G.hlsl_b, P.dll);
    G.cusl2_b = G.cusl2_b + sx_div(fbc2 * G.husl2_b, P.d22);
    sx_brooks_corey_b(ks, G.ks_b, cusl2, G.cusl2_b, clsl, G.clsl_b, h2_1, G.husl2_b, hl_1, G.hlsl_b, fbc_b);
    fbc_b = sx_div(G.husl2_b, P.d2) - sx_div(G.husl1_b, P.d1);
    G.cusl2_b = G.cusl2_b - sx_div(fbc1 * G.husl2_b, P.d22);
    G.cusl1_b = G.cusl1_b + sx_div(fbc1 * G.husl1_b, P.d11);
    sx_brooks_corey_b(ks, G.ks_b, cusl1, G.cusl1_b, cusl2, G.cusl2_b, h1_0, G.husl1_b, h2_0, G.husl2_b, fbc_b);
}

// husl2: level on entry; n = 5
SX_DEV void sx_vic_interflow_b(const SxVicParams& P, float cusl2, float cusl2_m4, float cusl2_m5, float husl2, float qi_b, SxVicGrads& G) {
    const float husl2_imd = husl2;
    const float pwx1 = husl2_imd * cusl2;
    float pwr1, pwx1_m5;
    sx_pow_m4_m5(pwx1, &pwr1, &pwx1_m5);
    const float pwx3 = pwr1 + cusl2_m4;
    float pwr3, pwx3_m125;
    sx_pow_m025_m125(pwx3, &pwr3, &pwx3_m125);
    const float husl2_new = sx_div(pwr3, P.d2);
    const float hb = G.husl2_b - cusl2 * qi_b;
    const float pwr3_b = sx_div(hb, P.d2);
    const float pwx3_b = (pwx3 <= 0.f) ? 0.f : -0.25f * pwx3_m125 * pwr3_b;
    const float pwr1_b = pwx3_b, pwr2_b = pwx3_b;
    const float pwx1_b = -4.f * pwx1_m5 * pwr1_b;
    const float husl2_imd_b = cusl2 * qi_b + cusl2 * pwx1_b;
    G.cusl2_b = G.cusl2_b + (husl2_imd - husl2_new) * qi_b + -4.f * cusl2_m5 * pwr2_b - sx_div(pwr3 * hb, P.d22) + husl2_imd * pwx1_b;
    G.husl2_b = husl2_imd_b;
}

// hlsl: level on entry
SX_DEV void sx_vic_baseflow_b(const SxVicParams& P, float hlsl, float qb_b, SxVicGrads& G) {
    const float clsl = P.clsl, ds = P.ds, dsm = P.dsm, ws = P.ws;
    float qb;
    int br1, br2;
    if (hlsl <= ws) { qb = P.qs * hlsl; br1 = 1; }
    else { qb = sx_fdiv(dsm * (1.f - P.rdw) * (hlsl - ws), 1.f - ws); br1 = 0; }
    const float wlsl = clsl * hlsl;
    if (wlsl > qb) br2 = 0; else { qb = wlsl; br2 = 1; }
    qb_b = qb_b - sx_div(G.hlsl_b, P.dl);
    G.clsl_b = G.clsl_b + sx_div(qb * G.hlsl_b, P.dll);
    float wlsl_b;
    if (br2 == 0) wlsl_b = 0.f;
    else { wlsl_b = qb_b; qb_b = 0.f; }
    G.clsl_b = G.clsl_b + hlsl * wlsl_b;
    G.hlsl_b = G.hlsl_b + clsl * wlsl_b;
    if (br1 == 0) {
        const float temp = P.tw;
        const float temp_b0 = -(sx_fdiv((hlsl - ws) * temp * qb_b, ws));
        const float temp_b1 = (1.f - P.rdw) * qb_b;
        G.hlsl_b = G.hlsl_b + temp * temp_b1;
        const float temp_b = sx_fdiv((hlsl - ws) * temp_b1, 1.f - ws);
        G.ws_b = G.ws_b + temp * temp_b - temp * temp_b1 - sx_fdiv(ds * temp_b0, ws);
        G.dsm_b = G.dsm_b + temp_b;
        G.ds_b = G.ds_b + temp_b0;
    } else {
        const float temp = sx_fdiv(hlsl, ws);
        G.ds_b = G.ds_b + dsm * temp * qb_b;
        G.dsm_b = G.dsm_b + ds * temp * qb_b;
        const float temp_b = sx_fdiv(ds * dsm * qb_b, ws);
        G.hlsl_b = G.hlsl_b + temp_b;
        G.ws_b = G.ws_b - temp * temp_b;
    }
}

// reverse of sx_vic_step given the pre-step levels and qt_b (inner body of VIC_A_FORWARD_B, forward_db.f90:10271-10316)
SX_DEV void sx_vic_step_b(const SxVicParams& P, float cusl2_m4, float cusl2_m5, float prcp, float pet, float husl1, float husl2, float hlsl,
                          float qt_b, SxVicGrads& G) {
    const bool wet = (prcp >= 0.f && pet >= 0.f);
    float h1 = husl1, h2 = husl2, hl = hlsl, runoff = 0.f;
    SxVicInfKeep K;
    K.B1 = sx_powbase_one(); K.B2 = K.B1; K.pwr1a = 0.f; K.pwr1b = 0.f;
    SxVicVT V;
    V.h1_0 = V.h2_0 = V.hl_0 = V.fbc1 = V.h2_1 = V.fbc2 = V.h1_2 = V.h2_2 = V.hl_2 = V.fe1 = V.fe2 = V.fe3 = V.pr1 = V.pr2 = 0.f;
    V.br1 = V.br2 = V.night = true;
    if (wet) {
        sx_vic_infiltration_k<true>(P, prcp, P.cusl1, P.cusl2, P.b, h1, h2, runoff, K);
        sx_vic_vertical_transfer_k<true>(P, pet, P.cusl1, P.cusl2, P.clsl, P.ks, h1, h2, hl, V);
    }
    sx_vic_baseflow_b(P, hl, qt_b, G);
    sx_vic_interflow_b(P, P.cusl2, cusl2_m4, cusl2_m5, h2, qt_b, G);
    if (wet) {
        sx_vic_vertical_transfer_b(P, V, pet, P.cusl1, P.cusl2, P.clsl, P.ks, G);
        sx_vic_infiltration_b(P, K, prcp, P.cusl1, P.cusl2, P.b, husl1, husl2, qt_b, G);
    }
}

// ---------------------------------------------------------------- tangent (VIC_*_D, forward_db.f90:6691-7695)
// (value, tangent) pairs; the primal follows the _D code, which re-associates vic_baseflow.
struct SxVD { float v, d; };
SX_DEV SxVD sx_vd(float v, float d) { SxVD x; x.v = v; x.d = d; return x; }

SX_DEV float sx_powd_full(float x, float y, float x_d, float y_d, float& r) {   // d(x**y), both active: Tapenade's three cases
    const SxPowBase B = sx_powbase(x);
    const float t = sx_powb(B, y);
    r = t;
    if (x <= 0.f && (y == 0.f || y != (float)(int)y)) return 0.f;
    if (x <= 0.f) return y * sx_powb(B, y - 1.f) * x_d;
    return y * sx_powb(B, y - 1.f) * x_d + t * sx_logb(B) * y_d;
}

SX_DEV void sx_vic_infiltration_d(float prcp, SxVD cusl1, SxVD cusl2, SxVD b, SxVD& husl1, SxVD& husl2, SxVD& runoff) {
    const float bp1_d = b.d, bp1 = b.v + 1.f;
    float ifl, ifl_d;
    if (prcp <= 0.f) { ifl = 0.f; ifl_d = 0.f; }
    else {
        const float cusl_d = cusl1.d + cusl2.d, cusl = cusl1.v + cusl2.v;
        float wusl_d = cusl1.v * husl1.d + husl1.v * cusl1.d + cusl2.v * husl2.d + husl2.v * cusl2.d;
        float wusl = husl1.v * cusl1.v + husl2.v * cusl2.v;
        if (!(1.e-6f < wusl)) { wusl = 1.e-6f; wusl_d = 0.f; }
        if (!(cusl - 1e-6f > wusl)) { wusl_d = cusl_d; wusl = cusl - 1e-6f; }
        const float iflm_d = bp1 * cusl_d + cusl * bp1_d, iflm = cusl * bp1;
        float pwx1_d = -(sx_fdiv((wusl_d - sx_fdiv(wusl * cusl_d, cusl)), cusl)), pwx1 = 1.f - sx_fdiv(wusl, cusl);
        const float pwy1_d = -(sx_fdiv(bp1_d, bp1 * bp1)), pwy1 = sx_fdiv(1.f, bp1);
        float pwr1;
        float pwr1_d = sx_powd_full(pwx1, pwy1, pwx1_d, pwy1_d, pwr1);
        const float iflc_d = (1.f - pwr1) * iflm_d - iflm * pwr1_d, iflc = iflm * (1.f - pwr1);
        if (iflc + prcp >= iflm) { ifl_d = cusl_d - wusl_d; ifl = cusl - wusl; }
        else {
            const float temp = sx_fdiv((prcp + iflc), iflm);
            pwx1_d = -(sx_fdiv((iflc_d - temp * iflm_d), iflm));
            pwx1 = 1.f - temp;
            pwr1_d = sx_powd_full(pwx1, bp1, pwx1_d, bp1_d, pwr1);
            ifl_d = (1.0f - pwr1) * cusl_d - wusl_d - cusl * pwr1_d;
            ifl = cusl - wusl - cusl * pwr1;
        }
        if (!(prcp > ifl)) { ifl = prcp; ifl_d = 0.f; }
    }
    float u1, u1_d, u2, u2_d;
    if ((1.f - husl1.v) * cusl1.v > ifl) { u1_d = ifl_d; u1 = ifl; }
    else { u1_d = (1.f - husl1.v) * cusl1.d - cusl1.v * husl1.d; u1 = (1.f - husl1.v) * cusl1.v; }
    ifl_d = ifl_d - u1_d; ifl = ifl - u1;
    if ((1.f - husl2.v) * cusl2.v > ifl) { u2_d = ifl_d; u2 = ifl; }
    else { u2_d = (1.f - husl2.v) * cusl2.d - cusl2.v * husl2.d; u2 = (1.f - husl2.v) * cusl2.v; }
    husl1.d = husl1.d + sx_fdiv((u1_d - sx_fdiv(u1 * cusl1.d, cusl1.v)), cusl1.v);
    husl1.v = husl1.v + sx_fdiv(u1, cusl1.v);
    husl2.d = husl2.d + sx_fdiv((u2_d - sx_fdiv(u2 * cusl2.d, cusl2.v)), cusl2.v);
    husl2.v = husl2.v + sx_fdiv(u2, cusl2.v);
    runoff.d = -u1_d - u2_d;
    runoff.v = prcp - (u1 + u2);
}
SX_DEV SxVD sx_brooks_corey_d(SxVD ks, SxVD c_upper, SxVD c_lower, SxVD h_upper, SxVD h_lower) {
    SxVD flow;
    const float pwx1_d = sx_fdiv(h_upper.d, 1.f - 0.f), pwx1 = sx_fdiv((h_upper.v - 0.f), 1.f - 0.f);
    const float pwr1_d = 1.f * 1.f * pwx1_d, pwr1 = pwx1;
    flow.d = pwr1 * ks.d + ks.v * pwr1_d;
    flow.v = ks.v * pwr1;
    const float w_upper_d = 1.f * (c_upper.v * h_upper.d + h_upper.v * c_upper.d), w_upper = h_upper.v * c_upper.v * 1.f;
    const float w_lower_d = 1.f * (c_lower.v * h_lower.d + h_lower.v * c_lower.d), w_lower = h_lower.v * c_lower.v * 1.f;
    float max_flow, max_flow_d;
    if (w_upper > c_lower.v - w_lower) { max_flow_d = c_lower.d - w_lower_d; max_flow = c_lower.v - w_lower; }
    else { max_flow_d = w_upper_d; max_flow = w_upper; }
    if (!(max_flow > flow.v)) { flow.d = max_flow_d; flow.v = max_flow; }
    return flow;
}
SX_DEV SxVD sx_linear_evap_d(SxVD e, SxVD c, SxVD h) {
    SxVD flow;
    flow.d = h.v * e.d + e.v * h.d; flow.v = e.v * h.v;
    const float w_d = h.v * c.d + c.v * h.d, w = c.v * h.v;
    if (!(w > flow.v)) { flow.d = w_d; flow.v = w; }
    return flow;
}
SX_DEV void sx_vic_vertical_transfer_d(float pet, SxVD cusl1, SxVD cusl2, SxVD clsl, SxVD ks, SxVD& husl1, SxVD& husl2, SxVD& hlsl) {
    SxVD fbc = sx_brooks_corey_d(ks, cusl1, cusl2, husl1, husl2);
    husl1.d = husl1.d - sx_fdiv((fbc.d - sx_fdiv(fbc.v * cusl1.d, cusl1.v)), cusl1.v); husl1.v = husl1.v - sx_fdiv(fbc.v, cusl1.v);
    husl2.d = husl2.d + sx_fdiv((fbc.d - sx_fdiv(fbc.v * cusl2.d, cusl2.v)), cusl2.v); husl2.v = husl2.v + sx_fdiv(fbc.v, cusl2.v);
    fbc = sx_brooks_corey_d(ks, cusl2, clsl, husl2, hlsl);
    husl2.d = husl2.d - sx_fdiv((fbc.d - sx_fdiv(fbc.v * cusl2.d, cusl2.v)), cusl2.v); husl2.v = husl2.v - sx_fdiv(fbc.v, cusl2.v);
    hlsl.d = hlsl.d + sx_fdiv((fbc.d - sx_fdiv(fbc.v * clsl.d, clsl.v)), clsl.v); hlsl.v = hlsl.v + sx_fdiv(fbc.v, clsl.v);
    SxVD fe = sx_linear_evap_d(sx_vd(pet, 0.f), cusl1, husl1);
    husl1.d = husl1.d - sx_fdiv((fe.d - sx_fdiv(fe.v * cusl1.d, cusl1.v)), cusl1.v); husl1.v = husl1.v - sx_fdiv(fe.v, cusl1.v);
    SxVD pr;
    if (0.f < pet - fe.v) { pr.d = -fe.d; pr.v = pet - fe.v; } else { pr.v = 0.f; pr.d = 0.f; }
    fe = sx_linear_evap_d(pr, cusl2, husl2);
    husl2.d = husl2.d - sx_fdiv((fe.d - sx_fdiv(fe.v * cusl2.d, cusl2.v)), cusl2.v); husl2.v = husl2.v - sx_fdiv(fe.v, cusl2.v);
    if (0.f < pr.v - fe.v) { pr.d = pr.d - fe.d; pr.v = pr.v - fe.v; } else { pr.v = 0.f; pr.d = 0.f; }
    fe = sx_linear_evap_d(pr, clsl, hlsl);
    hlsl.d = hlsl.d - sx_fdiv((fe.d - sx_fdiv(fe.v * clsl.d, clsl.v)), clsl.v); hlsl.v = hlsl.v - sx_fdiv(fe.v, clsl.v);
}
SX_DEV void sx_vic_interflow_d(SxVD cusl2, float cusl2_m4, float cusl2_m5, SxVD& husl2, SxVD& qi) {   // n = 5
    const SxVD him = husl2;
    const float pwx1_d = cusl2.v * him.d + him.v * cusl2.d, pwx1 = him.v * cusl2.v;
    float pwr1, pwx1_m5;
    sx_pow_m4_m5(pwx1, &pwr1, &pwx1_m5);
    const float pwr1_d = -4.f * pwx1_m5 * pwx1_d;
    const float pwr2_d = -4.f * cusl2_m5 * cusl2.d;
    const float pwx3_d = pwr1_d + pwr2_d, pwx3 = pwr1 + cusl2_m4;
    float pwr3, pwx3_m125;
    sx_pow_m025_m125(pwx3, &pwr3, &pwx3_m125);
    const float pwr3_d = (pwx3 <= 0.f) ? 0.f : -0.25f * pwx3_m125 * pwx3_d;
    husl2.d = sx_fdiv((pwr3_d - sx_fdiv(pwr3 * cusl2.d, cusl2.v)), cusl2.v);
    husl2.v = sx_fdiv(pwr3, cusl2.v);
    qi.d = cusl2.v * (him.d - husl2.d) + (him.v - husl2.v) * cusl2.d;
    qi.v = (him.v - husl2.v) * cusl2.v;
}
SX_DEV void sx_vic_baseflow_d(SxVD clsl, SxVD ds, SxVD dsm, SxVD ws, SxVD& hlsl, SxVD& qb) {
    float q, q_d;
    if (hlsl.v <= ws.v) {
        const float temp = sx_fdiv(hlsl.v, ws.v);
        q_d = temp * (dsm.v * ds.d + ds.v * dsm.d) + sx_fdiv(ds.v * dsm.v * (hlsl.d - temp * ws.d), ws.v);
        q = ds.v * dsm.v * temp;
    } else {
        const float temp = sx_fdiv(dsm.v, -ws.v + 1.f), temp0 = sx_fdiv(ds.v, ws.v);
        q_d = (1.f - temp0) * (temp * (hlsl.d - ws.d) + sx_fdiv((hlsl.v - ws.v) * (dsm.d + temp * ws.d), 1.f - ws.v)) -
              sx_fdiv((hlsl.v - ws.v) * temp * (ds.d - temp0 * ws.d), ws.v);
        q = (1.f - temp0) * ((hlsl.v - ws.v) * temp);
    }
    const float wlsl_d = hlsl.v * clsl.d + clsl.v * hlsl.d, wlsl = clsl.v * hlsl.v;
    if (!(wlsl > q)) { q_d = wlsl_d; q = wlsl; }
    hlsl.d = hlsl.d - sx_fdiv((q_d - sx_fdiv(q * clsl.d, clsl.v)), clsl.v);
    hlsl.v = hlsl.v - sx_fdiv(q, clsl.v);
    qb.d = q_d; qb.v = q;
}
struct SxVicTan { float b_d, cusl1_d, cusl2_d, clsl_d, ks_d, ds_d, dsm_d, ws_d; };
// one vertical cell-step of VIC_A_FORWARD_D (forward_db.f90:9949-10095): returns qt (value, tangent)
SX_DEV SxVD sx_vic_step_d(const SxVicParams& P, const SxVicTan& D, float cusl2_m4, float cusl2_m5, float prcp, float pet, SxVD& husl1,
                          SxVD& husl2, SxVD& hlsl) {
    SxVD runoff = sx_vd(0.f, 0.f), qi, qb, qt;
    const SxVD cusl1 = sx_vd(P.cusl1, D.cusl1_d), cusl2 = sx_vd(P.cusl2, D.cusl2_d), clsl = sx_vd(P.clsl, D.clsl_d);
    if (prcp >= 0.f && pet >= 0.f) {
        sx_vic_infiltration_d(prcp, cusl1, cusl2, sx_vd(P.b, D.b_d), husl1, husl2, runoff);
        sx_vic_vertical_transfer_d(pet, cusl1, cusl2, clsl, sx_vd(P.ks, D.ks_d), husl1, husl2, hlsl);
    }
    sx_vic_interflow_d(cusl2, cusl2_m4, cusl2_m5, husl2, qi);
    sx_vic_baseflow_d(clsl, sx_vd(P.ds, D.ds_d), sx_vd(P.dsm, D.dsm_d), sx_vd(P.ws, D.ws_d), hlsl, qb);
    qt.d = runoff.d + qi.d + qb.d;
    qt.v = runoff.v + qi.v + qb.v;
    return qt;
}
