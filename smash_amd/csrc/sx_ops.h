// sx_ops.h -- per-cell operators of the hot path as device functions (gfx950), forward and adjoint.
//
// Forward operators restate, operation by operation (fp32, no contraction):
//   gr_interception  smash/solver/operator/md_gr_operator.f90:20-34
//   gr_production    md_gr_operator.f90:36-67      (beta = 1000 at every call site, md_forward_structure.f90:122,306,492,677)
//   gr_exchange      md_gr_operator.f90:69-79
//   gr_transfer      md_gr_operator.f90:81-110     (n = 5)
// Adjoint operators evaluate the local adjoint expressions of the Tapenade output in its order:
//   GR_INTERCEPTION_B forward_db.f90:5878-5925, GR_PRODUCTION_B :6012-6103, GR_EXCHANGE_B :6147-6157,
//   GR_TRANSFER_B :6275-6412.
// libm calls are replaced by the sx_math.h routines (see its header for the parity argument).
#pragma once

#include "sx_math.h"

#define SX_DEV __device__ __forceinline__

// Conditions that are nearly always the same on all 64 lanes (rain or no rain, day or night, a data gap) can be tested per WAVEFRONT:
// a ballot and a scalar branch.  It pays where a body is a nest of per-lane branches (sx_tanhf's straight-line path, sx_math.h; the still
// steps below) and does NOT pay for the single-level branches of the operators here (measured in round 3, DESIGN.md 12: the selects
// cost registers), which keep the per-lane form.
SX_DEV bool sx_wave_all(bool pred) { return __builtin_amdgcn_ballot_w64(!pred) == 0ull; }   // over the active lanes
SX_DEV bool sx_wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }

struct SxCellParams {   // time-invariant per cell, hoisted out of the time loop
    float ci, cp, inv_cp, cft, cst, exc;
    float cft_m4, cst_m4;   // powf(ct, -4)
    SxDiv dci, dcft, dcst;  // exact division by ci / cft / cst (sx_div)
};

SX_DEV void sx_cell_params_init(SxCellParams& P) {
    P.inv_cp = 1.f / P.cp;
    P.dci = sx_mkdiv(P.ci);
    P.dcft = sx_mkdiv(P.cft);
    P.dcst = sx_mkdiv(P.cst);
}

// everything gr_production computes, kept for the adjoint so tanh / the two divisions are evaluated once
struct SxProd { float thp, the, ps, es, hp_imd, pwr1, pw125, pr, perc, hp_new; };

// ---------------------------------------------------------------- forward
SX_DEV void sx_interception(float prcp, float pet, float ci, const SxDiv& dci, float& hi, float& pn, float& ei) {
    ei = fminf(pet, prcp + hi * ci);
    pn = fmaxf(0.f, prcp - ci * (1.f - hi) - ei);
    hi = hi + sx_div(prcp - ei - pn, dci);
}

template <bool ADJ>
SX_DEV SxProd sx_production_full(float pn, float en, float cp, float inv_cp, float hp) {
    const SxDiv dbeta = {1000.f, 1.0f / 1000.f};   // beta = 1000 at every call site
    SxProd R;
    // tanh(0) = 0 exactly, and at most one of pn, en is non-zero in practice: skip the dead evaluation; with tanh = 0 the quotient
    // is (+-0)/1: skip the division too (dry steps: 90 %; nights: 45 %)
    R.thp = 0.f; R.the = 0.f; R.ps = 0.f; R.es = 0.f;
    const bool wp = pn > 0.f, we = en > 0.f;
    if (wp) {
        const float t = sx_tanhf(pn * inv_cp);
        const float q = sx_fdiv(cp * (1.f - hp * hp) * t, 1.f + hp * t);
        R.thp = wp ? t : 0.f; R.ps = wp ? q : 0.f;
    }
    if (we) {
        const float t = sx_tanhf(en * inv_cp);
        const float q = sx_fdiv((hp * cp) * (2.f - hp) * t, 1.f + (1.f - hp) * t);
        R.the = we ? t : 0.f; R.es = we ? q : 0.f;
    }
    R.hp_imd = hp + (R.ps - R.es) * inv_cp;
    R.pr = wp ? pn - (R.hp_imd - hp) * cp : 0.f;
    // |hp_imd| < 15  =>  (hp_imd / 1000)^4 < 2^-24  =>  1 + r^4 rounds to exactly 1 (beta = 1000): the usual case needs
    // neither the division nor the power
    R.pwr1 = 1.f; R.pw125 = 1.f;
    const bool big = !(fabsf(R.hp_imd) < 15.f);
    if (big) {
        if (big) {
            const float r = sx_div(R.hp_imd, dbeta);
            const float r2 = r * r;
            const float pwx1 = 1.f + r2 * r2;
            if (pwx1 != 1.f) {
                if (ADJ) sx_pow_m025_m125(pwx1, &R.pwr1, &R.pw125);
                else R.pwr1 = sx_pow_m025(pwx1);
            }
        }
    }
    R.perc = (R.hp_imd * cp) * (1.f - R.pwr1);
    R.hp_new = R.hp_imd - R.perc * inv_cp;
    return R;
}

SX_DEV void sx_production(float pn, float en, float cp, float inv_cp, float& hp, float& pr, float& perc) {
    const SxProd R = sx_production_full<false>(pn, en, cp, inv_cp, hp);
    pr = R.pr; perc = R.perc; hp = R.hp_new;
}

SX_DEV void sx_transfer(float prcp, float pr, float ct, const SxDiv& dct, float ct_m4, float& ht, float& q) {
    float pr_imd = pr;
    const bool gap = prcp < 0.f;
    if (gap) {   // data gap: closed-form inverse (md_gr_operator.f90:94-96)
        if (gap) pr_imd = sx_pow_m025(sx_pow_m4(ht * ct) - ct_m4) - (ht * ct);
    }
    const float ht_imd = fmaxf(1.e-6f, ht + sx_div(pr_imd, dct));
    ht = sx_div(sx_pow_m025(sx_pow_m4(ht_imd * ct) + ct_m4), dct);
    q = (ht_imd - ht) * ct;
}

// one vertical cell-step: everything of md_forward_structure.f90:94-151 (gr-a), 280-330 (gr-b),
// 466-517 (gr-c), 655-690 (gr-d) that precedes the routing module.  Returns qt.
template <int ST>
SX_DEV float sx_vertical_step(const SxCellParams& P, float prcp, float pet, float& hi, float& hp, float& hft, float& hst, bool still = false) {
    float ei = 0.f, pn = 0.f, en = 0.f, pr = 0.f, perc = 0.f, l = 0.f;
    if (still) {       // wave-uniform: every lane in a still step or in a data gap (see sx_is_still below)
        const bool gap = !(prcp >= 0.f && pet >= 0.f);          // a gap lane does what the general step does for it: nothing here
        if (ST == 2 || ST == 3) hi = gap ? hi : hi + 0.f;       // hi + (prcp - ei - pn) / ci
        hp = gap ? hp : hp + 0.f;                     // hp_imd = hp + (ps - es) / cp;  perc = (hp_imd cp) (1 - 1) = +-0;  hp = hp_imd - perc / cp
        if (ST != 4) { const float lw = P.exc * sx_pow_3p5(hft); l = gap ? 0.f : lw; }     // pr + perc = +0
    } else if (prcp >= 0.f && pet >= 0.f) {
        if (ST == 1 || ST == 4) {
            ei = fminf(pet, prcp);
            pn = fmaxf(0.f, prcp - ei);
        } else {
            sx_interception(prcp, pet, P.ci, P.dci, hi, pn, ei);
        }
        en = pet - ei;
        sx_production(pn, en, P.cp, P.inv_cp, hp, pr, perc);
        if (ST != 4) l = P.exc * sx_pow_3p5(hft);
    }
    float qr, ql, qd, qt;
    if (ST == 1 || ST == 2) {
        const float prr = 0.9f * (pr + perc) + l;
        const float prd = 0.1f * (pr + perc);
        sx_transfer(prcp, prr, P.cft, P.dcft, P.cft_m4, hft, qr);
        qd = fmaxf(0.f, prd + l);
        qt = (qr + qd);
    } else if (ST == 3) {
        const float prr = 0.9f * 0.6f * (pr + perc) + l;
        const float prl = 0.9f * 0.4f * (pr + perc);
        const float prd = 0.1f * (pr + perc);
        sx_transfer(prcp, prr, P.cft, P.dcft, P.cft_m4, hft, qr);
        sx_transfer(prcp, prl, P.cst, P.dcst, P.cst_m4, hst, ql);
        qd = fmaxf(0.f, prd + l);
        qt = (qr + ql + qd);
    } else {
        const float prr = pr + perc;
        sx_transfer(prcp, prr, P.cft, P.dcft, P.cft_m4, hft, qr);
        qt = qr;
    }
    return qt;
}

// ---------------------------------------------------------------- "still" steps
// A step with no rain and no evaporative demand (prcp = pet = 0) leaves the interception and production stores where they are:
// every night hour of the reference's own hourly PET disaggregation is one (RATIO_PET_HOURLY is zero for 12 of 24 hours,
// core/_constant.py:47-75), rain permitting.  When ALL lanes of a wavefront are in that state the step below replaces the
// general one -- a scalar branch, nothing is predicated -- and is the general step with the zeros carried through by hand
// (each line says which general expression it is).  Same bits: every dropped operation is an addition of +-0 to, or a product
// with 0 of, a finite value (only the sign of a zero gradient, or a NaN that the general step would have spread from an
// already infinite adjoint, can differ).
//   needs, per lane: prcp == 0, pet == 0; 0 <= hi <= 1 (then ei = min(0, hi ci) = 0 and pn = max(0, -ci (1 - hi)) = 0, no branch
//   of GR_INTERCEPTION_B is taken); |hp| < 15 (the percolation power is exactly 1, sx_production_full).
// Lanes in a data gap (prcp < 0 or pet < 0, md_forward_structure.f90:106) do not keep a wavefront from the short form: the general
// step skips interception, production and exchange for them, and so does the short one, lane by lane (a gap every thousand cell-steps,
// as in the synthetic forcing, otherwise sends 6 % of the wavefront-steps -- and a quarter of them would have been still -- the long way).
template <int ST>
SX_DEV bool sx_is_still(float prcp, float pet, float hi, float hp) {
    bool s = (prcp == 0.f) & (pet == 0.f) & (fabsf(hp) < 15.f);
    if (ST == 2 || ST == 3) s = s & (hi >= 0.f) & (hi <= 1.f);
    return s | !(prcp >= 0.f && pet >= 0.f);
}

// ---------------------------------------------------------------- adjoint
SX_DEV void sx_interception_b(float prcp, float pet, float ci, const SxDiv& dci, float& ci_b, float hi, float& hi_b, float& pn_b, float& ei_b) {
    float ei, pn;
    bool br_ei, br_pn;
    if (pet > prcp + hi * ci) { ei = prcp + hi * ci; br_ei = true; } else { ei = pet; br_ei = false; }
    if (0.f < prcp - ci * (1.f - hi) - ei) { pn = prcp - ci * (1.f - hi) - ei; br_pn = true; } else { pn = 0.f; br_pn = false; }
    const float temp_b = sx_div(hi_b, dci);
    ei_b = ei_b - temp_b;
    pn_b = pn_b - temp_b;
    ci_b = ci_b - sx_div((prcp - ei - pn) * temp_b, dci);
    if (br_pn) {
        ci_b = ci_b - (1.f - hi) * pn_b;
        hi_b = hi_b + ci * pn_b;
        ei_b = ei_b - pn_b;
    }
    if (br_ei) {
        hi_b = hi_b + ci * ei_b;
        ci_b = ci_b + hi * ei_b;
    }
}

// hp = pre-step level; R = sx_production_full<true> of this step; pr_b, perc_b in; pn_b, en_b out; hp_b, cp_b updated
SX_DEV void sx_production_b(const SxProd& R, float pn, float& pn_b, float en, float& en_b, float cp, float inv_cp,
                            const SxDiv& dcp2, float& cp_b, float hp, float& hp_b, float pr_b, float perc_b) {
    const SxDiv db4 = {1.0e12f, 1.0f / 1.0e12f};   // beta**4, beta = 1000
    const float thp = R.thp, the = R.the, ps = R.ps, es = R.es, hp_imd = R.hp_imd, pwr1 = R.pwr1, perc = R.perc;
    perc_b = perc_b - inv_cp * hp_b;
    float inv_cp_b = -(perc * hp_b);
    cp_b = cp_b + hp_imd * (1.f - pwr1) * perc_b;
    const float pwr1_b = -(hp_imd * cp * perc_b);
    const float pwx1_b = -(0.25f * R.pw125 * pwr1_b);
    float hp_imd_b = hp_b + cp * (1.f - pwr1) * perc_b + sx_div(4.f * (hp_imd * hp_imd * hp_imd) * pwx1_b, db4);
    const bool wp = pn > 0.f;
    hp_b = 0.f;
    pn_b = 0.f;
    if (wp) {
        if (wp) {
            pn_b = pr_b;
            hp_imd_b = hp_imd_b - cp * pr_b;
            hp_b = cp * pr_b;
            cp_b = cp_b - (hp_imd - hp) * pr_b;
        }
    }
    const float es_b = -(inv_cp * hp_imd_b);
    const float ps_b = inv_cp * hp_imd_b;
    float temp0 = hp * cp * (-hp + 2.f);
    float temp_b, temp_b0, temp_b4, temp_b5;
    const bool we = en > 0.f;
    if (we) {
        // day: the general form on every lane, the lane's own test selects (a lane with en = 0 has the = 0: same values as below up
        // to the sign of a zero, but the select keeps it to the letter)
        const float temp4 = the, temp1 = the;
        const SxDiv d3 = sx_mkdiv_fast((-hp + 1.f) * temp4 + 1.f);
        const float temp_b3 = sx_div(es_b, d3);
        const float g_temp_b = (2.f - hp) * temp1 * temp_b3;
        const float g_temp_b0 = -sx_div(temp0 * temp1 * temp_b3, d3);
        const float g_hp_b = hp_b + hp_imd_b + cp * g_temp_b - hp * cp * temp1 * temp_b3 - temp4 * g_temp_b0;
        const float g_temp_b4 = (1.0f - the * the) * temp0 * temp_b3;
        const float g_temp_b5 = (1.0f - the * the) * (1.f - hp) * g_temp_b0;
        temp_b4 = we ? g_temp_b4 : temp0 * es_b;
        temp_b5 = we ? g_temp_b5 : 0.f;
        hp_b = we ? g_hp_b : hp_b + hp_imd_b;
        en_b = we ? inv_cp * g_temp_b5 + inv_cp * g_temp_b4 : inv_cp * temp_b4;
        cp_b = we ? cp_b + hp * g_temp_b : cp_b;
    } else {
        // tanh(en/cp) = 0: the general expressions reduce to these exactly (every dropped term is +-0)
        temp_b4 = temp0 * es_b;
        temp_b5 = 0.f;
        hp_b = hp_b + hp_imd_b;
        en_b = inv_cp * temp_b4;
    }
    float temp_b2;
    const float temp2 = cp * (-(hp * hp) + 1.f);
    if (wp) {
        if (wp) {
            const float temp = thp, temp1 = thp;
            const SxDiv d0 = sx_mkdiv_fast(hp * temp + 1.f);
            temp_b = sx_div(ps_b, d0);
            temp_b0 = (1.0f - thp * thp) * temp2 * temp_b;
            const float temp_b1 = -sx_div(temp2 * temp1 * temp_b, d0);
            hp_b = hp_b + temp * temp_b1 - 2.f * hp * cp * temp1 * temp_b;
            temp_b2 = (1.0f - thp * thp) * hp * temp_b1;
            inv_cp_b = inv_cp_b + (ps - es) * hp_imd_b + en * temp_b5 + en * temp_b4 + pn * temp_b2 + pn * temp_b0;
            cp_b = cp_b + (1.f - hp * hp) * temp1 * temp_b - sx_div(inv_cp_b, dcp2);
        } else {
            // tanh(pn/cp) = 0 and pn = 0
            temp_b0 = temp2 * ps_b;
            temp_b2 = 0.f;
            inv_cp_b = inv_cp_b + (ps - es) * hp_imd_b + en * temp_b5 + en * temp_b4;
            cp_b = cp_b - sx_div(inv_cp_b, dcp2);
        }
    } else {
        temp_b0 = temp2 * ps_b;
        temp_b2 = 0.f;
        inv_cp_b = inv_cp_b + (ps - es) * hp_imd_b + en * temp_b5 + en * temp_b4;
        cp_b = cp_b - sx_div(inv_cp_b, dcp2);
    }
    pn_b = pn_b + inv_cp * temp_b2 + inv_cp * temp_b0;
}

// ht = pre-step level
SX_DEV void sx_transfer_b(float prcp, float pr, float& pr_b, float ct, const SxDiv& dct, const SxDiv& dct2, float ct_m4,
                          float ct_m5, float& ct_b, float ht, float& ht_b, float q_b) {
    float pr_imd = pr, g_pwx1 = 0.f, g_pwx3 = 0.f;
    const bool gap = prcp < 0.f;
    const bool any_gap = gap;
    if (any_gap) {
        if (gap) {
            g_pwx1 = ht * ct;
            g_pwx3 = sx_pow_m4(g_pwx1) - ct_m4;
            pr_imd = sx_pow_m025(g_pwx3) - ht * ct;
        }
    }
    const float ht_try = ht + sx_div(pr_imd, dct);
    const bool br_max = 1.e-6f < ht_try;
    const float ht_imd = br_max ? ht_try : 1.e-6f;
    const float pwx1 = ht_imd * ct;
    float pwr1, pwx1_m5;
    sx_pow_m4_m5(pwx1, &pwr1, &pwx1_m5);
    const float pwx3 = pwr1 + ct_m4;
    float pwr3, pwx3_m125;
    sx_pow_m025_m125(pwx3, &pwr3, &pwx3_m125);
    const float ht_new = sx_div(pwr3, dct);
    float htb = ht_b - ct * q_b;
    float pwr3_b = sx_div(htb, dct);
    float pwx3_b = -0.25f * pwx3_m125 * pwr3_b;      // pwy3*pwx3**(pwy3-1)*pwr3_b, pwy3 = -1/4 (pwx3 > 0 here)
    float pwr1_b = pwx3_b, pwr2_b = pwx3_b;
    float pwx1_b = -4.f * pwx1_m5 * pwr1_b;           // pwy1*pwx1**(pwy1-1)*pwr1_b, pwy1 = -4
    const float ht_imd_b = ct * q_b + ct * pwx1_b;
    ct_b = ct_b + (ht_imd - ht_new) * q_b + -4.f * ct_m5 * pwr2_b - sx_div(pwr3 * htb, dct2) + ht_imd * pwx1_b;
    // (1.e-6 < ht_try nearly always: both forms computed, the lane's own test selects)
    const float pr_imd_b1 = sx_div(ht_imd_b, dct), ct_b1 = ct_b - sx_div(pr_imd * ht_imd_b, dct2);
    htb = br_max ? ht_imd_b : 0.f;
    const float pr_imd_b = br_max ? pr_imd_b1 : 0.f;
    ct_b = br_max ? ct_b1 : ct_b;
    pr_b = gap ? 0.f : pr_imd_b;
    if (any_gap) {
        if (gap) {
            pwr3_b = pr_imd_b;
            // Tapenade guards pwx3 <= 0 with a non-integer exponent (forward_db.f90:6391-6395).  The two powers go through the same
            // fixed-exponent helpers as everywhere else: the general sx_powf would park ~24 fp64 polynomial constants (48 registers) in
            // the kernel for a branch that runs on 0.1 % of the steps, and cost the whole kernel a wave of occupancy
            float g_m025, g_m125 = 0.f, g_m4, g_m5;
            if (g_pwx3 > 0.f) sx_pow_m025_m125(g_pwx3, &g_m025, &g_m125);
            sx_pow_m4_m5(g_pwx1, &g_m4, &g_m5);
            pwx3_b = (g_pwx3 <= 0.f) ? 0.f : -0.25f * g_m125 * pwr3_b;
            pwr1_b = pwx3_b;
            pwr2_b = -pwx3_b;
            pwx1_b = -4.f * g_m5 * pwr1_b;
            htb = htb + ct * pwx1_b - ct * pr_imd_b;
            ct_b = ct_b + -4.f * ct_m5 * pwr2_b - ht * pr_imd_b + ht * pwx1_b;
        }
    }
    ht_b = htb;
}

struct SxCellGrads {   // running sums, one per cell, accumulated in reverse time order like the reference
    float ci_b, cp_b, cft_b, cst_b, exc_b;
    float hi_b, hp_b, hft_b, hst_b;
};

// reverse of sx_vertical_step given the pre-step states and the incoming qt_b
// (GR_{A,B,C,D}_FORWARD_B inner body: forward_db.f90:8128-8170 / 8674-8716 / 9233-9286 / 9762-9791)
struct SxAdjParams {   // extra per-cell invariants of the adjoint
    float cft_m5, cst_m5;          // powf(ct, -5)
    SxDiv dcft2, dcst2, dcp2;      // exact division by cft**2, cst**2, cp**2
};

// still (wave-uniform): every lane is in a still step (sx_is_still) -- prcp = pet = 0 then, pr + perc = +0 and of
// GR_PRODUCTION_B only the percolation term reaches hp_b (pn_b and en_b are read by nothing: GR_INTERCEPTION_B takes neither
// branch and its quotients multiply prcp - ei - pn = 0); ci_b, cp_b, hi_b stay.
template <int ST>
SX_DEV void sx_vertical_step_b(const SxCellParams& P, const SxAdjParams& Q, float prcp, float pet, float hi, float hp,
                               float hft, float hst, float qt_b, SxCellGrads& G, bool still = false) {
    const bool wet = prcp >= 0.f && pet >= 0.f;         // (per lane, also inside a still wavefront: its gap lanes take the dry path)
    float ei = 0.f, pn = 0.f, en = 0.f, pr = 0.f, perc = 0.f, l = 0.f, prr, prl = 0.f, prd = 0.f;
    float h35 = 0.f, h25 = 0.f;
    SxProd R;
    if (wet) {
        if (!still) {
            float hi2 = hi;
            if (ST == 1 || ST == 4) { ei = fminf(pet, prcp); pn = fmaxf(0.f, prcp - ei); }
            else sx_interception(prcp, pet, P.ci, P.dci, hi2, pn, ei);
            en = pet - ei;
            R = sx_production_full<true>(pn, en, P.cp, P.inv_cp, hp);
            pr = R.pr; perc = R.perc;
        }
        if (ST != 4) { sx_pow_3p5_2p5(hft, &h35, &h25); l = P.exc * h35; }
    }
    if (ST == 1 || ST == 2) { prr = 0.9f * (pr + perc) + l; prd = 0.1f * (pr + perc); }
    else if (ST == 3) { prr = 0.9f * 0.6f * (pr + perc) + l; prl = 0.9f * 0.4f * (pr + perc); prd = 0.1f * (pr + perc); }
    else prr = pr + perc;

    const float qr_b = qt_b, ql_b = qt_b, qd_b = qt_b;
    float prd_b = 0.f, l_b = 0.f, prr_b = 0.f, prl_b = 0.f, pr_b, perc_b;
    if (ST != 4) {
        if (0.f < prd + l) { prd_b = qd_b; l_b = qd_b; }
    }
    if (ST == 3) sx_transfer_b(prcp, prl, prl_b, P.cst, P.dcst, Q.dcst2, P.cst_m4, Q.cst_m5, G.cst_b, hst, G.hst_b, ql_b);
    sx_transfer_b(prcp, prr, prr_b, P.cft, P.dcft, Q.dcft2, P.cft_m4, Q.cft_m5, G.cft_b, hft, G.hft_b, qr_b);
    if (ST == 1 || ST == 2) {
        pr_b = 0.1f * prd_b + 0.9f * prr_b;
        perc_b = 0.1f * prd_b + 0.9f * prr_b;
        l_b = l_b + prr_b;
    } else if (ST == 3) {
        float tb = 0.4f * 0.9f * prl_b;
        pr_b = 0.1f * prd_b + tb;
        perc_b = 0.1f * prd_b + tb;
        tb = 0.6f * 0.9f * prr_b;
        l_b = l_b + prr_b;
        pr_b = pr_b + tb;
        perc_b = perc_b + tb;
    } else {
        pr_b = prr_b;
        perc_b = prr_b;
    }
    if (wet) {
        float pn_b = 0.f, en_b = 0.f;
        if (ST != 4) {   // GR_EXCHANGE_B
            G.exc_b = G.exc_b + h35 * l_b;
            G.hft_b = G.hft_b + 3.5f * h25 * P.exc * l_b;
        }
        if (still) {
            const SxDiv db4 = {1.0e12f, 1.0f / 1.0e12f};
            const float hp_imd = hp + 0.f;
            perc_b = perc_b - P.inv_cp * G.hp_b;
            const float pwr1_b = -(hp_imd * P.cp * perc_b);
            const float pwx1_b = -(0.25f * pwr1_b);       // pw125 = 1
            // hp_b + cp (1 - pwr1) perc_b + 4 hp_imd^3 pwx1_b / beta^4, then the "hp_b = 0; hp_b = hp_b + hp_imd_b" of the dry branches
            G.hp_b = 0.f + (G.hp_b + (P.cp * 0.f) * perc_b + sx_div(4.f * (hp_imd * hp_imd * hp_imd) * pwx1_b, db4));
            return;
        }
        sx_production_b(R, pn, pn_b, en, en_b, P.cp, P.inv_cp, Q.dcp2, G.cp_b, hp, G.hp_b, pr_b, perc_b);
        if (ST == 2 || ST == 3) {
            float ei_b = -en_b;
            sx_interception_b(prcp, pet, P.ci, P.dci, G.ci_b, hi, G.hi_b, pn_b, ei_b);
        }
    }
}
