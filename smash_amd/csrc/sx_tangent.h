// sx_tangent.h -- tangent-linear twins of the vertical operators (gfx950), for base_forward_d
// (reference forward_db.f90:10517-10601; GR_INTERCEPTION_D :5836, GR_PRODUCTION_D :5951, GR_EXCHANGE_D :6111,
// GR_TRANSFER_D :6166, inner body of GR_x_FORWARD_D :7748-9602).
//
// The primal values are produced by the very functions the forward and adjoint kernels use (sx_ops.h), so the
// discharge of a tangent sweep is bit-identical to a forward sweep; the tangents are the Tapenade expressions in
// plain fp32 (divisions by per-cell invariants through sx_div).  The Tapenade tangent code itself re-associates
// a few primal expressions, so bit-identity with the reference's forward_d is not defined: parity is 1e-6 /
// noise-aware on qsim_d and cost_d (tests/test_gpu_tangent.py).
#pragma once

#include "sx_ops.h"

struct SxDual { float v, d; };
SX_DEV SxDual sx_mk(float v, float d) { SxDual x; x.v = v; x.d = d; return x; }

struct SxTanParams {   // tangents of the per-cell parameters
    float ci_d, cp_d, cft_d, cst_d, exc_d;
};

SX_DEV void sx_interception_d(float prcp, float pet, float ci, float ci_d, const SxDiv& dci, SxDual& hi, SxDual& pn, SxDual& ei) {
    const float hv = hi.v;
    if (pet > prcp + hv * ci) ei.d = ci * hi.d + hv * ci_d; else ei.d = 0.f;
    if (0.f < prcp - ci * (1.f - hv) - fminf(pet, prcp + hv * ci)) pn.d = ci * hi.d - (1.f - hv) * ci_d - ei.d; else pn.d = 0.f;
    float h2 = hv;
    sx_interception(prcp, pet, ci, dci, h2, pn.v, ei.v);
    const float temp = sx_div(prcp - ei.v - pn.v, dci);
    hi.d = hi.d + sx_div(-ei.d - pn.d - temp * ci_d, dci);
    hi.v = h2;
}

SX_DEV void sx_production_d(SxDual pn, SxDual en, float cp, float cp_d, float inv_cp, const SxDiv& dcp2, SxDual& hp, SxDual& pr, SxDual& perc) {
    const SxProd R = sx_production_full<true>(pn.v, en.v, cp, inv_cp, hp.v);
    const float h = hp.v, h_d = hp.d;
    const float inv_cp_d = -sx_div(cp_d, dcp2);
    const float tp = R.thp, te = R.the;
    float ps_d = 0.f, es_d = 0.f;
    {
        const float temp1 = cp * (-(h * h) + 1.f);
        const float x_d = inv_cp * pn.d + pn.v * inv_cp_d;
        const float den = h * tp + 1.f;
        ps_d = sx_fdiv(tp * ((1.f - h * h) * cp_d - cp * 2.f * h * h_d) + temp1 * (1.0f - tp * tp) * x_d -
                       R.ps * (tp * h_d + h * (1.0f - tp * tp) * x_d), den);
    }
    {
        const float temp0 = h * cp * (-h + 2.f);
        const float x_d = inv_cp * en.d + en.v * inv_cp_d;
        const float den = (1.f - h) * te + 1.f;
        es_d = sx_fdiv(te * ((2.f - h) * (cp * h_d + h * cp_d) - h * cp * h_d) + temp0 * (1.0f - te * te) * x_d -
                       R.es * ((1.f - h) * (1.0f - te * te) * x_d - te * h_d), den);
    }
    const float hp_imd_d = h_d + inv_cp * (ps_d - es_d) + (R.ps - R.es) * inv_cp_d;
    if (pn.v > 0.f) pr.d = pn.d - cp * (hp_imd_d - h_d) - (R.hp_imd - h) * cp_d; else pr.d = 0.f;
    pr.v = R.pr;
    // pwx1 = 1 + (hp_imd / 1000)^4
    const float pwx1_d = 4.f * (R.hp_imd * R.hp_imd * R.hp_imd) * hp_imd_d * 1.0e-12f;
    const float pwr1_d = -(0.25f * R.pw125 * pwx1_d);
    perc.d = (1.f - R.pwr1) * (cp * hp_imd_d + R.hp_imd * cp_d) - R.hp_imd * cp * pwr1_d;
    perc.v = R.perc;
    hp.d = hp_imd_d - inv_cp * perc.d - R.perc * inv_cp_d;
    hp.v = R.hp_new;
}

// n = 5 (nm1 = 4, 1/nm1 = 0.25).  ct, ct_d and the hoisted powers of ct as in sx_transfer / sx_transfer_b.
SX_DEV void sx_transfer_d(float prcp, SxDual pr, float ct, float ct_d, const SxDiv& dct, float ct_m4, float ct_m5, SxDual& ht, SxDual& q) {
    SxDual pr_imd;
    if (prcp < 0.f) {   // data gap (md_gr_operator.f90:94-96)
        const float pwx1 = ht.v * ct, pwx1_d = ct * ht.d + ht.v * ct_d;
        float p4, p5; sx_pow_m4_m5(pwx1, &p4, &p5);
        const float pwr1_d = -4.f * p5 * pwx1_d;
        const float pwr2_d = -4.f * ct_m5 * ct_d;
        const float pwx3 = p4 - ct_m4, pwx3_d = pwr1_d - pwr2_d;
        float r, r125 = 0.f;
        float pwr3_d = 0.f;
        if (pwx3 > 0.f) { sx_pow_m025_m125(pwx3, &r, &r125); pwr3_d = -0.25f * r125 * pwx3_d; }
        else r = sx_pow_m025(pwx3);
        pr_imd.d = pwr3_d - ct * ht.d - ht.v * ct_d;
        pr_imd.v = r - (ht.v * ct);
    } else pr_imd = pr;
    SxDual ht_imd;
    const float ht_try = ht.v + sx_div(pr_imd.v, dct);
    if (1.e-6f < ht_try) { ht_imd.d = ht.d + sx_div(pr_imd.d - sx_div(pr_imd.v * ct_d, dct), dct); ht_imd.v = ht_try; }
    else { ht_imd.v = 1.e-6f; ht_imd.d = 0.f; }
    const float pwx1 = ht_imd.v * ct, pwx1_d = ct * ht_imd.d + ht_imd.v * ct_d;
    float pwr1, pwx1_m5;
    sx_pow_m4_m5(pwx1, &pwr1, &pwx1_m5);
    const float pwr1_d = -4.f * pwx1_m5 * pwx1_d;
    const float pwr2_d = -4.f * ct_m5 * ct_d;
    const float pwx3 = pwr1 + ct_m4, pwx3_d = pwr1_d + pwr2_d;
    float pwr3, pwx3_m125;
    sx_pow_m025_m125(pwx3, &pwr3, &pwx3_m125);
    const float pwr3_d = -0.25f * pwx3_m125 * pwx3_d;
    const float ht_new = sx_div(pwr3, dct);
    const float ht_new_d = sx_div(pwr3_d - sx_div(pwr3 * ct_d, dct), dct);
    q.d = ct * (ht_imd.d - ht_new_d) + (ht_imd.v - ht_new) * ct_d;
    q.v = (ht_imd.v - ht_new) * ct;
    ht.d = ht_new_d; ht.v = ht_new;
}

// one vertical cell-step with tangents: returns qt (value and tangent)
template <int ST>
SX_DEV SxDual sx_vertical_step_d(const SxCellParams& P, const SxAdjParams& Q, const SxTanParams& D, float prcp, float pet,
                                 SxDual& hi, SxDual& hp, SxDual& hft, SxDual& hst) {
    SxDual ei = sx_mk(0.f, 0.f), pn = ei, en = ei, pr = ei, perc = ei, l = ei;
    if (prcp >= 0.f && pet >= 0.f) {
        if (ST == 1 || ST == 4) { ei.v = fminf(pet, prcp); pn.v = fmaxf(0.f, prcp - ei.v); }
        else sx_interception_d(prcp, pet, P.ci, D.ci_d, P.dci, hi, pn, ei);
        en.v = pet - ei.v; en.d = -ei.d;
        sx_production_d(pn, en, P.cp, D.cp_d, P.inv_cp, Q.dcp2, hp, pr, perc);
        if (ST != 4) {
            float h35, h25;
            sx_pow_3p5_2p5(hft.v, &h35, &h25);
            l.d = h35 * D.exc_d + P.exc * 3.5f * h25 * hft.d;
            l.v = P.exc * h35;
        }
    }
    SxDual qr, ql = sx_mk(0.f, 0.f), qd = sx_mk(0.f, 0.f), qt;
    if (ST == 1 || ST == 2) {
        const SxDual prr = sx_mk(0.9f * (pr.v + perc.v) + l.v, 0.9f * (pr.d + perc.d) + l.d);
        const SxDual prd = sx_mk(0.1f * (pr.v + perc.v), 0.1f * (pr.d + perc.d));
        sx_transfer_d(prcp, prr, P.cft, D.cft_d, P.dcft, P.cft_m4, Q.cft_m5, hft, qr);
        qd.d = (0.f < prd.v + l.v) ? prd.d + l.d : 0.f;
        qd.v = fmaxf(0.f, prd.v + l.v);
        qt.v = (qr.v + qd.v); qt.d = qr.d + qd.d;
    } else if (ST == 3) {
        const SxDual prr = sx_mk(0.9f * 0.6f * (pr.v + perc.v) + l.v, 0.9f * 0.6f * (pr.d + perc.d) + l.d);
        const SxDual prl = sx_mk(0.9f * 0.4f * (pr.v + perc.v), 0.9f * 0.4f * (pr.d + perc.d));
        const SxDual prd = sx_mk(0.1f * (pr.v + perc.v), 0.1f * (pr.d + perc.d));
        sx_transfer_d(prcp, prr, P.cft, D.cft_d, P.dcft, P.cft_m4, Q.cft_m5, hft, qr);
        sx_transfer_d(prcp, prl, P.cst, D.cst_d, P.dcst, P.cst_m4, Q.cst_m5, hst, ql);
        qd.d = (0.f < prd.v + l.v) ? prd.d + l.d : 0.f;
        qd.v = fmaxf(0.f, prd.v + l.v);
        qt.v = (qr.v + ql.v + qd.v); qt.d = qr.d + ql.d + qd.d;
    } else {
        const SxDual prr = sx_mk(pr.v + perc.v, pr.d + perc.d);
        sx_transfer_d(prcp, prr, P.cft, D.cft_d, P.dcft, P.cft_m4, Q.cft_m5, hft, qr);
        qt = qr;
    }
    return qt;
}
