/* smash_oracle_d.c -- tangent-linear model of the hot path (reference: base_forward_d, forward_db.f90:10517-10601).
 *
 * TEST INFRASTRUCTURE ONLY (see smash_oracle.h).  Plain C, single thread, fp32.
 * The primal values follow the Tapenade tangent code, which re-associates two expressions of the original
 * (qup = dt*(qup/temp) and q = temp*((qt + f*qrout)/dt)): the discharge of forward_d differs from forward's in the
 * last bit, and because the criteria derivatives are differences of nearly equal sums that moves cost_d by ~5e-6.
 * The tangents follow the Tapenade tangent routines expression by expression (GR_INTERCEPTION_D :5836-5876, GR_PRODUCTION_D :5951-6007, GR_EXCHANGE_D
 * :6111-6128, GR_TRANSFER_D :6166-6268, UPSTREAM_DISCHARGE_D :6422-6470, LINEAR_ROUTING_D :6574-6600,
 * GR_x_FORWARD_D :7748-9602, COMPUTE_JOBS_D :2445-2551, NSE_D/KGE_D/SE_D/RMSE_D/LOGARITHMIC_D :3400-4200,
 * QUANTILE_D :4279-4322, COMPUTE_JREG_D :2810-2925, REG_PRIOR_D :5720-5750, REG_SMOOTHING_D :5382-5500,
 * (DE)NORMALIZE_*_D :760-1960, COMPUTE_COST_D :3196-3250).
 * Pinned against the reference's own forward_d (oracle/refbind.run(..., params_d=...)) to 1e-6: the Tapenade
 * tangent code re-associates a few primal expressions (e.g. qup = dt*(qup/temp)), so bit-identity is not defined.
 * Reference quirk NOT reproduced: COMPUTE_COST_D ends with an unconditional DENORMALIZE_PARAMETERS/STATES
 * (forward_db.f90:3246-3247), which corrupts parameters and states when denormalize_forward is off; here they
 * come back like base_forward leaves them.  parameters_bgd_d / states_bgd_d are passive in the reference. */
#include "smash_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { P_CI = 0, P_CP = 1, P_CFT = 3, P_CST = 4, P_EXC = 6, P_B = 7, P_CUSL1 = 8, P_CUSL2 = 9, P_CLSL = 10, P_KS = 11, P_DS = 12,
       P_DSM = 13, P_WS = 14, P_LR = 15 };
enum { S_HI = 0, S_HP = 1, S_HFT = 2, S_HST = 3, S_HUSL1 = 4, S_HUSL2 = 5, S_HLSL = 6, S_HLR = 7 };

typedef struct { float v, d; } dual;

static void interception_d(float prcp, float pet, dual ci, dual* hi, dual* pn, dual* ei) {
    if (pet > prcp + hi->v * ci.v) { ei->d = ci.v * hi->d + hi->v * ci.d; ei->v = prcp + hi->v * ci.v; }
    else { ei->v = pet; ei->d = 0.f; }
    if (0.f < prcp - ci.v * (1.f - hi->v) - ei->v) {
        pn->d = ci.v * hi->d - (1.f - hi->v) * ci.d - ei->d;
        pn->v = prcp - ci.v * (1.f - hi->v) - ei->v;
    } else { pn->v = 0.f; pn->d = 0.f; }
    float temp = (prcp - ei->v - pn->v) / ci.v;
    hi->d = hi->d + (-ei->d - pn->d - temp * ci.d) / ci.v;
    hi->v = hi->v + temp;
}

static void production_d(dual pn, dual en, dual cp, float beta, dual* hp, dual* pr, dual* perc) {
    float inv_cp_d = -(cp.d / (cp.v * cp.v));
    float inv_cp = 1.f / cp.v;
    float h = hp->v, h_d = hp->d;
    float tp = tanhf(pn.v * inv_cp), te = tanhf(en.v * inv_cp);
    float temp1 = cp.v * (-(h * h) + 1.f);
    float ps = temp1 * tp / (h * tp + 1.f);
    float xp_d = inv_cp * pn.d + pn.v * inv_cp_d;
    float ps_d = (tp * ((1.f - h * h) * cp.d - cp.v * 2.f * h * h_d) + temp1 * (1.0f - tp * tp) * xp_d -
                  ps * (tp * h_d + h * (1.0f - tp * tp) * xp_d)) / (h * tp + 1.f);
    float temp0 = h * cp.v * (-h + 2.f);
    float es = temp0 * te / ((-h + 1.f) * te + 1.f);
    float xe_d = inv_cp * en.d + en.v * inv_cp_d;
    float es_d = (te * ((2.f - h) * (cp.v * h_d + h * cp.d) - h * cp.v * h_d) + temp0 * (1.0f - te * te) * xe_d -
                  es * ((1.f - h) * (1.0f - te * te) * xe_d - te * h_d)) / ((1.f - h) * te + 1.f);
    /* primal exactly as gr_production (md_gr_operator.f90:36-67) */
    float psv = cp.v * (1.f - h * h) * tp / (1.f + h * tp);
    float esv = (h * cp.v) * (2.f - h) * te / (1.f + (1.f - h) * te);
    float hp_imd_d = h_d + inv_cp * (ps_d - es_d) + (psv - esv) * inv_cp_d;
    float hp_imd = h + (psv - esv) * inv_cp;
    if (pn.v > 0.f) { pr->d = pn.d - cp.v * (hp_imd_d - h_d) - (hp_imd - h) * cp.d; pr->v = pn.v - (hp_imd - h) * cp.v; }
    else { pr->d = 0.f; pr->v = 0.f; }
    float r = hp_imd / beta, r2 = r * r;
    float pwx1_d = 4.f * (hp_imd * hp_imd * hp_imd) * hp_imd_d / (beta * beta * beta * beta);
    float pwx1 = 1.f + r2 * r2;
    float pwr1_d = -(0.25f * powf(pwx1, -1.25f) * pwx1_d);
    float pwr1 = powf(pwx1, -0.25f);
    perc->d = (1.f - pwr1) * (cp.v * hp_imd_d + hp_imd * cp.d) - hp_imd * cp.v * pwr1_d;
    perc->v = (hp_imd * cp.v) * (1.f - pwr1);
    hp->d = hp_imd_d - inv_cp * perc->d - perc->v * inv_cp_d;
    hp->v = hp_imd - perc->v * inv_cp;
}

static dual exchange_d(dual exc, dual hft) {
    dual l;
    float temp = powf(hft.v, 3.5f);
    l.d = temp * exc.d + exc.v * 3.5f * powf(hft.v, 2.5f) * hft.d;
    l.v = exc.v * temp;
    return l;
}

static float pow_d(float x, float y, float x_d) {   /* d(x**y) for a passive exponent, with Tapenade's guard */
    if (x <= 0.f && (y == 0.f || y != (float)(int)y)) return 0.f;
    return y * powf(x, y - 1.f) * x_d;
}

static void transfer_d(float n, float prcp, dual pr, dual ct, dual* ht, dual* q) {
    const float nm1 = n - 1.f, d1pnm1 = 1.f / nm1;
    dual pr_imd;
    if (prcp < 0.f) {
        float pwx1 = ht->v * ct.v, pwx1_d = ct.v * ht->d + ht->v * ct.d;
        float pwr1 = powf(pwx1, -nm1), pwr1_d = pow_d(pwx1, -nm1, pwx1_d);
        float pwr2 = powf(ct.v, -nm1), pwr2_d = pow_d(ct.v, -nm1, ct.d);
        float pwx3 = pwr1 - pwr2, pwx3_d = pwr1_d - pwr2_d;
        float pwr3 = powf(pwx3, -d1pnm1), pwr3_d = pow_d(pwx3, -d1pnm1, pwx3_d);
        pr_imd.d = pwr3_d - ct.v * ht->d - ht->v * ct.d;
        pr_imd.v = pwr3 - ht->v * ct.v;
    } else pr_imd = pr;
    dual ht_imd;
    if (1.e-6f < ht->v + pr_imd.v / ct.v) {
        ht_imd.d = ht->d + (pr_imd.d - pr_imd.v * ct.d / ct.v) / ct.v;
        ht_imd.v = ht->v + pr_imd.v / ct.v;
    } else { ht_imd.v = 1.e-6f; ht_imd.d = 0.f; }
    float pwx1 = ht_imd.v * ct.v, pwx1_d = ct.v * ht_imd.d + ht_imd.v * ct.d;
    float pwr1 = powf(pwx1, -nm1), pwr1_d = pow_d(pwx1, -nm1, pwx1_d);
    float pwr2 = powf(ct.v, -nm1), pwr2_d = pow_d(ct.v, -nm1, ct.d);
    float pwx3 = pwr1 + pwr2, pwx3_d = pwr1_d + pwr2_d;
    float pwr3 = powf(pwx3, -d1pnm1), pwr3_d = pow_d(pwx3, -d1pnm1, pwx3_d);
    float ht_new_d = (pwr3_d - pwr3 * ct.d / ct.v) / ct.v;
    float ht_new = pwr3 / ct.v;
    q->d = ct.v * (ht_imd.d - ht_new_d) + (ht_imd.v - ht_new) * ct.d;
    q->v = (ht_imd.v - ht_new) * ct.v;
    ht->d = ht_new_d; ht->v = ht_new;
}

static dual upstream_d(float dt, float dx, int nrow, int ncol, const int* flwdir, const int* flwacc, int row, int col,
                       const float* q, const float* q_d) {
    static const int dcol[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    static const int drow[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    dual qup = {0.f, 0.f};
    if (flwacc[row + col * nrow] > 1) {
        for (int i = 0; i < 8; ++i) {
            int c = col + dcol[i], r = row + drow[i];
            if (c >= 0 && c < ncol && r >= 0 && r < nrow)
                if (flwdir[r + c * nrow] == i + 1) { qup.d = qup.d + q_d[r + c * nrow]; qup.v = qup.v + q[r + c * nrow]; }
        }
        float temp = 0.001f * (dx * dx) * (float)(flwacc[row + col * nrow] - 1);
        qup.d = dt * qup.d / temp;
        qup.v = dt * (qup.v / temp);   /* UPSTREAM_DISCHARGE_D re-associates the primal (forward_db.f90:6464) */
    }
    return qup;
}

static dual routing_d(float dt, dual qup, dual lr, dual* hr) {
    dual qrout;
    float hr_imd_d = hr->d + qup.d, hr_imd = hr->v + qup.v;
    float temp = dt / (60.f * lr.v);
    float arg1_d = temp * lr.d / lr.v;
    float e = expf(-dt / (lr.v * 60.f));
    hr->d = e * hr_imd_d + hr_imd * e * arg1_d;
    hr->v = hr_imd * e;
    qrout.d = hr_imd_d - hr->d;
    qrout.v = hr_imd - hr->v;
    return qrout;
}

#define FLD(A, f, c) (A)[(f) * n2 + (c)]
static dual mk(float v, float d) { dual x; x.v = v; x.d = d; return x; }

/* ---- vic-a tangents: VIC_INFILTRATION_D :6691-6802, VIC_VERTICAL_TRANSFER_D :7042-7095, VIC_INTERFLOW_D :7235-7295,
 * VIC_BASEFLOW_D :7402-7437, BROOKS_AND_COREY_FLOW_D :7530-7575, LINEAR_EVAPOTRANSPIRATION_D :7675-7695.  The primal
 * follows the _D code, which re-associates vic_baseflow (qb = ds*dsm*(hlsl/ws); (1-ds/ws)*((hlsl-ws)*(dsm/(1-ws)))). */
static float powd_full(float x, float y, float x_d, float y_d, float* r) {   /* d(x**y), both active, Tapenade's three cases */
    float t = powf(x, y);
    *r = t;
    if (x <= 0.f && (y == 0.f || y != (float)(int)y)) return 0.f;
    if (x <= 0.f) return y * powf(x, y - 1.f) * x_d;
    return y * powf(x, y - 1.f) * x_d + t * logf(x) * y_d;
}
static void vic_infiltration_d(float prcp, dual cusl1, dual cusl2, dual b, dual* husl1, dual* husl2, dual* runoff) {
    float bp1_d = b.d, bp1 = b.v + 1.f, ifl, ifl_d;
    if (prcp <= 0.f) { ifl = 0.f; ifl_d = 0.f; }
    else {
        float cusl_d = cusl1.d + cusl2.d, cusl = cusl1.v + cusl2.v;
        float wusl_d = cusl1.v * husl1->d + husl1->v * cusl1.d + cusl2.v * husl2->d + husl2->v * cusl2.d;
        float wusl = husl1->v * cusl1.v + husl2->v * cusl2.v;
        if (!(1.e-6f < wusl)) { wusl = 1.e-6f; wusl_d = 0.f; }
        if (!(cusl - 1e-6f > wusl)) { wusl_d = cusl_d; wusl = cusl - 1e-6f; }
        float iflm_d = bp1 * cusl_d + cusl * bp1_d, iflm = cusl * bp1;
        float pwx1_d = -((wusl_d - wusl * cusl_d / cusl) / cusl), pwx1 = 1.f - wusl / cusl;
        float pwy1_d = -(bp1_d / (bp1 * bp1)), pwy1 = 1.f / bp1;
        float pwr1, pwr1_d = powd_full(pwx1, pwy1, pwx1_d, pwy1_d, &pwr1);
        float iflc_d = (1.f - pwr1) * iflm_d - iflm * pwr1_d, iflc = iflm * (1.f - pwr1);
        if (iflc + prcp >= iflm) { ifl_d = cusl_d - wusl_d; ifl = cusl - wusl; }
        else {
            float temp = (prcp + iflc) / iflm;
            pwx1_d = -((iflc_d - temp * iflm_d) / iflm);
            pwx1 = 1.f - temp;
            pwr1_d = powd_full(pwx1, bp1, pwx1_d, bp1_d, &pwr1);
            ifl_d = (1.0f - pwr1) * cusl_d - wusl_d - cusl * pwr1_d;
            ifl = cusl - wusl - cusl * pwr1;
        }
        if (!(prcp > ifl)) { ifl = prcp; ifl_d = 0.f; }
    }
    float u1, u1_d, u2, u2_d;
    if ((1.f - husl1->v) * cusl1.v > ifl) { u1_d = ifl_d; u1 = ifl; }
    else { u1_d = (1.f - husl1->v) * cusl1.d - cusl1.v * husl1->d; u1 = (1.f - husl1->v) * cusl1.v; }
    ifl_d = ifl_d - u1_d; ifl = ifl - u1;
    if ((1.f - husl2->v) * cusl2.v > ifl) { u2_d = ifl_d; u2 = ifl; }
    else { u2_d = (1.f - husl2->v) * cusl2.d - cusl2.v * husl2->d; u2 = (1.f - husl2->v) * cusl2.v; }
    husl1->d = husl1->d + (u1_d - u1 * cusl1.d / cusl1.v) / cusl1.v;
    husl1->v = husl1->v + u1 / cusl1.v;
    husl2->d = husl2->d + (u2_d - u2 * cusl2.d / cusl2.v) / cusl2.v;
    husl2->v = husl2->v + u2 / cusl2.v;
    runoff->d = -u1_d - u2_d;
    runoff->v = prcp - (u1 + u2);
}
static dual brooks_d(dual ks, dual c_upper, dual c_lower, dual h_upper, dual h_lower) {   /* residual 0, porosity 1, lambda 1 */
    dual flow;
    float pwx1_d = h_upper.d / (1.f - 0.f), pwx1 = (h_upper.v - 0.f) / (1.f - 0.f);
    float pwr1_d = 1.f * powf(pwx1, 0.f) * pwx1_d, pwr1 = powf(pwx1, 1.f);
    flow.d = pwr1 * ks.d + ks.v * pwr1_d;
    flow.v = ks.v * pwr1;
    float w_upper_d = 1.f * (c_upper.v * h_upper.d + h_upper.v * c_upper.d), w_upper = h_upper.v * c_upper.v * 1.f;
    float w_lower_d = 1.f * (c_lower.v * h_lower.d + h_lower.v * c_lower.d), w_lower = h_lower.v * c_lower.v * 1.f;
    float max_flow, max_flow_d;
    if (w_upper > c_lower.v - w_lower) { max_flow_d = c_lower.d - w_lower_d; max_flow = c_lower.v - w_lower; }
    else { max_flow_d = w_upper_d; max_flow = w_upper; }
    if (!(max_flow > flow.v)) { flow.d = max_flow_d; flow.v = max_flow; }
    return flow;
}
static dual evap_d(dual e, dual c, dual h) {
    dual flow;
    flow.d = h.v * e.d + e.v * h.d; flow.v = e.v * h.v;
    float w_d = h.v * c.d + c.v * h.d, w = c.v * h.v;
    if (!(w > flow.v)) { flow.d = w_d; flow.v = w; }
    return flow;
}
static void vic_vertical_transfer_d(float pet, dual cusl1, dual cusl2, dual clsl, dual ks, dual* husl1, dual* husl2, dual* hlsl) {
    dual fbc = brooks_d(ks, cusl1, cusl2, *husl1, *husl2);
    husl1->d = husl1->d - (fbc.d - fbc.v * cusl1.d / cusl1.v) / cusl1.v; husl1->v = husl1->v - fbc.v / cusl1.v;
    husl2->d = husl2->d + (fbc.d - fbc.v * cusl2.d / cusl2.v) / cusl2.v; husl2->v = husl2->v + fbc.v / cusl2.v;
    fbc = brooks_d(ks, cusl2, clsl, *husl2, *hlsl);
    husl2->d = husl2->d - (fbc.d - fbc.v * cusl2.d / cusl2.v) / cusl2.v; husl2->v = husl2->v - fbc.v / cusl2.v;
    hlsl->d = hlsl->d + (fbc.d - fbc.v * clsl.d / clsl.v) / clsl.v; hlsl->v = hlsl->v + fbc.v / clsl.v;
    dual fe = evap_d(mk(pet, 0.f), cusl1, *husl1);
    husl1->d = husl1->d - (fe.d - fe.v * cusl1.d / cusl1.v) / cusl1.v; husl1->v = husl1->v - fe.v / cusl1.v;
    dual pr;
    if (0.f < pet - fe.v) { pr.d = -fe.d; pr.v = pet - fe.v; } else { pr.v = 0.f; pr.d = 0.f; }
    fe = evap_d(pr, cusl2, *husl2);
    husl2->d = husl2->d - (fe.d - fe.v * cusl2.d / cusl2.v) / cusl2.v; husl2->v = husl2->v - fe.v / cusl2.v;
    if (0.f < pr.v - fe.v) { pr.d = pr.d - fe.d; pr.v = pr.v - fe.v; } else { pr.v = 0.f; pr.d = 0.f; }
    fe = evap_d(pr, clsl, *hlsl);
    hlsl->d = hlsl->d - (fe.d - fe.v * clsl.d / clsl.v) / clsl.v; hlsl->v = hlsl->v - fe.v / clsl.v;
}
static void vic_interflow_d(float n, dual cusl2, dual* husl2, dual* qi) {
    const float nm1 = n - 1.f, d1pnm1 = 1.f / nm1;
    dual him = *husl2;
    float pwx1_d = cusl2.v * him.d + him.v * cusl2.d, pwx1 = him.v * cusl2.v;
    float pwr1_d = pow_d(pwx1, -nm1, pwx1_d), pwr1 = powf(pwx1, -nm1);
    float pwr2_d = pow_d(cusl2.v, -nm1, cusl2.d), pwr2 = powf(cusl2.v, -nm1);
    float pwx3_d = pwr1_d + pwr2_d, pwx3 = pwr1 + pwr2;
    float pwr3_d = pow_d(pwx3, -d1pnm1, pwx3_d), pwr3 = powf(pwx3, -d1pnm1);
    husl2->d = (pwr3_d - pwr3 * cusl2.d / cusl2.v) / cusl2.v;
    husl2->v = pwr3 / cusl2.v;
    qi->d = cusl2.v * (him.d - husl2->d) + (him.v - husl2->v) * cusl2.d;
    qi->v = (him.v - husl2->v) * cusl2.v;
}
static void vic_baseflow_d(dual clsl, dual ds, dual dsm, dual ws, dual* hlsl, dual* qb) {
    float q, q_d;
    if (hlsl->v <= ws.v) {
        float temp = hlsl->v / ws.v;
        q_d = temp * (dsm.v * ds.d + ds.v * dsm.d) + ds.v * dsm.v * (hlsl->d - temp * ws.d) / ws.v;
        q = ds.v * dsm.v * temp;
    } else {
        float temp = dsm.v / (-ws.v + 1.f), temp0 = ds.v / ws.v;
        q_d = (1.f - temp0) * (temp * (hlsl->d - ws.d) + (hlsl->v - ws.v) * (dsm.d + temp * ws.d) / (1.f - ws.v)) -
              (hlsl->v - ws.v) * temp * (ds.d - temp0 * ws.d) / ws.v;
        q = (1.f - temp0) * ((hlsl->v - ws.v) * temp);
    }
    float wlsl_d = hlsl->v * clsl.d + clsl.v * hlsl->d, wlsl = clsl.v * hlsl->v;
    if (!(wlsl > q)) { q_d = wlsl_d; q = wlsl; }
    hlsl->d = hlsl->d - (q_d - q * clsl.d / clsl.v) / clsl.v;
    hlsl->v = hlsl->v - q / clsl.v;
    qb->d = q_d; qb->v = q;
}


/* GR_{A,B,C,D}_FORWARD_D: one cell-step */
static void cell_step_d(int st, float dt, float dx, int nrow, int ncol, const int* flwdir, const int* flwacc, int row, int col,
                        float prcp, float pet, const float* P, const float* P_d, float* S, float* S_d, float* q, float* q_d) {
    const long n2 = (long)nrow * ncol, c = row + (long)col * nrow;
    if (st == ORC_VIC_A) {   /* VIC_A_FORWARD_D, forward_db.f90:9949-10095 */
#define DP(f) mk(FLD(P, f, c), FLD(P_d, f, c))
        dual h1 = mk(FLD(S, S_HUSL1, c), FLD(S_d, S_HUSL1, c)), h2 = mk(FLD(S, S_HUSL2, c), FLD(S_d, S_HUSL2, c));
        dual hl = mk(FLD(S, S_HLSL, c), FLD(S_d, S_HLSL, c)), hr = mk(FLD(S, S_HLR, c), FLD(S_d, S_HLR, c));
        dual runoff = {0.f, 0.f}, qi, qb, qtv;
        if (prcp >= 0.f && pet >= 0.f) {
            vic_infiltration_d(prcp, DP(P_CUSL1), DP(P_CUSL2), DP(P_B), &h1, &h2, &runoff);
            vic_vertical_transfer_d(pet, DP(P_CUSL1), DP(P_CUSL2), DP(P_CLSL), DP(P_KS), &h1, &h2, &hl);
        }
        vic_interflow_d(5.f, DP(P_CUSL2), &h2, &qi);
        vic_baseflow_d(DP(P_CLSL), DP(P_DS), DP(P_DSM), DP(P_WS), &hl, &qb);
        qtv.d = runoff.d + qi.d + qb.d; qtv.v = runoff.v + qi.v + qb.v;
        dual qupv = upstream_d(dt, dx, nrow, ncol, flwdir, flwacc, row, col, q, q_d);
        dual qro = routing_d(dt, qupv, DP(P_LR), &hr);
        const float fv = (float)(flwacc[c] - 1), tempv = 0.001f * (dx * dx);
        q_d[c] = tempv * (qtv.d + fv * qro.d) / dt;
        q[c] = tempv * ((qtv.v + fv * qro.v) / dt);
        FLD(S, S_HUSL1, c) = h1.v; FLD(S_d, S_HUSL1, c) = h1.d; FLD(S, S_HUSL2, c) = h2.v; FLD(S_d, S_HUSL2, c) = h2.d;
        FLD(S, S_HLSL, c) = hl.v; FLD(S_d, S_HLSL, c) = hl.d; FLD(S, S_HLR, c) = hr.v; FLD(S_d, S_HLR, c) = hr.d;
#undef DP
        return;
    }
    dual ei = {0, 0}, pn = {0, 0}, en = {0, 0}, pr = {0, 0}, perc = {0, 0}, l = {0, 0}, qr = {0, 0}, ql = {0, 0}, qd, qt;
    dual hi = mk(FLD(S, S_HI, c), FLD(S_d, S_HI, c)), hp = mk(FLD(S, S_HP, c), FLD(S_d, S_HP, c));
    dual hft = mk(FLD(S, S_HFT, c), FLD(S_d, S_HFT, c)), hst = mk(FLD(S, S_HST, c), FLD(S_d, S_HST, c));
    dual hlr = mk(FLD(S, S_HLR, c), FLD(S_d, S_HLR, c));
    dual cp = mk(FLD(P, P_CP, c), FLD(P_d, P_CP, c)), cft = mk(FLD(P, P_CFT, c), FLD(P_d, P_CFT, c));
    if (prcp >= 0.f && pet >= 0.f) {
        if (st == ORC_GR_A || st == ORC_GR_D) {
            ei.v = fminf(pet, prcp); ei.d = 0.f;
            pn.v = fmaxf(0.f, prcp - ei.v); pn.d = 0.f;
        } else {
            interception_d(prcp, pet, mk(FLD(P, P_CI, c), FLD(P_d, P_CI, c)), &hi, &pn, &ei);
        }
        en.v = pet - ei.v; en.d = -ei.d;
        production_d(pn, en, cp, 1000.f, &hp, &pr, &perc);
        if (st != ORC_GR_D) l = exchange_d(mk(FLD(P, P_EXC, c), FLD(P_d, P_EXC, c)), hft);
    }
    dual prr, prl, prd;
    if (st == ORC_GR_A || st == ORC_GR_B) {
        prr.v = 0.9f * (pr.v + perc.v) + l.v; prr.d = 0.9f * (pr.d + perc.d) + l.d;
        prd.v = 0.1f * (pr.v + perc.v); prd.d = 0.1f * (pr.d + perc.d);
        transfer_d(5.f, prcp, prr, cft, &hft, &qr);
        qd.d = (0.f < prd.v + l.v) ? prd.d + l.d : 0.f;
        qd.v = fmaxf(0.f, prd.v + l.v);
        qt.v = (qr.v + qd.v); qt.d = qr.d + qd.d;
    } else if (st == ORC_GR_C) {
        prr.v = 0.9f * 0.6f * (pr.v + perc.v) + l.v; prr.d = 0.9f * 0.6f * (pr.d + perc.d) + l.d;
        prl.v = 0.9f * 0.4f * (pr.v + perc.v); prl.d = 0.9f * 0.4f * (pr.d + perc.d);
        prd.v = 0.1f * (pr.v + perc.v); prd.d = 0.1f * (pr.d + perc.d);
        transfer_d(5.f, prcp, prr, cft, &hft, &qr);
        transfer_d(5.f, prcp, prl, mk(FLD(P, P_CST, c), FLD(P_d, P_CST, c)), &hst, &ql);
        qd.d = (0.f < prd.v + l.v) ? prd.d + l.d : 0.f;
        qd.v = fmaxf(0.f, prd.v + l.v);
        qt.v = (qr.v + ql.v + qd.v); qt.d = qr.d + ql.d + qd.d;
    } else {
        prr.v = pr.v + perc.v; prr.d = pr.d + perc.d;
        transfer_d(5.f, prcp, prr, cft, &hft, &qr);
        qt = qr;
    }
    dual qup = upstream_d(dt, dx, nrow, ncol, flwdir, flwacc, row, col, q, q_d);
    dual qrout = routing_d(dt, qup, mk(FLD(P, P_LR, c), FLD(P_d, P_LR, c)), &hlr);
    const float f = (float)(flwacc[c] - 1);
    const float temp = 0.001f * (dx * dx);   /* GR_x_FORWARD_D: q = temp*((qt + temp0*qrout)/dt), e.g. forward_db.f90:8445-8448 */
    q_d[c] = temp * (qt.d + f * qrout.d) / dt;
    q[c] = temp * ((qt.v + f * qrout.v) / dt);
    FLD(S, S_HI, c) = hi.v; FLD(S_d, S_HI, c) = hi.d; FLD(S, S_HP, c) = hp.v; FLD(S_d, S_HP, c) = hp.d;
    FLD(S, S_HFT, c) = hft.v; FLD(S_d, S_HFT, c) = hft.d; FLD(S, S_HST, c) = hst.v; FLD(S_d, S_HST, c) = hst.d;
    FLD(S, S_HLR, c) = hlr.v; FLD(S_d, S_HLR, c) = hlr.d;
}

/* ---- cost: COMPUTE_JOBS_D and the criteria --------------------------------------------------------------- */
static float nse_d(const float* x, const float* y, const float* y_d, int n_, float* res) {
    int n = 0;
    float sum_x = 0.f, sum_xx = 0.f, sum_yy = 0.f, sum_xy = 0.f, sum_yy_d = 0.f, sum_xy_d = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) {
            n++;
            sum_x = sum_x + x[i]; sum_xx = sum_xx + x[i] * x[i];
            sum_yy_d = sum_yy_d + 2.f * y[i] * y_d[i]; sum_yy = sum_yy + y[i] * y[i];
            sum_xy_d = sum_xy_d + x[i] * y_d[i]; sum_xy = sum_xy + x[i] * y[i];
        }
    float mean_x = sum_x / (float)n;
    float num_d = sum_yy_d - 2.f * sum_xy_d, num = sum_xx - 2.f * sum_xy + sum_yy;
    float den = sum_xx - (float)n * mean_x * mean_x;
    *res = num / den;
    return num_d / den;
}
static float kge_d(const float* x, const float* y, const float* y_d, int n_, float* res) {
    int n = 0;
    float sum_x = 0.f, sum_y = 0.f, sum_xx = 0.f, sum_yy = 0.f, sum_xy = 0.f, sum_yy_d = 0.f, sum_y_d = 0.f, sum_xy_d = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) {
            n++;
            sum_x = sum_x + x[i]; sum_y_d = sum_y_d + y_d[i]; sum_y = sum_y + y[i];
            sum_xx = sum_xx + x[i] * x[i];
            sum_yy_d = sum_yy_d + 2.f * y[i] * y_d[i]; sum_yy = sum_yy + y[i] * y[i];
            sum_xy_d = sum_xy_d + x[i] * y_d[i]; sum_xy = sum_xy + x[i] * y[i];
        }
    float fn = (float)n;
    float mean_x = sum_x / fn, mean_y_d = sum_y_d / fn, mean_y = sum_y / fn;
    float var_x = sum_xx / fn - mean_x * mean_x;
    float var_y_d = sum_yy_d / fn - 2.f * mean_y * mean_y_d, var_y = sum_yy / fn - mean_y * mean_y;
    float cov_d = sum_xy_d / fn - mean_x * mean_y_d, cov = sum_xy / fn - mean_x * mean_y;
    float sx = sqrtf(var_x), sy = sqrtf(var_y);
    float sy_d = (var_y == 0.f) ? 0.f : var_y_d / (2.0f * sy);
    float r = cov / (sx * sy);
    float r_d = (cov_d - r * sx * sy_d) / (sx * sy);
    float a_d = sy_d / sx, a = sy / sx;
    float b_d = mean_y_d / mean_x, b = mean_y / mean_x;
    float arg1_d = 2.f * (r - 1.f) * r_d + 2.f * (b - 1.f) * b_d + 2.f * (a - 1.f) * a_d;
    float arg1 = (r - 1.f) * (r - 1.f) + (b - 1.f) * (b - 1.f) + (a - 1.f) * (a - 1.f);
    float t = sqrtf(arg1);
    *res = t;
    return (arg1 == 0.f) ? 0.f : arg1_d / (2.0f * t);
}
static float se_d(const float* x, const float* y, const float* y_d, int n_, float* res) {
    float r = 0.f, r_d = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) { r_d = r_d - 2.f * (x[i] - y[i]) * y_d[i]; r = r + (x[i] - y[i]) * (x[i] - y[i]); }
    *res = r;
    return r_d;
}
static float rmse_d(const float* x, const float* y, const float* y_d, int n_, float* res) {
    int n = 0;
    for (int i = 0; i < n_; ++i) if (x[i] >= 0.f) n++;
    float r1, r1_d = se_d(x, y, y_d, n_, &r1);
    float t = sqrtf(r1 / (float)n);
    *res = t;
    return (r1 / (float)n == 0.f) ? 0.f : r1_d / (2.0f * t * (float)n);
}
static float logarithmic_d(const float* x, const float* y, const float* y_d, int n_, float* res) {
    float r = 0.f, r_d = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] > 0.f && y[i] > 0.f) {
            float a_d = y_d[i] / x[i], a = y[i] / x[i];
            float t = logf(a);
            r_d = r_d + x[i] * (t * a_d / a + t * a_d / a);
            r = r + x[i] * (t * t);
        }
    *res = r;
    return r_d;
}

static void heap_sort2(int n, float* arr, float* arr_d) {   /* HEAP_SORT_D: the tangents move with their values */
    if (n < 2) return;
    int l = n / 2 + 1, ir = n;
    for (;;) {
        float a, ad;
        if (l > 1) { l = l - 1; a = arr[l - 1]; ad = arr_d[l - 1]; }
        else {
            a = arr[ir - 1]; ad = arr_d[ir - 1];
            arr[ir - 1] = arr[0]; arr_d[ir - 1] = arr_d[0];
            ir = ir - 1;
            if (ir == 1) { arr[0] = a; arr_d[0] = ad; return; }
        }
        int i = l, j = l + l;
        while (j <= ir) {
            if (j < ir && arr[j - 1] < arr[j]) j = j + 1;
            if (a < arr[j - 1]) { arr[i - 1] = arr[j - 1]; arr_d[i - 1] = arr_d[j - 1]; i = j; j = j + j; }
            else j = ir + 1;
        }
        arr[i - 1] = a; arr_d[i - 1] = ad;
    }
}

static void jobs_d(const orc_config* cfg, const int* flwacc, const int* gauge_pos, const float* area, const float* qobs,
                   const float* wgauge, const float* qsim, const float* qsim_d, float* jobs_out, float* jobs_d_out) {
    const int ng = cfg->ng, nt = cfg->nt, s0 = cfg->optimize_start_step - 1, n = nt - s0;
    const size_t m = (size_t)(n > 0 ? n : 1);
    float* qo = (float*)malloc(4 * m); float* qs = (float*)malloc(4 * m); float* qsd = (float*)malloc(4 * m);
    float* arr = (float*)malloc(4 * (size_t)(ng + 1)); float* arr_d = (float*)malloc(4 * (size_t)(ng + 1));
    int arr_size = 0;
    float jobs = 0.f, jd = 0.f;
    for (int g = 0; g < ng; ++g) {
        if (!(wgauge[g] > 0.f || wgauge[g] < 0.f)) continue;
        int row = gauge_pos[g], col = gauge_pos[g + ng], any = 0;
        for (int i = 0; i < n; ++i) {
            qs[i] = qsim[g + (long)ng * (s0 + i)] * cfg->dt / area[g] * 1e3f;
            qsd[i] = 1e3f * cfg->dt * qsim_d[g + (long)ng * (s0 + i)] / area[g];
            qo[i] = qobs[g + (long)ng * (s0 + i)] * cfg->dt / ((float)flwacc[row + (long)col * cfg->nrow] * cfg->dx * cfg->dx) * 1e3f;
            if (qo[i] >= 0.f) any = 1;
        }
        float gj = 0.f, gj_d = 0.f, j_imd = 0.f, j_imd_d = 0.f;
        for (int j = 0; j < cfg->njf; ++j) {
            if (any) {
                switch (cfg->jobs_fun[j]) {
                    case ORC_NSE: j_imd_d = nse_d(qo, qs, qsd, n, &j_imd); break;
                    case ORC_KGE: j_imd_d = kge_d(qo, qs, qsd, n, &j_imd); break;
                    case ORC_KGE2: { float imd, imd_d = kge_d(qo, qs, qsd, n, &imd); j_imd_d = 2.f * imd * imd_d; j_imd = imd * imd; } break;
                    case ORC_SE: j_imd_d = se_d(qo, qs, qsd, n, &j_imd); break;
                    case ORC_RMSE: j_imd_d = rmse_d(qo, qs, qsd, n, &j_imd); break;
                    case ORC_LOGARITHMIC: j_imd_d = logarithmic_d(qo, qs, qsd, n, &j_imd); break;
                    default: break;
                }
            }
            gj_d = gj_d + cfg->wjobs_fun[j] * j_imd_d;
            gj = gj + cfg->wjobs_fun[j] * j_imd;
        }
        if (wgauge[g] > 0.f) { jd = jd + wgauge[g] * gj_d; jobs = jobs + wgauge[g] * gj; }
        else { arr[arr_size] = gj; arr_d[arr_size] = gj_d; ++arr_size; }
    }
    if (arr_size > 0) {   /* QUANTILE_D, p = 0.5 */
        jobs = arr[0]; jd = arr_d[0];
        if (arr_size > 1) {
            heap_sort2(arr_size, arr, arr_d);
            const float frac = (float)(arr_size - 1) * 0.5f + 1.f;
            if (frac <= 1.f) { jobs = arr[0]; jd = arr_d[0]; }
            else if (frac >= (float)arr_size) { jobs = arr[arr_size - 1]; jd = arr_d[arr_size - 1]; }
            else {
                const int k = (int)frac;
                const float t = frac - (float)k;
                jd = arr_d[k - 1] + t * (arr_d[k] - arr_d[k - 1]);
                jobs = arr[k - 1] + t * (arr[k] - arr[k - 1]);
            }
        }
    }
    *jobs_out = jobs; *jobs_d_out = jd;
    free(qo); free(qs); free(qsd); free(arr); free(arr_d);
}

static void bounds(const int* active, int nrow, int ncol, int row, int col, int* mnc, int* mxc, int* mnr, int* mxr) {
    int a = col - 1 > 0 ? col - 1 : 0, b = col + 1 < ncol - 1 ? col + 1 : ncol - 1;
    int c = row - 1 > 0 ? row - 1 : 0, d = row + 1 < nrow - 1 ? row + 1 : nrow - 1;
    if (active[row + (long)a * nrow] == 0) a = col;
    if (active[row + (long)b * nrow] == 0) b = col;
    if (active[c + (long)col * nrow] == 0) c = row;
    if (active[d + (long)col * nrow] == 0) d = row;
    *mnc = a; *mxc = b; *mnr = c; *mxr = d;
}
static float prior_d(const int* optim, int nf, long n2, const float* m, const float* md, const float* mb) {
    float r = 0.f;
    for (int i = 0; i < nf; ++i)
        if (optim[i] > 0)
            for (long c = 0; c < n2; ++c) r = r + 2.f * (m[i * n2 + c] - mb[i * n2 + c]) * md[i * n2 + c];
    return r;
}
static float smoothing_d(const int* active, int nrow, int ncol, const int* optim, int nf, const float* m, const float* md,
                         const float* mb, int rel) {
    const long n2 = (long)nrow * ncol;
    float r = 0.f;
#define MAT(rr, cc, i) (rel ? (m[(i) * n2 + (rr) + (long)(cc) * nrow] - mb[(i) * n2 + (rr) + (long)(cc) * nrow]) : m[(i) * n2 + (rr) + (long)(cc) * nrow])
#define MATD(rr, cc, i) md[(i) * n2 + (rr) + (long)(cc) * nrow]
    for (int i = 0; i < nf; ++i)
        if (optim[i] > 0)
            for (int col = 0; col < ncol; ++col)
                for (int row = 0; row < nrow; ++row)
                    if (active[row + (long)col * nrow] == 1) {
                        int mnc, mxc, mnr, mxr;
                        bounds(active, nrow, ncol, row, col, &mnc, &mxc, &mnr, &mxr);
                        r = r + 2.f * (MAT(mxr, col, i) - 2.f * MAT(row, col, i) + MAT(mnr, col, i)) *
                                    (MATD(mxr, col, i) - 2.f * MATD(row, col, i) + MATD(mnr, col, i)) +
                            2.f * (MAT(row, mxc, i) - 2.f * MAT(row, col, i) + MAT(row, mnc, i)) *
                                (MATD(row, mxc, i) - 2.f * MATD(row, col, i) + MATD(row, mnc, i));
                    }
#undef MAT
#undef MATD
    return r;
}

/* base_forward_d.  params_d / states_d in: the direction (normalised space when denormalize_forward); they come back as
 * BASE_FORWARD_D leaves them apart from the quirk in the header.  qsim_d (ng, nt), cost_d out. */
int orc_forward_d(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path, const int* active,
                  const int* gauge_pos, const float* area, const float* prcp, const float* pet, const float* qobs,
                  const float* wgauge, float* P, float* P_d, const float* Pb, float* S, float* S_d, const float* Sb,
                  float* qsim, float* qsim_d, float* costs, float* cost_d) {
    const int nrow = cfg->nrow, ncol = cfg->ncol;
    const long n2 = (long)nrow * ncol;
    if (cfg->structure < ORC_GR_A || cfg->structure > ORC_VIC_A) return -2;
    if (cfg->denormalize_forward) {   /* DENORMALIZE_*_D */
        for (int i = 0; i < ORC_GNP; ++i)
            for (long c = 0; c < n2; ++c) {
                P_d[i * n2 + c] = (cfg->ub_parameters[i] - cfg->lb_parameters[i]) * P_d[i * n2 + c];
                P[i * n2 + c] = P[i * n2 + c] * (cfg->ub_parameters[i] - cfg->lb_parameters[i]) + cfg->lb_parameters[i];
            }
        for (int i = 0; i < ORC_GNS; ++i)
            for (long c = 0; c < n2; ++c) {
                S_d[i * n2 + c] = (cfg->ub_states[i] - cfg->lb_states[i]) * S_d[i * n2 + c];
                S[i * n2 + c] = S[i * n2 + c] * (cfg->ub_states[i] - cfg->lb_states[i]) + cfg->lb_states[i];
            }
    }
    float* S0 = (float*)malloc(sizeof(float) * (size_t)(ORC_GNS * n2));
    float* S0_d = (float*)malloc(sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(S0, S, sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(S0_d, S_d, sizeof(float) * (size_t)(ORC_GNS * n2));
    float* q = (float*)calloc((size_t)n2, sizeof(float));
    float* q_d = (float*)calloc((size_t)n2, sizeof(float));
    for (int t = 0; t < cfg->nt; ++t) {
        for (long i = 0; i < n2; ++i) {
            int row = path[2 * i], col = path[2 * i + 1];
            if (row < 0 || col < 0) continue;
            long c = row + (long)col * nrow;
            if (active[c] != 1) continue;
            cell_step_d(cfg->structure, cfg->dt, cfg->dx, nrow, ncol, flwdir, flwacc, row, col, prcp[c + n2 * t], pet[c + n2 * t],
                        P, P_d, S, S_d, q, q_d);
        }
        for (int g = 0; g < cfg->ng; ++g) {
            long c = gauge_pos[g] + (long)gauge_pos[g + cfg->ng] * nrow;
            qsim[g + (long)cfg->ng * t] = q[c];
            qsim_d[g + (long)cfg->ng * t] = q_d[c];
        }
    }
    free(q); free(q_d);
    memcpy(S, S0, sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(S_d, S0_d, sizeof(float) * (size_t)(ORC_GNS * n2));
    free(S0); free(S0_d);
    float jobs = 0.f, jd = 0.f;
    jobs_d(cfg, flwacc, gauge_pos, area, qobs, wgauge, qsim, qsim_d, &jobs, &jd);
    /* jreg on the control vector as compute_cost sees it (normalised again when denormalize_forward) */
    float jreg_d = 0.f;
    if (cfg->njr > 0) {
        float* X = (float*)malloc(sizeof(float) * (size_t)((ORC_GNP + ORC_GNS) * n2));
        float* Xd = (float*)malloc(sizeof(float) * (size_t)((ORC_GNP + ORC_GNS) * n2));
        float *XP = X, *XS = X + ORC_GNP * n2, *XPd = Xd, *XSd = Xd + ORC_GNP * n2;
        memcpy(XP, P, sizeof(float) * (size_t)(ORC_GNP * n2)); memcpy(XS, S, sizeof(float) * (size_t)(ORC_GNS * n2));
        memcpy(XPd, P_d, sizeof(float) * (size_t)(ORC_GNP * n2)); memcpy(XSd, S_d, sizeof(float) * (size_t)(ORC_GNS * n2));
        if (cfg->denormalize_forward) {   /* NORMALIZE_*_D */
            for (int i = 0; i < ORC_GNP; ++i)
                for (long c = 0; c < n2; ++c) {
                    XPd[i * n2 + c] = XPd[i * n2 + c] / (cfg->ub_parameters[i] - cfg->lb_parameters[i]);
                    XP[i * n2 + c] = (XP[i * n2 + c] - cfg->lb_parameters[i]) / (cfg->ub_parameters[i] - cfg->lb_parameters[i]);
                }
            for (int i = 0; i < ORC_GNS; ++i)
                for (long c = 0; c < n2; ++c) {
                    XSd[i * n2 + c] = XSd[i * n2 + c] / (cfg->ub_states[i] - cfg->lb_states[i]);
                    XS[i * n2 + c] = (XS[i * n2 + c] - cfg->lb_states[i]) / (cfg->ub_states[i] - cfg->lb_states[i]);
                }
        }
        float pj = 0.f, sj = 0.f;
        for (int i = 0; i < cfg->njr; ++i) {
            const float w = cfg->wjreg_fun[i];
            if (cfg->jreg_fun[i] == ORC_PRIOR) {
                pj = pj + w * prior_d(cfg->optim_parameters, ORC_GNP, n2, XP, XPd, Pb);
                sj = sj + w * prior_d(cfg->optim_states, ORC_GNS, n2, XS, XSd, Sb);
            } else if (cfg->jreg_fun[i] == ORC_SMOOTHING || cfg->jreg_fun[i] == ORC_HARD_SMOOTHING) {
                const int rel = cfg->jreg_fun[i] == ORC_SMOOTHING;
                const float w2 = powf(w, 2.f);
                pj = pj + w2 * smoothing_d(active, nrow, ncol, cfg->optim_parameters, ORC_GNP, XP, XPd, Pb, rel);
                sj = sj + w2 * smoothing_d(active, nrow, ncol, cfg->optim_states, ORC_GNS, XS, XSd, Sb, rel);
            }
        }
        jreg_d = pj + sj;
        free(X); free(Xd);
    }
    *cost_d = jd + cfg->wjreg * jreg_d;
    if (costs) { costs[1] = jobs; }
    return 0;
}
