/* smash_oracle.c -- plain-C, single-thread, fp32 restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see smash_oracle.h).  Build: gcc -O2 -ffp-contract=off (oracle/Makefile).
 * Every routine cites the reference file:line it follows (paths relative to /root/reference).
 * The forward follows the reference statement by statement (same operation order, same libm calls:
 * tanhf/powf/expf/logf/sqrtf), so against the flang -O2 -ffp-contract=off build it is expected to be
 * bit-identical; the adjoint is hand-derived in store-all form but evaluates the same local
 * adjoint expressions, in the same order, as the Tapenade output (forward_db.f90).
 */
#include "smash_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* operators: smash/solver/operator/md_gr_operator.f90, md_routing_operator.f90                 */
/* ------------------------------------------------------------------------------------------- */

/* md_gr_operator.f90:20-34 */
static void gr_interception(float prcp, float pet, float ci, float* hi, float* pn, float* ei) {
    *ei = fminf(pet, prcp + (*hi) * ci);
    *pn = fmaxf(0.f, prcp - ci * (1.f - *hi) - *ei);
    *hi = *hi + (prcp - *ei - *pn) / ci;
}

/* md_gr_operator.f90:36-67 */
static void gr_production(float pn, float en, float cp, float beta, float* hp, float* pr, float* perc) {
    float inv_cp = 1.f / cp;
    *pr = 0.f;
    float h = *hp;
    float ps = cp * (1.f - h * h) * tanhf(pn * inv_cp) / (1.f + h * tanhf(pn * inv_cp));
    float es = (h * cp) * (2.f - h) * tanhf(en * inv_cp) / (1.f + (1.f - h) * tanhf(en * inv_cp));
    float hp_imd = h + (ps - es) * inv_cp;
    if (pn > 0.f) *pr = pn - (hp_imd - h) * cp;
    float r = hp_imd / beta;
    float r2 = r * r;
    *perc = (hp_imd * cp) * (1.f - powf(1.f + r2 * r2, -0.25f));
    *hp = hp_imd - (*perc) * inv_cp;
}

/* md_gr_operator.f90:69-79 */
static void gr_exchange(float exc, float hft, float* l) { *l = exc * powf(hft, 3.5f); }

/* md_gr_operator.f90:81-110 */
static void gr_transfer(float n, float prcp, float pr, float ct, float* ht, float* q) {
    float nm1 = n - 1.f;
    float d1pnm1 = 1.f / nm1;
    float pr_imd;
    if (prcp < 0.f)
        pr_imd = powf(powf((*ht) * ct, -nm1) - powf(ct, -nm1), -d1pnm1) - ((*ht) * ct);
    else
        pr_imd = pr;
    float ht_imd = fmaxf(1.e-6f, *ht + pr_imd / ct);
    *ht = powf(powf(ht_imd * ct, -nm1) + powf(ct, -nm1), -d1pnm1) / ct;
    *q = (ht_imd - *ht) * ct;
}

/* md_routing_operator.f90:17-60 */
static float upstream_discharge(float dt, float dx, int nrow, int ncol, const int* flwdir,
                                const int* flwacc, int row, int col, const float* q) {
    static const int dcol[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    static const int drow[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    float qup = 0.f;
    if (flwacc[row + col * nrow] > 1) {
        for (int i = 0; i < 8; ++i) {
            int c = col + dcol[i], r = row + drow[i];
            if (c >= 0 && c < ncol && r >= 0 && r < nrow)
                if (flwdir[r + c * nrow] == i + 1) qup = qup + q[r + c * nrow];
        }
        qup = (qup * dt) / (0.001f * dx * dx * (float)(flwacc[row + col * nrow] - 1));
    }
    return qup;
}

/* md_routing_operator.f90:62-79 */
static void linear_routing(float dt, float qup, float lr, float* hr, float* qrout) {
    float hr_imd = *hr + qup;
    *hr = hr_imd * expf(-dt / (lr * 60.f));
    *qrout = hr_imd - *hr;
}

/* ------------------------------------------------------------------------------------------- */
/* field views into the (nrow,ncol,16)/(nrow,ncol,8) packings                                   */
/* ------------------------------------------------------------------------------------------- */
enum { P_CI = 0, P_CP = 1, P_BETA = 2, P_CFT = 3, P_CST = 4, P_ALPHA = 5, P_EXC = 6, P_B = 7, P_CUSL1 = 8, P_CUSL2 = 9,
       P_CLSL = 10, P_KS = 11, P_DS = 12, P_DSM = 13, P_WS = 14, P_LR = 15 };
enum { S_HI = 0, S_HP = 1, S_HFT = 2, S_HST = 3, S_HUSL1 = 4, S_HUSL2 = 5, S_HLSL = 6, S_HLR = 7 };

#include "smash_oracle_vic.h"

/* one cell-step of gr_{a,b,c,d}_forward: md_forward_structure.f90:62-156 / 248-340 / 432-520 / 621-698.
 * tape (optional, 8 floats): pre-step hi,hp,hft,hst,hlr, qup, prcp, pet -- what the reverse sweep needs. */
/* optional whole-domain stores (md_forward_structure.f90:158-194): set through orc_set_domain_outputs */
static float* g_qsim_domain = 0;     /* (nrow, ncol, nt) column-major, inactive cells left untouched */
static float* g_net_prcp_domain = 0;
static float* g_qt_out = 0;
void orc_set_domain_outputs(float* qsim_domain, float* net_prcp_domain) {
    g_qsim_domain = qsim_domain; g_net_prcp_domain = net_prcp_domain;
}

static void cell_step(int st, float dt, float dx, int nrow, int ncol, const int* flwdir, const int* flwacc,
                      int row, int col, float prcp, float pet, const float* P, float* S, float* q,
                      float* tape) {
    const long n2 = (long)nrow * ncol;
    const long c = row + (long)col * nrow;
    float ei = 0.f, pn = 0.f, en = 0.f, pr = 0.f, perc = 0.f, l = 0.f, prr, prl = 0.f, prd = 0.f;
    float qr = 0.f, ql = 0.f, qd = 0.f, qt, qup, qrout;
    float* hi = S + S_HI * n2 + c;
    float* hp = S + S_HP * n2 + c;
    float* hft = S + S_HFT * n2 + c;
    float* hst = S + S_HST * n2 + c;
    float* hlr = S + S_HLR * n2 + c;
    if (st == ORC_VIC_A) {   /* vic_a_forward, md_forward_structure.f90:762-931 */
        float* husl1 = S + S_HUSL1 * n2 + c;
        float* husl2 = S + S_HUSL2 * n2 + c;
        float* hlsl = S + S_HLSL * n2 + c;
        if (tape) { tape[0] = *husl1; tape[1] = *husl2; tape[2] = *hlsl; tape[3] = 0.f; tape[4] = *hlr; tape[6] = prcp; tape[7] = pet; }
        float runoff = 0.f, qi = 0.f, qb = 0.f;
        if (prcp >= 0.f && pet >= 0.f) {
            vic_infiltration(prcp, P[P_CUSL1 * n2 + c], P[P_CUSL2 * n2 + c], P[P_B * n2 + c], husl1, husl2, &runoff);
            vic_vertical_transfer(pet, P[P_CUSL1 * n2 + c], P[P_CUSL2 * n2 + c], P[P_CLSL * n2 + c], P[P_KS * n2 + c], husl1, husl2, hlsl);
        }
        vic_interflow(5.f, P[P_CUSL2 * n2 + c], husl2, &qi);
        vic_baseflow(P[P_CLSL * n2 + c], P[P_DS * n2 + c], P[P_DSM * n2 + c], P[P_WS * n2 + c], hlsl, &qb);
        qt = (runoff + qi + qb);
        if (g_qt_out) *g_qt_out = qt;
        qup = upstream_discharge(dt, dx, nrow, ncol, flwdir, flwacc, row, col, q);
        if (tape) tape[5] = qup;
        linear_routing(dt, qup, P[P_LR * n2 + c], hlr, &qrout);
        q[c] = (qt + qrout * (float)(flwacc[c] - 1)) * dx * dx * 0.001f / dt;
        return;
    }
    if (tape) { tape[0] = *hi; tape[1] = *hp; tape[2] = *hft; tape[3] = *hst; tape[4] = *hlr; tape[6] = prcp; tape[7] = pet; }

    if (prcp >= 0.f && pet >= 0.f) {
        if (st == ORC_GR_A || st == ORC_GR_D) {
            ei = fminf(pet, prcp);
            pn = fmaxf(0.f, prcp - ei);
        } else {
            gr_interception(prcp, pet, P[P_CI * n2 + c], hi, &pn, &ei);
        }
        en = pet - ei;
        gr_production(pn, en, P[P_CP * n2 + c], 1000.f, hp, &pr, &perc);
        if (st != ORC_GR_D) gr_exchange(P[P_EXC * n2 + c], *hft, &l);
    }
    if (st == ORC_GR_A || st == ORC_GR_B) {
        prr = 0.9f * (pr + perc) + l;
        prd = 0.1f * (pr + perc);
        gr_transfer(5.f, prcp, prr, P[P_CFT * n2 + c], hft, &qr);
        qd = fmaxf(0.f, prd + l);
        qt = (qr + qd);
    } else if (st == ORC_GR_C) {
        prr = 0.9f * 0.6f * (pr + perc) + l;
        prl = 0.9f * 0.4f * (pr + perc);
        prd = 0.1f * (pr + perc);
        gr_transfer(5.f, prcp, prr, P[P_CFT * n2 + c], hft, &qr);
        gr_transfer(5.f, prcp, prl, P[P_CST * n2 + c], hst, &ql);
        qd = fmaxf(0.f, prd + l);
        qt = (qr + ql + qd);
    } else {
        prr = pr + perc;
        gr_transfer(5.f, prcp, prr, P[P_CFT * n2 + c], hft, &qr);
        qt = qr;
    }
    if (g_qt_out) *g_qt_out = qt;
    qup = upstream_discharge(dt, dx, nrow, ncol, flwdir, flwacc, row, col, q);
    if (tape) tape[5] = qup;
    linear_routing(dt, qup, P[P_LR * n2 + c], hlr, &qrout);
    q[c] = (qt + qrout * (float)(flwacc[c] - 1)) * dx * dx * 0.001f / dt;
}

/* time x path loops + gauge sampling (md_forward_structure.f90:57-60, 206-210) */
static void structure_forward(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                              const int* active, const int* gauge_pos, const float* prcp, const float* pet,
                              const float* P, float* S, float* qsim, float* tape) {
    const int nrow = cfg->nrow, ncol = cfg->ncol;
    const long n2 = (long)nrow * ncol;
    float* q = (float*)calloc((size_t)n2, sizeof(float));
    for (int t = 0; t < cfg->nt; ++t) {
        for (long i = 0; i < n2; ++i) {
            int row = path[2 * i], col = path[2 * i + 1];
            if (row < 0 || col < 0) continue;
            long c = row + (long)col * nrow;
            if (active[c] != 1) continue;
            float qt_cell = 0.f;
            g_qt_out = g_net_prcp_domain ? &qt_cell : 0;
            cell_step(cfg->structure, cfg->dt, cfg->dx, nrow, ncol, flwdir, flwacc, row, col,
                      prcp[c + n2 * t], pet[c + n2 * t], P, S, q, tape ? tape + 8 * (i + n2 * t) : 0);
            g_qt_out = 0;
            if (g_net_prcp_domain) g_net_prcp_domain[c + n2 * t] = qt_cell;
            if (g_qsim_domain) g_qsim_domain[c + n2 * t] = q[c];
        }
        for (int g = 0; g < cfg->ng; ++g)
            qsim[g + (long)cfg->ng * t] = q[gauge_pos[g] + (long)gauge_pos[g + cfg->ng] * nrow];
    }
    free(q);
}

/* ------------------------------------------------------------------------------------------- */
/* normalise / denormalise: mwd_parameters_manipulation.f90:154-206, mwd_states_manipulation.f90:137-189 */
/* ------------------------------------------------------------------------------------------- */
static void normalize(float* a, long n2, int nf, const float* lb, const float* ub) {
    for (int i = 0; i < nf; ++i)
        for (long c = 0; c < n2; ++c) a[i * n2 + c] = (a[i * n2 + c] - lb[i]) / (ub[i] - lb[i]);
}
static void denormalize(float* a, long n2, int nf, const float* lb, const float* ub) {
    for (int i = 0; i < nf; ++i)
        for (long c = 0; c < n2; ++c) a[i * n2 + c] = a[i * n2 + c] * (ub[i] - lb[i]) + lb[i];
}

/* ------------------------------------------------------------------------------------------- */
/* cost functions: smash/solver/optimize/mwd_cost.f90                                           */
/* ------------------------------------------------------------------------------------------- */
/* mwd_cost.f90:350-401 */
static float nse(const float* x, const float* y, int n_) {
    int n = 0;
    float sum_x = 0.f, sum_xx = 0.f, sum_yy = 0.f, sum_xy = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) {
            n++;
            sum_x = sum_x + x[i];
            sum_xx = sum_xx + (x[i] * x[i]);
            sum_yy = sum_yy + (y[i] * y[i]);
            sum_xy = sum_xy + (x[i] * y[i]);
        }
    float mean_x = sum_x / (float)n;
    float num = sum_xx - 2.f * sum_xy + sum_yy;
    float den = sum_xx - (float)n * mean_x * mean_x;
    return num / den;
}
/* forward_db.f90:3505-3545 */
static void nse_b(const float* x, const float* y, float* y_b, int n_, float res_b) {
    int n = 0;
    float sum_x = 0.f, sum_xx = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) { n++; sum_x = sum_x + x[i]; sum_xx = sum_xx + x[i] * x[i]; }
    float mean_x = sum_x / (float)n;
    float den = sum_xx - (float)n * mean_x * mean_x;
    float num_b = res_b / den;
    float sum_yy_b = num_b, sum_xy_b = -(2.f * num_b);
    for (int i = n_ - 1; i >= 0; --i)
        if (x[i] >= 0.f) y_b[i] = y_b[i] + x[i] * sum_xy_b + 2.f * y[i] * sum_yy_b;
}

typedef struct { int n; float mean_x, mean_y, var_x, var_y, cov, r, a, b; } kge_c;
/* mwd_cost.f90:403-455 */
static kge_c kge_components(const float* x, const float* y, int n_) {
    kge_c k;
    int n = 0;
    float sum_x = 0.f, sum_y = 0.f, sum_xx = 0.f, sum_yy = 0.f, sum_xy = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) {
            n++;
            sum_x = sum_x + x[i];
            sum_y = sum_y + y[i];
            sum_xx = sum_xx + (x[i] * x[i]);
            sum_yy = sum_yy + (y[i] * y[i]);
            sum_xy = sum_xy + (x[i] * y[i]);
        }
    k.n = n;
    k.mean_x = sum_x / (float)n;
    k.mean_y = sum_y / (float)n;
    k.var_x = (sum_xx / (float)n) - (k.mean_x * k.mean_x);
    k.var_y = (sum_yy / (float)n) - (k.mean_y * k.mean_y);
    k.cov = (sum_xy / (float)n) - (k.mean_x * k.mean_y);
    k.r = (k.cov / sqrtf(k.var_x)) / sqrtf(k.var_y);
    k.a = sqrtf(k.var_y) / sqrtf(k.var_x);
    k.b = k.mean_y / k.mean_x;
    return k;
}
/* mwd_cost.f90:457-490 */
static float kge(const float* x, const float* y, int n_) {
    kge_c k = kge_components(x, y, n_);
    return sqrtf((k.r - 1.f) * (k.r - 1.f) + (k.b - 1.f) * (k.b - 1.f) + (k.a - 1.f) * (k.a - 1.f));
}
/* forward_db.f90:3657-3729 + 3805-3829 */
static void kge_b(const float* x, const float* y, float* y_b, int n_, float res_b) {
    kge_c k = kge_components(x, y, n_);
    float arg1 = (k.r - 1.f) * (k.r - 1.f) + (k.b - 1.f) * (k.b - 1.f) + (k.a - 1.f) * (k.a - 1.f);
    float arg1_b = (arg1 == 0.f) ? 0.f : res_b / (2.0f * sqrtf(arg1));
    float r_b = 2.f * (k.r - 1.f) * arg1_b, b_b = 2.f * (k.b - 1.f) * arg1_b, a_b = 2.f * (k.a - 1.f) * arg1_b;
    float n = (float)k.n;
    float result1 = sqrtf(k.var_x), result2 = sqrtf(k.var_y);
    float result1_b = a_b / sqrtf(k.var_x);
    float var_y_b = (k.var_y == 0.f) ? 0.f : result1_b / (2.0f * sqrtf(k.var_y));
    float temp_b = r_b / (result1 * result2);
    float cov_b = temp_b;
    float result2_b = -(k.cov * temp_b / result2);
    if (!(k.var_y == 0.f)) var_y_b = var_y_b + result2_b / (2.0f * sqrtf(k.var_y));
    float mean_y_b = b_b / k.mean_x - k.mean_x * cov_b - 2.f * k.mean_y * var_y_b;
    float sum_xy_b = cov_b / n, sum_yy_b = var_y_b / n, sum_y_b = mean_y_b / n;
    for (int i = n_ - 1; i >= 0; --i)
        if (x[i] >= 0.f) y_b[i] = y_b[i] + x[i] * sum_xy_b + 2.f * y[i] * sum_yy_b + sum_y_b;
}
/* mwd_cost.f90:492-519 */
static float se(const float* x, const float* y, int n_) {
    float res = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) res = res + (x[i] - y[i]) * (x[i] - y[i]);
    return res;
}
/* forward_db.f90:3868-3890 */
static void se_b(const float* x, const float* y, float* y_b, int n_, float res_b) {
    for (int i = n_ - 1; i >= 0; --i)
        if (x[i] >= 0.f) y_b[i] = y_b[i] - 2.f * (x[i] - y[i]) * res_b;
}
/* mwd_cost.f90:521-556 */
static float rmse(const float* x, const float* y, int n_) {
    int n = 0;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) n++;
    return sqrtf(se(x, y, n_) / (float)n);
}
/* forward_db.f90:3936-3970 */
static void rmse_b(const float* x, const float* y, float* y_b, int n_, float res_b) {
    int n = 0;
    for (int i = 0; i < n_; ++i)
        if (x[i] >= 0.f) n++;
    float result1 = se(x, y, n_);
    float result1_b = (result1 / (float)n == 0.f) ? 0.f : res_b / ((float)n * 2.0f * sqrtf(result1 / (float)n));
    se_b(x, y, y_b, n_, result1_b);
}
/* mwd_cost.f90:558-592 */
static float logarithmic(const float* x, const float* y, int n_) {
    float res = 0.f;
    for (int i = 0; i < n_; ++i)
        if (x[i] > 0.f && y[i] > 0.f) res = res + x[i] * logf(y[i] / x[i]) * logf(y[i] / x[i]);
    return res;
}
/* forward_db.f90:4025-4060 */
static void logarithmic_b(const float* x, const float* y, float* y_b, int n_, float res_b) {
    for (int i = n_ - 1; i >= 0; --i)
        if (x[i] > 0.f && y[i] > 0.f) {
            float arg1 = y[i] / x[i], arg2 = y[i] / x[i];
            float arg1_b = logf(arg2) * x[i] * res_b / arg1;
            float arg2_b = logf(arg1) * x[i] * res_b / arg2;
            y_b[i] = y_b[i] + arg2_b / x[i] + arg1_b / x[i];
        }
}

/* mwd_cost.f90:594-673 (heap_sort): sift-down heap sort; idx moves with the values so that the adjoint
 * (HEAP_SORT_B, forward_db.f90: pure data movement) is the inverse permutation, ties included. */
static void heap_sort_idx(int n, float* arr, int* idx) {
    if (n < 2) return;
    int l = n / 2 + 1, ir = n;
    for (;;) {
        float arr_l; int idx_l;
        if (l > 1) { l = l - 1; arr_l = arr[l - 1]; idx_l = idx[l - 1]; }
        else {
            arr_l = arr[ir - 1]; idx_l = idx[ir - 1];
            arr[ir - 1] = arr[0]; idx[ir - 1] = idx[0];
            ir = ir - 1;
            if (ir == 1) { arr[0] = arr_l; idx[0] = idx_l; return; }
        }
        int i = l, j = l + l;
        while (j <= ir) {
            if (j < ir && arr[j - 1] < arr[j]) j = j + 1;
            if (arr_l < arr[j - 1]) { arr[i - 1] = arr[j - 1]; idx[i - 1] = idx[j - 1]; i = j; j = j + j; }
            else j = ir + 1;
        }
        arr[i - 1] = arr_l; idx[i - 1] = idx_l;
    }
}
/* mwd_cost.f90:675-723 (quantile, p = 0.5) and QUANTILE_B forward_db.f90:4327-4370: returns the median of
 * dat[0..n) and, when dat_b != NULL, the weights d(res)/d(dat[k]) * res_b */
static float median_b(int n, const float* dat, float* dat_b, float res_b) {
    float res = dat[0];
    if (dat_b) for (int k = 0; k < n; ++k) dat_b[k] = 0.f;
    if (n > 1) {
        float* sd = (float*)malloc(sizeof(float) * (size_t)n);
        int* id = (int*)malloc(sizeof(int) * (size_t)n);
        for (int k = 0; k < n; ++k) { sd[k] = dat[k]; id[k] = k; }
        heap_sort_idx(n, sd, id);
        const float frac = (float)(n - 1) * 0.5f + 1.f;
        if (frac <= 1.f) { res = sd[0]; if (dat_b) dat_b[id[0]] = dat_b[id[0]] + res_b; }
        else if (frac >= (float)n) { res = sd[n - 1]; if (dat_b) dat_b[id[n - 1]] = dat_b[id[n - 1]] + res_b; }
        else {
            const int k = (int)frac;
            const float q1 = sd[k - 1], q2 = sd[k];
            res = q1 + (q2 - q1) * (frac - (float)k);
            if (dat_b) {
                const float temp_b = (frac - (float)k) * res_b;
                dat_b[id[k]] = dat_b[id[k]] + temp_b;
                dat_b[id[k - 1]] = dat_b[id[k - 1]] + (res_b - temp_b);
            }
        }
        free(sd); free(id);
    } else if (dat_b) {
        dat_b[0] = dat_b[0] + res_b;
    }
    return res;
}

/* mwd_cost.f90:37-156 (compute_jobs); with qsim_b != NULL also forward_db.f90:2553-2715 (compute_jobs_b).
 * Gauges with a negative weight enter a median instead of the weighted sum; as soon as there is one, the
 * median REPLACES the weighted sum (mwd_cost.f90:139-154) and jobs_b no longer reaches the weighted gauges. */
static int compute_jobs(const orc_config* cfg, const int* flwacc, const int* gauge_pos, const float* area,
                        const float* qobs, const float* wgauge, const float* qsim, float* jobs, float jobs_b,
                        float* qsim_b) {
    const int ng = cfg->ng, nt = cfg->nt, s0 = cfg->optimize_start_step - 1, n = nt - s0;
    float* qo = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* qs = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* qs_b = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* arr = (float*)malloc(sizeof(float) * (size_t)(ng > 0 ? ng : 1));
    float* arr_b = (float*)malloc(sizeof(float) * (size_t)(ng > 0 ? ng : 1));
    int arr_size = 0;
    *jobs = 0.f;
    if (qsim_b) memset(qsim_b, 0, sizeof(float) * (size_t)ng * nt);
    for (int g = 0; g < ng; ++g) {
        if (!(wgauge[g] > 0.f || wgauge[g] < 0.f)) continue;
        int row = gauge_pos[g], col = gauge_pos[g + ng];
        int any = 0;
        for (int i = 0; i < n; ++i) {
            qs[i] = qsim[g + (long)ng * (s0 + i)] * cfg->dt / area[g] * 1e3f;
            qo[i] = qobs[g + (long)ng * (s0 + i)] * cfg->dt /
                    ((float)flwacc[row + (long)col * cfg->nrow] * cfg->dx * cfg->dx) * 1e3f;
            if (qo[i] >= 0.f) any = 1;
        }
        float gauge_jobs = 0.f, j_imd = 0.f;
        for (int j = 0; j < cfg->njf; ++j) {
            if (any) {
                switch (cfg->jobs_fun[j]) {
                    case ORC_NSE: j_imd = nse(qo, qs, n); break;
                    case ORC_KGE: j_imd = kge(qo, qs, n); break;
                    case ORC_KGE2: { float imd = kge(qo, qs, n); j_imd = imd * imd; } break;
                    case ORC_SE: j_imd = se(qo, qs, n); break;
                    case ORC_RMSE: j_imd = rmse(qo, qs, n); break;
                    case ORC_LOGARITHMIC: j_imd = logarithmic(qo, qs, n); break;
                    default: break;
                }
            }
            gauge_jobs = gauge_jobs + cfg->wjobs_fun[j] * j_imd;
        }
        if (wgauge[g] > 0.f) *jobs = *jobs + wgauge[g] * gauge_jobs;
        else arr[arr_size++] = gauge_jobs;
    }
    if (arr_size > 0) {
        *jobs = median_b(arr_size, arr, qsim_b ? arr_b : 0, jobs_b);
        jobs_b = 0.f;
    }
    if (qsim_b) {
        float j_imd_b = 0.f;
        for (int g = ng - 1; g >= 0; --g) {
            if (!(wgauge[g] > 0.f || wgauge[g] < 0.f)) continue;
            int row = gauge_pos[g], col = gauge_pos[g + ng];
            int any = 0;
            for (int i = 0; i < n; ++i) {
                qs[i] = qsim[g + (long)ng * (s0 + i)] * cfg->dt / area[g] * 1e3f;
                qo[i] = qobs[g + (long)ng * (s0 + i)] * cfg->dt /
                        ((float)flwacc[row + (long)col * cfg->nrow] * cfg->dx * cfg->dx) * 1e3f;
                if (qo[i] >= 0.f) any = 1;
                qs_b[i] = 0.f;
            }
            float gauge_jobs_b;
            if (wgauge[g] > 0.f) gauge_jobs_b = wgauge[g] * jobs_b;
            else gauge_jobs_b = arr_b[--arr_size];
            for (int j = cfg->njf - 1; j >= 0; --j) {
                j_imd_b = j_imd_b + cfg->wjobs_fun[j] * gauge_jobs_b;
                if (!any) continue; /* control 8: j_imd_b keeps accumulating (forward_db.f90:2672-2704) */
                switch (cfg->jobs_fun[j]) {
                    case ORC_NSE: nse_b(qo, qs, qs_b, n, j_imd_b); j_imd_b = 0.f; break;
                    case ORC_KGE: kge_b(qo, qs, qs_b, n, j_imd_b); j_imd_b = 0.f; break;
                    case ORC_KGE2: { float imd = kge(qo, qs, n); kge_b(qo, qs, qs_b, n, 2.f * imd * j_imd_b); j_imd_b = 0.f; } break;
                    case ORC_SE: se_b(qo, qs, qs_b, n, j_imd_b); j_imd_b = 0.f; break;
                    case ORC_RMSE: rmse_b(qo, qs, qs_b, n, j_imd_b); j_imd_b = 0.f; break;
                    case ORC_LOGARITHMIC: logarithmic_b(qo, qs, qs_b, n, j_imd_b); j_imd_b = 0.f; break;
                    default: break;
                }
            }
            for (int i = 0; i < n; ++i)
                qsim_b[g + (long)ng * (s0 + i)] = qsim_b[g + (long)ng * (s0 + i)] + cfg->dt * 1e3f * qs_b[i] / area[g];
        }
    }
    free(qo); free(qs); free(qs_b); free(arr); free(arr_b);
    return 0;
}

/* mwd_cost.f90:1180-1221 */
static float reg_prior(const int* optim, int nf, long n2, const float* m, const float* mb) {
    float res = 0.f;
    for (int i = 0; i < nf; ++i)
        if (optim[i] > 0)
            for (long c = 0; c < n2; ++c) res = res + powf(m[i * n2 + c] - mb[i * n2 + c], 2.f);
    return res;
}
/* forward_db.f90:5756-5799 */
static void reg_prior_b(const int* optim, int nf, long n2, const float* m, const float* mb, float* m_b, float res_b) {
    for (int i = nf - 1; i >= 0; --i)
        if (optim[i] > 0)
            for (long c = n2 - 1; c >= 0; --c) m_b[i * n2 + c] = m_b[i * n2 + c] + 2.f * (m[i * n2 + c] - mb[i * n2 + c]) * res_b;
}
static void smoothing_bounds(const int* active, int nrow, int ncol, int row, int col, int* mnc, int* mxc, int* mnr, int* mxr) {
    int min_col = col - 1 > 0 ? col - 1 : 0, max_col = col + 1 < ncol - 1 ? col + 1 : ncol - 1;
    int min_row = row - 1 > 0 ? row - 1 : 0, max_row = row + 1 < nrow - 1 ? row + 1 : nrow - 1;
    if (active[row + (long)min_col * nrow] == 0) min_col = col;
    if (active[row + (long)max_col * nrow] == 0) max_col = col;
    if (active[min_row + (long)col * nrow] == 0) min_row = row;
    if (active[max_row + (long)col * nrow] == 0) max_row = row;
    *mnc = min_col; *mxc = max_col; *mnr = min_row; *mxr = max_row;
}
/* mwd_cost.f90:1100-1178 */
static float reg_smoothing(const int* active, int nrow, int ncol, const int* optim, int nf, const float* m,
                           const float* mb, int rel) {
    const long n2 = (long)nrow * ncol;
    float res = 0.f;
#define MAT(r, c, i) (rel ? (m[(i) * n2 + (r) + (long)(c) * nrow] - mb[(i) * n2 + (r) + (long)(c) * nrow]) : m[(i) * n2 + (r) + (long)(c) * nrow])
    for (int i = 0; i < nf; ++i)
        if (optim[i] > 0)
            for (int col = 0; col < ncol; ++col)
                for (int row = 0; row < nrow; ++row)
                    if (active[row + (long)col * nrow] == 1) {
                        int mnc, mxc, mnr, mxr;
                        smoothing_bounds(active, nrow, ncol, row, col, &mnc, &mxc, &mnr, &mxr);
                        res = res + (powf(MAT(mxr, col, i) - 2.f * MAT(row, col, i) + MAT(mnr, col, i), 2.f) +
                                     powf(MAT(row, mxc, i) - 2.f * MAT(row, col, i) + MAT(row, mnc, i), 2.f));
                    }
    return res;
}
/* forward_db.f90:5504-5657 */
static void reg_smoothing_b(const int* active, int nrow, int ncol, const int* optim, int nf, const float* m,
                            const float* mb, int rel, float* m_b, float res_b) {
    const long n2 = (long)nrow * ncol;
    float* mat_b = (float*)calloc((size_t)(nf * n2), sizeof(float));
#define MB(r, c, i) mat_b[(i) * n2 + (r) + (long)(c) * nrow]
    for (int i = nf - 1; i >= 0; --i)
        if (optim[i] > 0)
            for (int col = ncol - 1; col >= 0; --col)
                for (int row = nrow - 1; row >= 0; --row)
                    if (active[row + (long)col * nrow] == 1) {
                        int mnc, mxc, mnr, mxr;
                        smoothing_bounds(active, nrow, ncol, row, col, &mnc, &mxc, &mnr, &mxr);
                        float temp_b = 2.f * (MAT(mxr, col, i) - 2.f * MAT(row, col, i) + MAT(mnr, col, i)) * res_b;
                        float temp_b0 = 2.f * (MAT(row, mxc, i) - 2.f * MAT(row, col, i) + MAT(row, mnc, i)) * res_b;
                        MB(row, mxc, i) = MB(row, mxc, i) + temp_b0;
                        MB(row, col, i) = MB(row, col, i) - 2.f * temp_b0;
                        MB(row, mnc, i) = MB(row, mnc, i) + temp_b0;
                        MB(mxr, col, i) = MB(mxr, col, i) + temp_b;
                        MB(row, col, i) = MB(row, col, i) - 2.f * temp_b;
                        MB(mnr, col, i) = MB(mnr, col, i) + temp_b;
                    }
    for (long k = 0; k < nf * n2; ++k) m_b[k] = m_b[k] + mat_b[k];
    free(mat_b);
#undef MB
#undef MAT
}

/* mwd_cost.f90:159-245 (compute_jreg); with P_b/S_b != NULL also forward_db.f90:2927-3092 */
static float compute_jreg(const orc_config* cfg, const int* active, const float* P, const float* Pb,
                          const float* S, const float* Sb, float jreg_b, float* P_b, float* S_b) {
    const long n2 = (long)cfg->nrow * cfg->ncol;
    float pj = 0.f, sj = 0.f;
    for (int i = 0; i < cfg->njr; ++i) {
        float w = cfg->wjreg_fun[i];
        switch (cfg->jreg_fun[i]) {
            case ORC_PRIOR:
                pj = pj + w * reg_prior(cfg->optim_parameters, ORC_GNP, n2, P, Pb);
                sj = sj + w * reg_prior(cfg->optim_states, ORC_GNS, n2, S, Sb);
                break;
            case ORC_SMOOTHING:
            case ORC_HARD_SMOOTHING: {
                int rel = cfg->jreg_fun[i] == ORC_SMOOTHING;
                pj = pj + powf(w, 2.f) * reg_smoothing(active, cfg->nrow, cfg->ncol, cfg->optim_parameters, ORC_GNP, P, Pb, rel);
                sj = sj + powf(w, 2.f) * reg_smoothing(active, cfg->nrow, cfg->ncol, cfg->optim_states, ORC_GNS, S, Sb, rel);
            } break;
            default: break;
        }
    }
    if (P_b && S_b) {
        for (int i = cfg->njr - 1; i >= 0; --i) {
            float w = cfg->wjreg_fun[i];
            switch (cfg->jreg_fun[i]) {
                case ORC_PRIOR:
                    reg_prior_b(cfg->optim_states, ORC_GNS, n2, S, Sb, S_b, w * jreg_b);
                    reg_prior_b(cfg->optim_parameters, ORC_GNP, n2, P, Pb, P_b, w * jreg_b);
                    break;
                case ORC_SMOOTHING:
                case ORC_HARD_SMOOTHING: {
                    int rel = cfg->jreg_fun[i] == ORC_SMOOTHING;
                    reg_smoothing_b(active, cfg->nrow, cfg->ncol, cfg->optim_states, ORC_GNS, S, Sb, rel, S_b, powf(w, 2.f) * jreg_b);
                    reg_smoothing_b(active, cfg->nrow, cfg->ncol, cfg->optim_parameters, ORC_GNP, P, Pb, rel, P_b, powf(w, 2.f) * jreg_b);
                } break;
                default: break;
            }
        }
    }
    return pj + sj;
}

/* mwd_cost.f90:247-306 (compute_cost); adjoint forward_db.f90:3252-3400 */
static int compute_cost(const orc_config* cfg, const int* flwacc, const int* active, const int* gauge_pos,
                        const float* area, const float* qobs, const float* wgauge, float* P, const float* Pb,
                        float* S, const float* Sb, const float* qsim, float* costs, int adjoint, float cost_b,
                        float* qsim_b, float* P_b, float* S_b) {
    const long n2 = (long)cfg->nrow * cfg->ncol;
    float jobs = 0.f, jreg = 0.f;
    if (compute_jobs(cfg, flwacc, gauge_pos, area, qobs, wgauge, qsim, &jobs, cost_b, adjoint ? qsim_b : 0)) return -1;
    if (cfg->denormalize_forward) {
        normalize(P, n2, ORC_GNP, cfg->lb_parameters, cfg->ub_parameters);
        normalize(S, n2, ORC_GNS, cfg->lb_states, cfg->ub_states);
    }
    jreg = compute_jreg(cfg, active, P, Pb, S, Sb, cfg->wjreg * cost_b, adjoint ? P_b : 0, adjoint ? S_b : 0);
    if (cfg->denormalize_forward) {
        if (adjoint) { /* NORMALIZE_*_B: forward_db.f90:809-889, 1877-1900 */
            for (int i = 0; i < ORC_GNS; ++i)
                for (long c = 0; c < n2; ++c) S_b[i * n2 + c] = S_b[i * n2 + c] / (cfg->ub_states[i] - cfg->lb_states[i]);
            for (int i = 0; i < ORC_GNP; ++i)
                for (long c = 0; c < n2; ++c) P_b[i * n2 + c] = P_b[i * n2 + c] / (cfg->ub_parameters[i] - cfg->lb_parameters[i]);
        }
        denormalize(P, n2, ORC_GNP, cfg->lb_parameters, cfg->ub_parameters);
        denormalize(S, n2, ORC_GNS, cfg->lb_states, cfg->ub_states);
    }
    costs[0] = jobs + cfg->wjreg * jreg;
    costs[1] = jobs;
    costs[2] = jreg;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* base_forward: smash/solver/forward/forward.f90:1-80                                          */
/* ------------------------------------------------------------------------------------------- */
int orc_forward(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                const int* active, const int* gauge_pos, const float* area, const float* prcp,
                const float* pet, const float* qobs, const float* wgauge, float* P, const float* Pb,
                float* S, const float* Sb, float* qsim, float* costs, float* fstates) {
    const long n2 = (long)cfg->nrow * cfg->ncol;
    if (cfg->structure < ORC_GR_A || cfg->structure > ORC_VIC_A) return -2;
    if (cfg->denormalize_forward) {
        denormalize(P, n2, ORC_GNP, cfg->lb_parameters, cfg->ub_parameters);
        denormalize(S, n2, ORC_GNS, cfg->lb_states, cfg->ub_states);
    }
    float* S_imd = (float*)malloc(sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(S_imd, S, sizeof(float) * (size_t)(ORC_GNS * n2));
    structure_forward(cfg, flwdir, flwacc, path, active, gauge_pos, prcp, pet, P, S, qsim, 0);
    if (fstates) memcpy(fstates, S, sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(S, S_imd, sizeof(float) * (size_t)(ORC_GNS * n2));
    free(S_imd);
    return compute_cost(cfg, flwacc, active, gauge_pos, area, qobs, wgauge, P, Pb, S, Sb, qsim, costs, 0, 0.f, 0, 0, 0);
}

/* ------------------------------------------------------------------------------------------- */
/* local adjoints: forward_db.f90 MD_GR_OPERATOR_DIFF / MD_ROUTING_OPERATOR_DIFF                */
/* ------------------------------------------------------------------------------------------- */
/* forward_db.f90:5878-5925 */
static void gr_interception_b(float prcp, float pet, float ci, float* ci_b, float hi, float* hi_b, float* pn_b, float* ei_b) {
    float ei, pn;
    int br_ei, br_pn;
    if (pet > prcp + hi * ci) { ei = prcp + hi * ci; br_ei = 0; } else { ei = pet; br_ei = 1; }
    if (0.f < prcp - ci * (1.f - hi) - ei) { pn = prcp - ci * (1.f - hi) - ei; br_pn = 0; } else { pn = 0.f; br_pn = 1; }
    float temp_b = *hi_b / ci;
    *ei_b = *ei_b - temp_b;
    *pn_b = *pn_b - temp_b;
    *ci_b = *ci_b - (prcp - ei - pn) * temp_b / ci;
    if (br_pn == 0) {
        *ci_b = *ci_b - (1.f - hi) * (*pn_b);
        *hi_b = *hi_b + ci * (*pn_b);
        *ei_b = *ei_b - *pn_b;
    }
    if (br_ei == 0) {
        *hi_b = *hi_b + ci * (*ei_b);
        *ci_b = *ci_b + hi * (*ei_b);
    }
}

/* forward_db.f90:6012-6103.  hp = pre-step level; outputs pn_b, en_b; updates hp_b, cp_b. */
static void gr_production_b(float pn, float* pn_b, float en, float* en_b, float cp, float* cp_b, float beta,
                            float hp, float* hp_b, float pr_b, float perc_b) {
    float inv_cp = 1.f / cp;
    float ps = cp * (1.f - hp * hp) * tanhf(pn * inv_cp) / (1.f + hp * tanhf(pn * inv_cp));
    float es = hp * cp * (2.f - hp) * tanhf(en * inv_cp) / (1.f + (1.f - hp) * tanhf(en * inv_cp));
    float hp_imd = hp + (ps - es) * inv_cp;
    float r = hp_imd / beta, r2 = r * r;
    float pwx1 = 1.f + r2 * r2;
    float pwr1 = powf(pwx1, -0.25f);
    float perc = hp_imd * cp * (1.f - pwr1);
    perc_b = perc_b - inv_cp * (*hp_b);
    float inv_cp_b = -(perc * (*hp_b));
    *cp_b = *cp_b + hp_imd * (1.f - pwr1) * perc_b;
    float pwr1_b = -(hp_imd * cp * perc_b);
    float pwx1_b = -(0.25f * powf(pwx1, -1.25f) * pwr1_b);
    float b2 = beta * beta;
    float hp_imd_b = *hp_b + cp * (1.f - pwr1) * perc_b + 4.f * (hp_imd * hp_imd * hp_imd) * pwx1_b / (b2 * b2);
    if (pn > 0.f) {
        *pn_b = pr_b;
        hp_imd_b = hp_imd_b - cp * pr_b;
        *hp_b = cp * pr_b;
        *cp_b = *cp_b - (hp_imd - hp) * pr_b;
    } else {
        *hp_b = 0.f;
        *pn_b = 0.f;
    }
    float es_b = -(inv_cp * hp_imd_b);
    float temp4 = tanhf(en * inv_cp);
    float temp3 = (-hp + 1.f) * temp4 + 1.f;
    float temp1 = tanhf(en * inv_cp);
    float temp0 = hp * cp * (-hp + 2.f);
    float temp_b3 = es_b / temp3;
    float temp_b = (2.f - hp) * temp1 * temp_b3;
    float temp_b0 = -(temp0 * temp1 * temp_b3 / temp3);
    *hp_b = *hp_b + hp_imd_b + cp * temp_b - hp * cp * temp1 * temp_b3 - temp4 * temp_b0;
    float ps_b = inv_cp * hp_imd_b;
    float th = tanhf(en * inv_cp);
    float temp_b4 = (1.0f - th * th) * temp0 * temp_b3;
    float temp_b5 = (1.0f - th * th) * (1.f - hp) * temp_b0;
    *en_b = inv_cp * temp_b5 + inv_cp * temp_b4;
    *cp_b = *cp_b + hp * temp_b;
    float temp = tanhf(pn * inv_cp);
    temp0 = hp * temp + 1.f;
    temp1 = tanhf(pn * inv_cp);
    float temp2 = cp * (-(hp * hp) + 1.f);
    temp_b = ps_b / temp0;
    th = tanhf(pn * inv_cp);
    temp_b0 = (1.0f - th * th) * temp2 * temp_b;
    float temp_b1 = -(temp2 * temp1 * temp_b / temp0);
    *hp_b = *hp_b + temp * temp_b1 - 2.f * hp * cp * temp1 * temp_b;
    float temp_b2 = (1.0f - th * th) * hp * temp_b1;
    inv_cp_b = inv_cp_b + (ps - es) * hp_imd_b + en * temp_b5 + en * temp_b4 + pn * temp_b2 + pn * temp_b0;
    *cp_b = *cp_b + (1.f - hp * hp) * temp1 * temp_b - inv_cp_b / (cp * cp);
    *pn_b = *pn_b + inv_cp * temp_b2 + inv_cp * temp_b0;
}

/* forward_db.f90:6147-6157 */
static void gr_exchange_b(float exc, float* exc_b, float hft, float* hft_b, float l_b) {
    *exc_b = *exc_b + powf(hft, 3.5f) * l_b;
    *hft_b = *hft_b + 3.5f * powf(hft, 2.5f) * exc * l_b;
}

/* forward_db.f90:6275-6412.  ht = pre-step level. */
static void gr_transfer_b(float n, float prcp, float pr, float* pr_b, float ct, float* ct_b, float ht,
                          float* ht_b, float q_b) {
    float nm1 = n - 1.f, d1pnm1 = 1.f / nm1;
    float pr_imd, pwx1 = 0.f, pwx3 = 0.f;
    int br_gap, br_max;
    if (prcp < 0.f) {
        pwx1 = ht * ct;
        float pwr1 = powf(pwx1, -nm1), pwr2 = powf(ct, -nm1);
        pwx3 = pwr1 - pwr2;
        pr_imd = powf(pwx3, -d1pnm1) - ht * ct;
        br_gap = 1;
    } else { pr_imd = pr; br_gap = 0; }
    float ht_imd;
    if (1.e-6f < ht + pr_imd / ct) { ht_imd = ht + pr_imd / ct; br_max = 0; } else { ht_imd = 1.e-6f; br_max = 1; }
    float g_pwx1 = pwx1, g_pwx3 = pwx3; /* the values Tapenade pushes/pops around the second block */
    pwx1 = ht_imd * ct;
    float pwy1 = -nm1, pwy2 = -nm1, pwy3 = -d1pnm1;
    float pwr1 = powf(pwx1, pwy1);
    float pwr2 = powf(ct, pwy2);
    pwx3 = pwr1 + pwr2;
    float pwr3 = powf(pwx3, pwy3);
    float ht_new = pwr3 / ct;
    float htb = *ht_b - ct * q_b;
    float pwr3_b = htb / ct;
    float pwx3_b = (pwx3 <= 0.f && (pwy3 == 0.f || pwy3 != (float)(int)pwy3)) ? 0.f : pwy3 * powf(pwx3, pwy3 - 1.f) * pwr3_b;
    float pwr1_b = pwx3_b, pwr2_b = pwx3_b;
    float pwx1_b = (pwx1 <= 0.f && (pwy1 == 0.f || pwy1 != (float)(int)pwy1)) ? 0.f : pwy1 * powf(pwx1, pwy1 - 1.f) * pwr1_b;
    float ht_imd_b = ct * q_b + ct * pwx1_b;
    if (ct <= 0.f && (pwy2 == 0.f || pwy2 != (float)(int)pwy2))
        *ct_b = *ct_b + (ht_imd - ht_new) * q_b + ht_imd * pwx1_b - pwr3 * htb / (ct * ct);
    else
        *ct_b = *ct_b + (ht_imd - ht_new) * q_b + pwy2 * powf(ct, pwy2 - 1.f) * pwr2_b - pwr3 * htb / (ct * ct) + ht_imd * pwx1_b;
    float pr_imd_b;
    if (br_max == 0) {
        htb = ht_imd_b;
        pr_imd_b = ht_imd_b / ct;
        *ct_b = *ct_b - pr_imd * ht_imd_b / (ct * ct);
    } else { htb = 0.f; pr_imd_b = 0.f; }
    if (br_gap == 0) {
        *pr_b = pr_imd_b;
    } else {
        pwx1 = g_pwx1; pwx3 = g_pwx3;
        pwr3_b = pr_imd_b;
        pwx3_b = (pwx3 <= 0.f && (pwy3 == 0.f || pwy3 != (float)(int)pwy3)) ? 0.f : pwy3 * powf(pwx3, pwy3 - 1.f) * pwr3_b;
        pwr1_b = pwx3_b; pwr2_b = -pwx3_b;
        pwx1_b = (pwx1 <= 0.f && (pwy1 == 0.f || pwy1 != (float)(int)pwy1)) ? 0.f : pwy1 * powf(pwx1, pwy1 - 1.f) * pwr1_b;
        htb = htb + ct * pwx1_b - ct * pr_imd_b;
        if (ct <= 0.f && (pwy2 == 0.f || pwy2 != (float)(int)pwy2))
            *ct_b = *ct_b + ht * pwx1_b - ht * pr_imd_b;
        else
            *ct_b = *ct_b + pwy2 * powf(ct, pwy2 - 1.f) * pwr2_b - ht * pr_imd_b + ht * pwx1_b;
        *pr_b = 0.f;
    }
    *ht_b = htb;
}

/* forward_db.f90:6628-6652.  hr = pre-step level. */
static void linear_routing_b(float dt, float qup, float* qup_b, float lr, float* lr_b, float hr, float* hr_b, float qrout_b) {
    float hr_imd = hr + qup;
    float arg1 = -(dt / (lr * 60.f));
    *hr_b = *hr_b - qrout_b;
    float hr_imd_b = qrout_b + expf(arg1) * (*hr_b);
    float arg1_b = expf(arg1) * hr_imd * (*hr_b);
    *lr_b = *lr_b + dt * arg1_b / ((lr * lr) * 60.f);
    *hr_b = hr_imd_b;
    *qup_b = hr_imd_b;
}

/* forward_db.f90:6520-6564 */
static void upstream_discharge_b(float dt, float dx, int nrow, int ncol, const int* flwdir, const int* flwacc,
                                 int row, int col, float* q_b, float qup_b) {
    static const int dcol[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    static const int drow[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    if (flwacc[row + (long)col * nrow] > 1) {
        qup_b = dt * qup_b / (0.001f * (dx * dx) * (float)(flwacc[row + (long)col * nrow] - 1));
        for (int i = 7; i >= 0; --i) {
            int c = col + dcol[i], r = row + drow[i];
            if (c >= 0 && c < ncol && r >= 0 && r < nrow)
                if (flwdir[r + (long)c * nrow] == i + 1) q_b[r + (long)c * nrow] = q_b[r + (long)c * nrow] + qup_b;
        }
    }
}

/* reverse sweep of GR_{A,B,C,D}_FORWARD_B: forward_db.f90:8102-8176 / 8648-8723 / 9206-9293 / 9736-9797 */
static void structure_reverse(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                              const int* active, const int* gauge_pos, const float* P, const float* tape,
                              float* qsim_b, float* P_b, float* S_b) {
    const int nrow = cfg->nrow, ncol = cfg->ncol, st = cfg->structure, ng = cfg->ng;
    const long n2 = (long)nrow * ncol;
    const float dt = cfg->dt, dx = cfg->dx;
    float* q_b = (float*)calloc((size_t)n2, sizeof(float));
    for (int t = cfg->nt - 1; t >= 0; --t) {
        for (int g = ng - 1; g >= 0; --g) {
            long gc = gauge_pos[g] + (long)gauge_pos[g + ng] * nrow;
            q_b[gc] = q_b[gc] + qsim_b[g + (long)ng * t];
            qsim_b[g + (long)ng * t] = 0.f;
        }
        for (long i = n2 - 1; i >= 0; --i) {
            int row = path[2 * i], col = path[2 * i + 1];
            if (row < 0 || col < 0) continue;
            long c = row + (long)col * nrow;
            if (active[c] != 1) continue;
            const float* tp = tape + 8 * (i + n2 * t);
            float hi = tp[0], hp = tp[1], hft = tp[2], hst = tp[3], hlr = tp[4], qup = tp[5], prcp = tp[6], pet = tp[7];
            const int wet = (prcp >= 0.f && pet >= 0.f);
            if (st == ORC_VIC_A) {   /* VIC_A_FORWARD_B, forward_db.f90:10246-10329 */
                const float husl1_0 = tp[0], husl2_0 = tp[1], hlsl_0 = tp[2];
                float h1 = husl1_0, h2 = husl2_0, hl = hlsl_0, runoff = 0.f;
                const float cusl1 = P[P_CUSL1 * n2 + c], cusl2 = P[P_CUSL2 * n2 + c], clsl = P[P_CLSL * n2 + c];
                float h1_1 = h1, h2_1 = h2;                       /* after infiltration */
                if (wet) {
                    vic_infiltration(prcp, cusl1, cusl2, P[P_B * n2 + c], &h1, &h2, &runoff);
                    h1_1 = h1; h2_1 = h2;
                    vic_vertical_transfer(pet, cusl1, cusl2, clsl, P[P_KS * n2 + c], &h1, &h2, &hl);
                }
                const float h2_2 = h2, hl_2 = hl;                 /* on entry of interflow / baseflow */
                float temp_b = (dx * dx) * 0.001f * q_b[c] / dt;
                q_b[c] = 0.f;
                float qt_b = temp_b;
                float qrout_b = (float)(flwacc[c] - 1) * temp_b;
                float qup_b = 0.f;
                linear_routing_b(dt, qup, &qup_b, P[P_LR * n2 + c], &P_b[P_LR * n2 + c], hlr, &S_b[S_HLR * n2 + c], qrout_b);
                upstream_discharge_b(dt, dx, nrow, ncol, flwdir, flwacc, row, col, q_b, qup_b);
                vic_baseflow_b(clsl, &P_b[P_CLSL * n2 + c], P[P_DS * n2 + c], &P_b[P_DS * n2 + c], P[P_DSM * n2 + c], &P_b[P_DSM * n2 + c],
                               P[P_WS * n2 + c], &P_b[P_WS * n2 + c], hl_2, &S_b[S_HLSL * n2 + c], qt_b);
                vic_interflow_b(5.f, cusl2, &P_b[P_CUSL2 * n2 + c], h2_2, &S_b[S_HUSL2 * n2 + c], qt_b);
                if (wet) {
                    vic_vertical_transfer_b(pet, cusl1, &P_b[P_CUSL1 * n2 + c], cusl2, &P_b[P_CUSL2 * n2 + c], clsl, &P_b[P_CLSL * n2 + c],
                                            P[P_KS * n2 + c], &P_b[P_KS * n2 + c], h1_1, &S_b[S_HUSL1 * n2 + c], h2_1, &S_b[S_HUSL2 * n2 + c],
                                            hlsl_0, &S_b[S_HLSL * n2 + c]);
                    vic_infiltration_b(prcp, cusl1, &P_b[P_CUSL1 * n2 + c], cusl2, &P_b[P_CUSL2 * n2 + c], P[P_B * n2 + c],
                                       &P_b[P_B * n2 + c], husl1_0, &S_b[S_HUSL1 * n2 + c], husl2_0, &S_b[S_HUSL2 * n2 + c], qt_b);
                }
                continue;
            }
            /* recompute the primal intermediates the reference pops from its tape */
            float ei = 0.f, pn = 0.f, en = 0.f, pr = 0.f, perc = 0.f, l = 0.f, prr, prl = 0.f, prd = 0.f;
            if (wet) {
                float hi2 = hi, hp2 = hp;
                if (st == ORC_GR_A || st == ORC_GR_D) { ei = fminf(pet, prcp); pn = fmaxf(0.f, prcp - ei); }
                else gr_interception(prcp, pet, P[P_CI * n2 + c], &hi2, &pn, &ei);
                en = pet - ei;
                gr_production(pn, en, P[P_CP * n2 + c], 1000.f, &hp2, &pr, &perc);
                if (st != ORC_GR_D) gr_exchange(P[P_EXC * n2 + c], hft, &l);
            }
            if (st == ORC_GR_A || st == ORC_GR_B) { prr = 0.9f * (pr + perc) + l; prd = 0.1f * (pr + perc); }
            else if (st == ORC_GR_C) { prr = 0.9f * 0.6f * (pr + perc) + l; prl = 0.9f * 0.4f * (pr + perc); prd = 0.1f * (pr + perc); }
            else prr = pr + perc;

            float temp_b = (dx * dx) * 0.001f * q_b[c] / dt;
            q_b[c] = 0.f;
            float qt_b = temp_b;
            float qrout_b = (float)(flwacc[c] - 1) * temp_b;
            float qup_b = 0.f;
            linear_routing_b(dt, qup, &qup_b, P[P_LR * n2 + c], &P_b[P_LR * n2 + c], hlr, &S_b[S_HLR * n2 + c], qrout_b);
            upstream_discharge_b(dt, dx, nrow, ncol, flwdir, flwacc, row, col, q_b, qup_b);
            float qr_b = qt_b, ql_b = qt_b, qd_b = qt_b;
            float prd_b = 0.f, l_b = 0.f, prr_b = 0.f, prl_b = 0.f, pr_b, perc_b;
            if (st != ORC_GR_D) {
                if (0.f < prd + l) { prd_b = qd_b; l_b = qd_b; }
            }
            if (st == ORC_GR_C)
                gr_transfer_b(5.f, prcp, prl, &prl_b, P[P_CST * n2 + c], &P_b[P_CST * n2 + c], hst, &S_b[S_HST * n2 + c], ql_b);
            gr_transfer_b(5.f, prcp, prr, &prr_b, P[P_CFT * n2 + c], &P_b[P_CFT * n2 + c], hft, &S_b[S_HFT * n2 + c], qr_b);
            if (st == ORC_GR_A || st == ORC_GR_B) {
                pr_b = 0.1f * prd_b + 0.9f * prr_b;
                perc_b = 0.1f * prd_b + 0.9f * prr_b;
                l_b = l_b + prr_b;
            } else if (st == ORC_GR_C) {
                float tb = 0.4f * 0.9f * prl_b;
                pr_b = 0.1f * prd_b + tb;
                perc_b = 0.1f * prd_b + tb;
                tb = 0.6f * 0.9f * prr_b;
                l_b = l_b + prr_b;
                pr_b = pr_b + tb;
                perc_b = perc_b + tb;
            } else { pr_b = prr_b; perc_b = prr_b; }
            if (wet) {
                float pn_b = 0.f, en_b = 0.f;
                if (st != ORC_GR_D)
                    gr_exchange_b(P[P_EXC * n2 + c], &P_b[P_EXC * n2 + c], hft, &S_b[S_HFT * n2 + c], l_b);
                gr_production_b(pn, &pn_b, en, &en_b, P[P_CP * n2 + c], &P_b[P_CP * n2 + c], 1000.f, hp, &S_b[S_HP * n2 + c], pr_b, perc_b);
                if (st == ORC_GR_B || st == ORC_GR_C) {
                    float ei_b = -en_b;
                    gr_interception_b(prcp, pet, P[P_CI * n2 + c], &P_b[P_CI * n2 + c], hi, &S_b[S_HI * n2 + c], &pn_b, &ei_b);
                }
            }
        }
    }
    free(q_b);
}

/* BASE_FORWARD_B: forward_db.f90:10648-10936 */
int orc_forward_b(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                  const int* active, const int* gauge_pos, const float* area, const float* prcp,
                  const float* pet, const float* qobs, const float* wgauge, float* P, const float* Pb,
                  float* S, const float* Sb, float cost_b, float* qsim, float* costs, float* P_b, float* S_b) {
    const long n2 = (long)cfg->nrow * cfg->ncol;
    if (cfg->structure < ORC_GR_A || cfg->structure > ORC_VIC_A) return -2;
    if (cfg->denormalize_forward) {
        denormalize(P, n2, ORC_GNP, cfg->lb_parameters, cfg->ub_parameters);
        denormalize(S, n2, ORC_GNS, cfg->lb_states, cfg->ub_states);
    }
    float* S_imd = (float*)malloc(sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(S_imd, S, sizeof(float) * (size_t)(ORC_GNS * n2));
    float* tape = (float*)malloc(sizeof(float) * 8 * (size_t)n2 * (size_t)cfg->nt);
    float* qsim_b = (float*)calloc((size_t)cfg->ng * cfg->nt + 1, sizeof(float));
    if (!tape || !qsim_b) return -3;
    structure_forward(cfg, flwdir, flwacc, path, active, gauge_pos, prcp, pet, P, S, qsim, tape);
    memcpy(S, S_imd, sizeof(float) * (size_t)(ORC_GNS * n2));
    free(S_imd);
    memset(P_b, 0, sizeof(float) * (size_t)(ORC_GNP * n2));
    memset(S_b, 0, sizeof(float) * (size_t)(ORC_GNS * n2));
    /* BASE_FORWARD_B pushes/pops parameters and states around COMPUTE_COST / COMPUTE_COST_B
     * (forward_db.f90:10775-10868, 3275-3330), so unlike base_forward they come back denormalised
     * WITHOUT the normalise->denormalise round trip.  (Quirk not restated: the reference leaves the
     * inactive fields beta and alpha normalised because Tapenade does not push them.) */
    float* P_keep = (float*)malloc(sizeof(float) * (size_t)(ORC_GNP * n2));
    float* S_keep = (float*)malloc(sizeof(float) * (size_t)(ORC_GNS * n2));
    memcpy(P_keep, P, sizeof(float) * (size_t)(ORC_GNP * n2));
    memcpy(S_keep, S, sizeof(float) * (size_t)(ORC_GNS * n2));
    int rc = compute_cost(cfg, flwacc, active, gauge_pos, area, qobs, wgauge, P, Pb, S, Sb, qsim, costs, 1, cost_b, qsim_b, P_b, S_b);
    memcpy(P, P_keep, sizeof(float) * (size_t)(ORC_GNP * n2));
    memcpy(S, S_keep, sizeof(float) * (size_t)(ORC_GNS * n2));
    free(P_keep); free(S_keep);
    if (rc == 0) {
        structure_reverse(cfg, flwdir, flwacc, path, active, gauge_pos, P, tape, qsim_b, P_b, S_b);
        if (cfg->denormalize_forward) { /* DENORMALIZE_*_B: forward_db.f90:967-1055, 1959-2012 */
            for (int i = 0; i < ORC_GNS; ++i)
                for (long c = 0; c < n2; ++c) S_b[i * n2 + c] = (cfg->ub_states[i] - cfg->lb_states[i]) * S_b[i * n2 + c];
            for (int i = 0; i < ORC_GNP; ++i)
                for (long c = 0; c < n2; ++c) P_b[i * n2 + c] = (cfg->ub_parameters[i] - cfg->lb_parameters[i]) * P_b[i * n2 + c];
        }
    }
    free(tape); free(qsim_b);
    return rc;
}
