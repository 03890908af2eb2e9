// sx_cost.h -- gauge-discharge cost and its adjoint seeds on the device.
//
//   compute_jobs     smash/solver/optimize/mwd_cost.f90:37-156   (nse :350, kge :403-490, se :492, rmse :521, logarithmic :558)
//   COMPUTE_JOBS_B   smash/solver/forward/forward_db.f90:2553-2715 (NSE_B :3505, KGE_B :3657-3829, SE_B :3868, RMSE_B :3936, LOGARITHMIC_B :4025)
//
// The reference accumulates every sum sequentially in fp32 over time.  A tree reduction would move
// the cost by ~1e-6 relative, i.e. by the whole parity budget, so each gauge is reduced by ONE
// wavefront that loads 64 consecutive steps coalesced and folds them in time order through
// cross-lane broadcasts (v_readlane): wave-wide loads, reference summation order.  The seeds
// qsim_b(g,t) are then independent per (g,t) and computed fully in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include "sx_math.h"

#define SX_MAXJF 8

struct SxGaugeSums {   // per gauge, reference order sums
    int n;
    float sum_x, sum_y, sum_xx, sum_yy, sum_xy, se, lg;
};

struct SxCostCoef {    // per (gauge, cost function): what the per-element adjoint needs
    int kind;          // 0 none, 1 nse "x*c_xy + 2*y*c_yy", 4 kge "... + c_y", 2 se-like "-2*(x-y)*c", 3 logarithmic
    float c_xy, c_yy, c_y, c;
};

struct SxCostArgs {
    int ng, nt, s0;                 // s0 = optimize_start_step - 1
    int njf;
    int jobs_fun[SX_MAXJF];
    float wjobs_fun[SX_MAXJF];
    float dt, dx;
    const float* qg;                // [ngc][nt]
    float* qgb;                     // [ngc][nt]
    int ngc;
    const int* gauge_gid;           // [ng] gauge -> gauge-cell id
    const int* gauge_flwacc;        // [ng]
    const float* area;              // [ng]
    const float* wgauge;            // [ng]
    const float* qobs;              // [ng][nt]  (time fastest)
    float* qsim_b;                  // [ng][nt]
    SxGaugeSums* sums;              // [ng]
    SxCostCoef* coef;               // [ng][SX_MAXJF]
    float* out;                     // [0] = jobs
    float jobs_b;
    float* med;                     // [2 ng] gauge_jobs of the negative-weight gauges, then their adjoint weights
    int* med_idx;                   // [2 ng] sort permutation, then gauge of each entry
    // tiled plans (smashx_set_median_slots): the negative-weight gauges of the whole decomposition, in global gauge order
    int nslots;                     // 0: every gauge of the median is on this plan (the path above)
    const int* slot;                // [ng] slot of a local negative-weight gauge, else -1
    float* medx;                    // [3 nslots]: gauge_jobs of every slot (summed over the ranks between the two phases), sorted copy, weights
    int* medx_idx;                  // [nslots] sort permutation
};

__device__ __forceinline__ float sx_qs(const SxCostArgs& C, int g, int t) {
    return C.qg[(size_t)C.gauge_gid[g] * C.nt + t] * C.dt / C.area[g] * 1e3f;          // mwd_cost.f90:84-85
}
__device__ __forceinline__ float sx_qo(const SxCostArgs& C, int g, int t) {
    return C.qobs[(size_t)g * C.nt + t] * C.dt / ((float)C.gauge_flwacc[g] * C.dx * C.dx) * 1e3f;   // :90-92
}

// one wavefront per gauge
__global__ __launch_bounds__(64) void sx_k_cost_sums(SxCostArgs C) {
    SX_LIBM_INIT();
    const int g = blockIdx.x, lane = threadIdx.x;
    SxGaugeSums S; S.n = 0; S.sum_x = S.sum_y = S.sum_xx = S.sum_yy = S.sum_xy = S.se = S.lg = 0.f;
    const float w = C.wgauge[g];
    if (w > 0.f || w < 0.f) {
        bool want_lg = false;
        for (int j = 0; j < C.njf; ++j) want_lg |= (C.jobs_fun[j] == 6);
        for (int tb = C.s0; tb < C.nt; tb += 64) {
            const int t = tb + lane;
            float x = -1.f, y = 0.f;
            if (t < C.nt) { x = sx_qo(C, g, t); y = sx_qs(C, g, t); }
            const int cnt = min(64, C.nt - tb);
            for (int i = 0; i < cnt; ++i) {
                const float xi = __shfl(x, i), yi = __shfl(y, i);
                if (xi >= 0.f) {
                    S.n++;
                    S.sum_x = S.sum_x + xi;
                    S.sum_y = S.sum_y + yi;
                    S.sum_xx = S.sum_xx + (xi * xi);
                    S.sum_yy = S.sum_yy + (yi * yi);
                    S.sum_xy = S.sum_xy + (xi * yi);
                    S.se = S.se + (xi - yi) * (xi - yi);
                }
                if (want_lg && xi > 0.f && yi > 0.f) {
                    const float lgv = sx_logf(yi / xi);
                    S.lg = S.lg + xi * lgv * lgv;
                }
            }
        }
    }
    if (lane == 0) C.sums[g] = S;
}

struct SxKge { float mean_x, mean_y, var_x, var_y, cov, r, a, b; };
__device__ __forceinline__ SxKge sx_kge_components(const SxGaugeSums& S) {
    SxKge k;
    const float n = (float)S.n;
    k.mean_x = S.sum_x / n;
    k.mean_y = S.sum_y / n;
    k.var_x = (S.sum_xx / n) - (k.mean_x * k.mean_x);
    k.var_y = (S.sum_yy / n) - (k.mean_y * k.mean_y);
    k.cov = (S.sum_xy / n) - (k.mean_x * k.mean_y);
    k.r = (k.cov / sqrtf(k.var_x)) / sqrtf(k.var_y);
    k.a = sqrtf(k.var_y) / sqrtf(k.var_x);
    k.b = k.mean_y / k.mean_x;
    return k;
}
__device__ __forceinline__ float sx_kge_value(const SxKge& k) {
    return sqrtf((k.r - 1.f) * (k.r - 1.f) + (k.b - 1.f) * (k.b - 1.f) + (k.a - 1.f) * (k.a - 1.f));
}
__device__ __forceinline__ void sx_kge_coef(const SxGaugeSums& S, const SxKge& k, float res_b, SxCostCoef& c) {
    const float arg1 = (k.r - 1.f) * (k.r - 1.f) + (k.b - 1.f) * (k.b - 1.f) + (k.a - 1.f) * (k.a - 1.f);
    const float arg1_b = (arg1 == 0.f) ? 0.f : res_b / (2.0f * sqrtf(arg1));
    const float r_b = 2.f * (k.r - 1.f) * arg1_b, b_b = 2.f * (k.b - 1.f) * arg1_b, a_b = 2.f * (k.a - 1.f) * arg1_b;
    const float n = (float)S.n;
    const float result1 = sqrtf(k.var_x), result2 = sqrtf(k.var_y);
    const float result1_b = a_b / sqrtf(k.var_x);
    float var_y_b = (k.var_y == 0.f) ? 0.f : result1_b / (2.0f * sqrtf(k.var_y));
    const float temp_b = r_b / (result1 * result2);
    const float cov_b = temp_b;
    const float result2_b = -(k.cov * temp_b / result2);
    if (!(k.var_y == 0.f)) var_y_b = var_y_b + result2_b / (2.0f * sqrtf(k.var_y));
    const float mean_y_b = b_b / k.mean_x - k.mean_x * cov_b - 2.f * k.mean_y * var_y_b;
    c.kind = 4; c.c_xy = cov_b / n; c.c_yy = var_y_b / n; c.c_y = mean_y_b / n; c.c = 0.f;
}

// single thread: per-gauge criteria, weighted sum over gauges in gauge order, adjoint coefficients
// heap_sort (mwd_cost.f90:594-673) with the permutation carried along: its adjoint only moves data back
__device__ inline void sx_heap_sort_idx(int n, float* arr, int* idx) {
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

// the per-criterion adjoint coefficients of one gauge given the seed of its gauge_jobs (COMPUTE_JOBS_B, forward_db.f90:2656-2704);
// j_imd_b is the reference's scalar, carried from gauge to gauge
__device__ inline void sx_cost_gauge_coef(const SxCostArgs& C, int g, float gauge_jobs_b, float& j_imd_b) {
    const SxGaugeSums S = C.sums[g];
    const bool any = S.n > 0;
    const float n = (float)S.n;
    for (int j = C.njf - 1; j >= 0; --j) {
        j_imd_b = j_imd_b + C.wjobs_fun[j] * gauge_jobs_b;
        if (!any) continue;
        SxCostCoef c; c.kind = 0; c.c_xy = c.c_yy = c.c_y = c.c = 0.f;
        switch (C.jobs_fun[j]) {
            case 1: { const float mean_x = S.sum_x / n;
                      const float den = S.sum_xx - n * mean_x * mean_x;
                      const float num_b = j_imd_b / den;
                      c.kind = 1; c.c_yy = num_b; c.c_xy = -(2.f * num_b); c.c_y = 0.f; j_imd_b = 0.f; } break;
            case 2: { const SxKge k = sx_kge_components(S); sx_kge_coef(S, k, j_imd_b, c); j_imd_b = 0.f; } break;
            case 3: { const SxKge k = sx_kge_components(S); const float imd = sx_kge_value(k);
                      sx_kge_coef(S, k, 2.f * imd * j_imd_b, c); j_imd_b = 0.f; } break;
            case 4: c.kind = 2; c.c = j_imd_b; j_imd_b = 0.f; break;
            case 5: { const float result1 = S.se;
                      c.kind = 2; c.c = (result1 / n == 0.f) ? 0.f : j_imd_b / (n * 2.0f * sqrtf(result1 / n)); j_imd_b = 0.f; } break;
            case 6: c.kind = 3; c.c = j_imd_b; j_imd_b = 0.f; break;
            default: break;
        }
        C.coef[g * SX_MAXJF + j] = c;
    }
}

// phase 0: everything (plans that hold every gauge of the median).  Tiled plans with a median over gauges of several ranks run
// phase 1 -- the gauge_jobs of the local negative-weight gauges into their slots of medx -- then sum medx over the ranks (one all-reduce of nslots floats) and run phase 2: the median of ALL slots, this plan's
// share of it (the interpolation weights of its own slots: the shares of the ranks add up to the median) and the seeds of its gauges.
__global__ void sx_k_cost_final(SxCostArgs C, int adjoint, int phase = 0) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float jobs = 0.f;
    int arr_size = 0;
    float* arr = C.med; float* arr_b = C.med + C.ng;
    int* perm = C.med_idx; int* arr_gauge = C.med_idx + C.ng;
    if (phase == 2) {
        // QUANTILE / QUANTILE_B over the slots of the whole decomposition (mwd_cost.f90:154, 675-723; forward_db.f90:4327-4370)
        const int ns = C.nslots;
        float* val = C.medx; float* srt = C.medx + ns; float* wq = C.medx + 2 * ns; int* pm = C.medx_idx;
        for (int i = 0; i < ns; ++i) { srt[i] = val[i]; wq[i] = 0.f; pm[i] = i; }
        if (ns > 1) {
            sx_heap_sort_idx(ns, srt, pm);
            const float frac = (float)(ns - 1) * 0.5f + 1.f;
            if (frac <= 1.f) wq[pm[0]] = 1.f;
            else if (frac >= (float)ns) wq[pm[ns - 1]] = 1.f;
            else { const int k = (int)frac; const float f = frac - (float)k; wq[pm[k]] = wq[pm[k]] + f; wq[pm[k - 1]] = wq[pm[k - 1]] + (1.f - f); }
        } else if (ns == 1) wq[0] = 1.f;
        // the median REPLACES the weighted sum of the positive-weight gauges (mwd_cost.f90:154; phase 0 does the same: jobs = res,
        // jobs_b = 0), so those neither enter the cost nor receive a seed
        jobs = 0.f;
        for (int g = 0; g < C.ng; ++g) { const int sl = C.slot[g]; if (sl >= 0) jobs = jobs + wq[sl] * val[sl]; }
        C.out[0] = jobs;
        if (!adjoint) return;
        // seeds: the positive-weight gauges see jobs_b through their weight; the median's seed reaches a slot through its weight
        float j_imd_b2 = 0.f;
        for (int g = C.ng - 1; g >= 0; --g) {
            for (int j = 0; j < SX_MAXJF; ++j) { SxCostCoef z; z.kind = 0; z.c_xy = z.c_yy = z.c_y = z.c = 0.f; C.coef[g * SX_MAXJF + j] = z; }
            const float w = C.wgauge[g];
            if (!(w > 0.f || w < 0.f)) continue;
            const float gauge_jobs_b = (w > 0.f) ? 0.f : wq[C.slot[g]] * C.jobs_b;
            sx_cost_gauge_coef(C, g, gauge_jobs_b, j_imd_b2);
        }
        return;
    }
    for (int g = 0; g < C.ng; ++g) {
        const float w = C.wgauge[g];
        if (!(w > 0.f || w < 0.f)) continue;
        const SxGaugeSums S = C.sums[g];
        const bool any = S.n > 0;
        float gauge_jobs = 0.f, j_imd = 0.f;
        for (int j = 0; j < C.njf; ++j) {
            if (any) {
                const float n = (float)S.n;
                switch (C.jobs_fun[j]) {
                    case 1: { const float mean_x = S.sum_x / n;
                              const float num = S.sum_xx - 2.f * S.sum_xy + S.sum_yy;
                              const float den = S.sum_xx - n * mean_x * mean_x;
                              j_imd = num / den; } break;
                    case 2: j_imd = sx_kge_value(sx_kge_components(S)); break;
                    case 3: { const float imd = sx_kge_value(sx_kge_components(S)); j_imd = imd * imd; } break;
                    case 4: j_imd = S.se; break;
                    case 5: j_imd = sqrtf(S.se / n); break;
                    case 6: j_imd = S.lg; break;
                    default: break;
                }
            }
            gauge_jobs = gauge_jobs + C.wjobs_fun[j] * j_imd;
        }
        if (w > 0.f) jobs = jobs + w * gauge_jobs;
        else if (phase == 1) C.medx[C.slot[g]] = gauge_jobs;
        else { arr[arr_size] = gauge_jobs; arr_b[arr_size] = 0.f; perm[arr_size] = arr_size; arr_gauge[arr_size] = g; ++arr_size; }
    }
    if (phase == 1) { C.out[2] = jobs; return; }
    float jobs_b = C.jobs_b;
    if (arr_size > 0) {
        // quantile(arr, 0.5) replaces the weighted sum (mwd_cost.f90:154, 675-723); QUANTILE_B forward_db.f90:4327-4370
        float res = arr[0];
        if (arr_size > 1) {
            sx_heap_sort_idx(arr_size, arr, perm);
            const float frac = (float)(arr_size - 1) * 0.5f + 1.f;
            if (frac <= 1.f) { res = arr[0]; arr_b[perm[0]] = arr_b[perm[0]] + jobs_b; }
            else if (frac >= (float)arr_size) { res = arr[arr_size - 1]; arr_b[perm[arr_size - 1]] = arr_b[perm[arr_size - 1]] + jobs_b; }
            else {
                const int k = (int)frac;
                const float q1 = arr[k - 1], q2 = arr[k];
                res = q1 + (q2 - q1) * (frac - (float)k);
                const float temp_b = (frac - (float)k) * jobs_b;
                arr_b[perm[k]] = arr_b[perm[k]] + temp_b;
                arr_b[perm[k - 1]] = arr_b[perm[k - 1]] + (jobs_b - temp_b);
            }
        } else {
            arr_b[0] = arr_b[0] + jobs_b;
        }
        jobs = res;
        jobs_b = 0.f;
    }
    C.out[0] = jobs;
    if (!adjoint) return;
    float j_imd_b = 0.f;   // carried across gauges exactly like the reference's scalar (forward_db.f90:2656-2704)
    for (int g = C.ng - 1; g >= 0; --g) {
        for (int j = 0; j < SX_MAXJF; ++j) { SxCostCoef z; z.kind = 0; z.c_xy = z.c_yy = z.c_y = z.c = 0.f; C.coef[g * SX_MAXJF + j] = z; }
        const float w = C.wgauge[g];
        if (!(w > 0.f || w < 0.f)) continue;
        const SxGaugeSums S = C.sums[g];
        const bool any = S.n > 0;
        const float gauge_jobs_b = (w > 0.f) ? w * jobs_b : arr_b[--arr_size];
        (void)S; (void)any;
        sx_cost_gauge_coef(C, g, gauge_jobs_b, j_imd_b);
    }
}

// parallel over (gauge, t): qsim_b(g,t)  (forward_db.f90:2709-2712)
__global__ void sx_k_cost_seeds(SxCostArgs C) {
    SX_LIBM_INIT();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (t >= C.nt) return;
    float out = 0.f;
    if (t >= C.s0 && (C.wgauge[g] > 0.f || C.wgauge[g] < 0.f)) {
        const float x = sx_qo(C, g, t), y = sx_qs(C, g, t);
        float y_b = 0.f;
        for (int j = C.njf - 1; j >= 0; --j) {
            const SxCostCoef c = C.coef[g * SX_MAXJF + j];
            if (c.kind == 1) { if (x >= 0.f) y_b = y_b + x * c.c_xy + 2.f * y * c.c_yy; }
            else if (c.kind == 4) { if (x >= 0.f) y_b = y_b + x * c.c_xy + 2.f * y * c.c_yy + c.c_y; }
            else if (c.kind == 2) { if (x >= 0.f) y_b = y_b - 2.f * (x - y) * c.c; }
            else if (c.kind == 3) {
                if (x > 0.f && y > 0.f) {
                    const float arg1 = y / x, arg2 = y / x;
                    const float arg1_b = sx_logf(arg2) * x * c.c / arg1;
                    const float arg2_b = sx_logf(arg1) * x * c.c / arg2;
                    y_b = y_b + arg2_b / x + arg1_b / x;
                }
            }
        }
        out = 0.f + C.dt * 1e3f * y_b / C.area[g];
    }
    C.qsim_b[(size_t)g * C.nt + t] = out;
}

// seeds per gauge CELL: q_b(cell) accumulates output_b%qsim(g,t) for g = ng..1 (forward_db.f90:8650-8654)
__global__ void sx_k_cost_cellseeds(SxCostArgs C) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int gc = blockIdx.y;
    if (t >= C.nt) return;
    float acc = 0.f;
    for (int g = C.ng - 1; g >= 0; --g)
        if (C.gauge_gid[g] == gc) acc = acc + C.qsim_b[(size_t)g * C.nt + t];
    C.qgb[(size_t)gc * C.nt + t] = acc;
}

// ------------------------------------------------------------------------------------------------
// tangent of the cost: COMPUTE_JOBS_D (forward_db.f90:2445-2551) with NSE_D / KGE_D / SE_D / RMSE_D / LOGARITHMIC_D
// and QUANTILE_D.  The criteria derivatives are differences of nearly equal fp32 sums, so they are evaluated with the
// reference's formulas and summation order (one thread per gauge, sequential in time); a mathematically equal
// evaluation through the adjoint seeds lands 5e-6 away.  qgd = q_d at the gauge cells; out[0] = jobs_d.
// ------------------------------------------------------------------------------------------------
__device__ inline void sx_heap_sort_pair(int n, float* arr, float* arr_d) {
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

__global__ void sx_k_cost_tangent(SxCostArgs C, const float* qgd, float* out) {
    SX_LIBM_INIT();
    float* gj = C.med; float* gjd = C.med + C.ng;     // gauge_jobs, gauge_jobs_d
    for (int g = threadIdx.x; g < C.ng; g += blockDim.x) {
        gj[g] = 0.f; gjd[g] = 0.f;
        const float w = C.wgauge[g];
        if (!(w > 0.f || w < 0.f)) continue;
        int n = 0;
        float sum_x = 0.f, sum_y = 0.f, sum_xx = 0.f, sum_yy = 0.f, sum_xy = 0.f, sum_y_d = 0.f, sum_yy_d = 0.f, sum_xy_d = 0.f;
        float se = 0.f, se_d = 0.f, lg = 0.f, lg_d = 0.f;
        bool any = false;
        const float* yd = qgd + (size_t)C.gauge_gid[g] * C.nt;
        for (int t = C.s0; t < C.nt; ++t) {
            const float x = sx_qo(C, g, t), y = sx_qs(C, g, t);
            const float y_d = C.dt * 1e3f * yd[t] / C.area[g];
            if (x >= 0.f) {
                any = true;
                n++;
                sum_x = sum_x + x; sum_y_d = sum_y_d + y_d; sum_y = sum_y + y;
                sum_xx = sum_xx + x * x;
                sum_yy_d = sum_yy_d + 2.f * y * y_d; sum_yy = sum_yy + y * y;
                sum_xy_d = sum_xy_d + x * y_d; sum_xy = sum_xy + x * y;
                se_d = se_d - 2.f * (x - y) * y_d; se = se + (x - y) * (x - y);
            }
            if (x > 0.f && y > 0.f) {
                const float a_d = y_d / x, a = y / x;
                const float lt = sx_logf(a);
                lg_d = lg_d + x * (lt * a_d / a + lt * a_d / a);
                lg = lg + x * (lt * lt);
            }
        }
        const float fn = (float)n;
        float gauge_jobs = 0.f, gauge_jobs_d = 0.f, j_imd = 0.f, j_imd_d = 0.f;
        for (int j = 0; j < C.njf; ++j) {
            if (any) {
                const int fun = C.jobs_fun[j];
                if (fun == 1) {
                    const float mean_x = sum_x / fn;
                    const float num_d = sum_yy_d - 2.f * sum_xy_d, num = sum_xx - 2.f * sum_xy + sum_yy;
                    const float den = sum_xx - fn * mean_x * mean_x;
                    j_imd_d = num_d / den; j_imd = num / den;
                } else if (fun == 2 || fun == 3) {
                    const float mean_x = sum_x / fn, mean_y_d = sum_y_d / fn, mean_y = sum_y / fn;
                    const float var_x = sum_xx / fn - mean_x * mean_x;
                    const float var_y_d = sum_yy_d / fn - 2.f * mean_y * mean_y_d, var_y = sum_yy / fn - mean_y * mean_y;
                    const float cov_d = sum_xy_d / fn - mean_x * mean_y_d, cov = sum_xy / fn - mean_x * mean_y;
                    const float sx = sqrtf(var_x), sy = sqrtf(var_y);
                    const float sy_d = (var_y == 0.f) ? 0.f : var_y_d / (2.0f * sy);
                    const float r = cov / (sx * sy);
                    const float r_d = (cov_d - r * sx * sy_d) / (sx * sy);
                    const float a_d = sy_d / sx, a = sy / sx;
                    const float b_d = mean_y_d / mean_x, b = mean_y / mean_x;
                    const float arg1_d = 2.f * (r - 1.f) * r_d + 2.f * (b - 1.f) * b_d + 2.f * (a - 1.f) * a_d;
                    const float arg1 = (r - 1.f) * (r - 1.f) + (b - 1.f) * (b - 1.f) + (a - 1.f) * (a - 1.f);
                    const float kv = sqrtf(arg1);
                    const float kd = (arg1 == 0.f) ? 0.f : arg1_d / (2.0f * kv);
                    if (fun == 2) { j_imd_d = kd; j_imd = kv; } else { j_imd_d = 2.f * kv * kd; j_imd = kv * kv; }
                } else if (fun == 4) { j_imd_d = se_d; j_imd = se; }
                else if (fun == 5) {
                    const float tv = sqrtf(se / fn);
                    j_imd_d = (se / fn == 0.f) ? 0.f : se_d / (2.0f * tv * fn);
                    j_imd = tv;
                } else if (fun == 6) { j_imd_d = lg_d; j_imd = lg; }
            }
            gauge_jobs_d = gauge_jobs_d + C.wjobs_fun[j] * j_imd_d;
            gauge_jobs = gauge_jobs + C.wjobs_fun[j] * j_imd;
        }
        gj[g] = gauge_jobs; gjd[g] = gauge_jobs_d;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float jobs_d = 0.f;
    int arr_size = 0;
    for (int g = 0; g < C.ng; ++g) {
        const float w = C.wgauge[g];
        if (w > 0.f) jobs_d = jobs_d + w * gjd[g];
        else if (w < 0.f) { gj[arr_size] = gj[g]; gjd[arr_size] = gjd[g]; ++arr_size; }   // in place: arr_size <= g
    }
    if (arr_size > 0) {   // QUANTILE_D, p = 0.5
        jobs_d = gjd[0];
        if (arr_size > 1) {
            sx_heap_sort_pair(arr_size, gj, gjd);
            const float frac = (float)(arr_size - 1) * 0.5f + 1.f;
            if (frac <= 1.f) jobs_d = gjd[0];
            else if (frac >= (float)arr_size) jobs_d = gjd[arr_size - 1];
            else { const int k = (int)frac; jobs_d = gjd[k - 1] + (frac - (float)k) * (gjd[k] - gjd[k - 1]); }
        }
    }
    out[0] = jobs_d;
}
