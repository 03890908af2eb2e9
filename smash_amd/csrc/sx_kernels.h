// sx_kernels.h -- HIP kernels of the smashx hot path (gfx950 / MI355X, wave64).
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   cells are renumbered k = 0..n-1 in routing-group order (sx_plan.h); npad = n rounded up to 256.
//   forcing      prcp/pet [nt][npad]     cell index fastest  -> lanes = consecutive cells, coalesced
//   params/state [npad] per field
//   qtT, hrT     [npad][Tc]              time fastest per cell: the time-skewed routing threads and the
//                                        marching vertical threads both read/write 16-B (4-step) pieces
//   tape_*       [Tc][npad]              pre-step reservoir levels of the current time chunk
//   exchange     [nx][Tc]                discharge (forward) / adjoint (reverse) series between groups
//   gauge series [ngauge][nt]
#pragma once

#include <hip/hip_runtime.h>

#include "sx_ops.h"

#define SX_BT 4           // time steps per routing super-step (one float4 per cell per super-step)
#define SX_VBLOCK 256     // threads (cells) per vertical workgroup
#define SX_VTILE 16       // steps per LDS transpose tile in the vertical kernels

struct SxDeviceArrays {
    // sizes
    int n, npad, nt, Tc;          // Tc = allocated chunk length (multiple of 16)
    float dt, dx;
    // forcing
    const float* prcp; const float* pet;
    // parameters (denormalised) and per-cell invariants
    float *ci, *cp, *cft, *cst, *exc, *lr;
    float *rt_a, *rt_f, *rt_denf, *rt_denb;   // exp(-dt/(60 lr)), real(flwacc-1), forward / adjoint denominators
    int* flwacc;
    // states (running) and adjoint state
    float *hi, *hp, *hft, *hst, *hlr;
    float *ci_b, *cp_b, *cft_b, *cst_b, *exc_b, *lr_b, *hi_b, *hp_b, *hft_b, *hst_b, *hlr_b;
    // chunk buffers
    float *qtT, *hrT;
    float *tape_hi, *tape_hp, *tape_hft, *tape_hst;
    float* xT;                    // exchange series
    // gauges
    float *qg, *qgb;              // [ngc][nt] discharge at gauge cells / adjoint seeds
    int* cell_gauge;              // [npad] gauge-cell id or -1
    // schedule
    const int *g_slot_begin, *g_dmax;
    const int *s_cell, *s_stage, *s_cstart, *s_ccount, *s_parent, *s_xout;
};

// ------------------------------------------------------------------------------------------------
// per-cell invariants of the routing operators (md_routing_operator.f90:55-56,75; LINEAR_ROUTING_B
// forward_db.f90:6643-6648; UPSTREAM_DISCHARGE_B :6551)
// ------------------------------------------------------------------------------------------------
__global__ void sx_k_prep_routing(SxDeviceArrays A) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n) return;
    const float lr = A.lr[k];
    const float f = (float)(A.flwacc[k] - 1);
    A.rt_a[k] = sx_expf(-A.dt / (lr * 60.f));
    A.rt_f[k] = f;
    A.rt_denf[k] = 0.001f * A.dx * A.dx * f;
    A.rt_denb[k] = 0.001f * (A.dx * A.dx) * f;
}

// ------------------------------------------------------------------------------------------------
// vertical forward: one thread per cell marches the time chunk [t0, t0+T)
// ------------------------------------------------------------------------------------------------
template <int ST, bool TAPE>
__global__ __launch_bounds__(SX_VBLOCK) void sx_k_vert_fwd(SxDeviceArrays A, int t0, int T) {
    __shared__ float tile[SX_VTILE][SX_VBLOCK + 1];
    const int kbase = blockIdx.x * SX_VBLOCK;
    const int k = kbase + threadIdx.x;
    const bool valid = k < A.n;
    const size_t npad = (size_t)A.npad;

    SxCellParams P;
    float hi = 0.f, hp = 0.f, hft = 0.f, hst = 0.f;
    if (valid) {
        P.ci = (ST == 2 || ST == 3) ? A.ci[k] : 1.f;
        P.cp = A.cp[k];
        P.inv_cp = 1.f / P.cp;
        P.cft = A.cft[k];
        P.cst = (ST == 3) ? A.cst[k] : 1.f;
        P.exc = (ST != 4) ? A.exc[k] : 0.f;
        P.cft_m4 = sx_pow_m4(P.cft);
        P.cst_m4 = (ST == 3) ? sx_pow_m4(P.cst) : 1.f;
        if (ST == 2 || ST == 3) hi = A.hi[k];
        hp = A.hp[k];
        hft = A.hft[k];
        if (ST == 3) hst = A.hst[k];
    }
    float prcp_n = 0.f, pet_n = 0.f;
    if (valid && T > 0) { prcp_n = A.prcp[(size_t)t0 * npad + k]; pet_n = A.pet[(size_t)t0 * npad + k]; }

    for (int tb = 0; tb < T; tb += SX_VTILE) {
        const int nstep = min(SX_VTILE, T - tb);
        for (int j = 0; j < nstep; ++j) {
            const int tt = tb + j;
            const float prcp = prcp_n, pet = pet_n;
            if (valid && tt + 1 < T) {
                prcp_n = A.prcp[(size_t)(t0 + tt + 1) * npad + k];
                pet_n = A.pet[(size_t)(t0 + tt + 1) * npad + k];
            }
            float qt = 0.f;
            if (valid) {
                if (TAPE) {
                    const size_t o = (size_t)tt * npad + k;
                    if (ST == 2 || ST == 3) A.tape_hi[o] = hi;
                    A.tape_hp[o] = hp;
                    A.tape_hft[o] = hft;
                    if (ST == 3) A.tape_hst[o] = hst;
                }
                qt = sx_vertical_step<ST>(P, prcp, pet, hi, hp, hft, hst);
            }
            tile[j][threadIdx.x] = qt;
        }
        __syncthreads();
        // transposed write-out: 4 lanes cover the 16 consecutive steps (64 B) of one cell
        for (int p = 0; p < (SX_VBLOCK * SX_VTILE / 4) / SX_VBLOCK; ++p) {
            const int idx = p * SX_VBLOCK + threadIdx.x;
            const int cl = idx >> 2, qd = idx & 3;
            const int kk = kbase + cl;
            if (kk < A.n && qd * 4 < nstep) {
                float4 v;
                v.x = tile[qd * 4 + 0][cl]; v.y = tile[qd * 4 + 1][cl];
                v.z = tile[qd * 4 + 2][cl]; v.w = tile[qd * 4 + 3][cl];
                *reinterpret_cast<float4*>(A.qtT + (size_t)kk * A.Tc + tb + qd * 4) = v;
            }
        }
        __syncthreads();
    }
    if (valid) {
        if (ST == 2 || ST == 3) A.hi[k] = hi;
        A.hp[k] = hp;
        A.hft[k] = hft;
        if (ST == 3) A.hst[k] = hst;
    }
}

// ------------------------------------------------------------------------------------------------
// routing forward: one workgroup per routing group, time-skewed wavefront through LDS.
// Slot j at stage s handles time block (w - s) in super-step w; its children (stage s-1) published that
// block in super-step w-1.  upstream_discharge + linear_routing + the q update of
// md_forward_structure.f90:150-156, in the reference's operation order.
// ------------------------------------------------------------------------------------------------
template <bool TAPE>
__global__ void sx_k_route_fwd(SxDeviceArrays A, int g0, int t0, int T) {
    extern __shared__ __attribute__((aligned(16))) float4 sx_lds[];   // [2][blockDim.x]
    const int g = g0 + blockIdx.x;
    const int sb = A.g_slot_begin[g], m = A.g_slot_begin[g + 1] - sb, dmax = A.g_dmax[g];
    const int j = threadIdx.x, M = blockDim.x;
    const bool valid = j < m;
    const int nb = (T + SX_BT - 1) / SX_BT;

    int cell = -1, stage = 0, cstart = 0, ccount = 0, xout = -1, xin = -1, gid = -1;
    float a = 0.f, f = 0.f, den = 1.f, hlr = 0.f;
    bool hasup = false;
    if (valid) {
        const int c = A.s_cell[sb + j];
        stage = A.s_stage[sb + j];
        if (c >= 0) {
            cell = c;
            cstart = A.s_cstart[sb + j]; ccount = A.s_ccount[sb + j]; xout = A.s_xout[sb + j];
            a = A.rt_a[c]; f = A.rt_f[c]; den = A.rt_denf[c]; hlr = A.hlr[c];
            hasup = A.flwacc[c] > 1;
            gid = A.cell_gauge[c];
        } else {
            xin = -1 - c;
        }
    }
    const float dt = A.dt, dx = A.dx;
    const size_t Tc = (size_t)A.Tc;
    const float* src = (cell >= 0) ? (A.qtT + (size_t)cell * Tc) : (xin >= 0 ? A.xT + (size_t)xin * Tc : nullptr);
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && stage == 0 && nb > 0) nxt = *reinterpret_cast<const float4*>(src);   // first block of stage-0 slots

    const int nsuper = nb + dmax;
    for (int w = 0; w < nsuper; ++w) {
        const int tb = w - stage;
        const bool act = valid && tb >= 0 && tb < nb;
        const float4 cur4 = nxt;
        // prefetch the block this slot handles in the next super-step
        if (valid && tb + 1 >= 0 && tb + 1 < nb) nxt = *reinterpret_cast<const float4*>(src + (size_t)(tb + 1) * SX_BT);
        float4* pub = sx_lds + (size_t)(w & 1) * M;
        const float4* prev = sx_lds + (size_t)((w + 1) & 1) * M;
        if (act) {
            const int tl = tb * SX_BT;           // first step of the block, chunk-local
            if (cell >= 0) {
                float s[SX_BT] = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < ccount; ++c) {
                    const float4 v = prev[cstart + c];
                    s[0] = s[0] + v.x; s[1] = s[1] + v.y; s[2] = s[2] + v.z; s[3] = s[3] + v.w;
                }
                const float qt[SX_BT] = {cur4.x, cur4.y, cur4.z, cur4.w};
                float q[SX_BT], hr[SX_BT];
#pragma unroll
                for (int i = 0; i < SX_BT; ++i) {
                    q[i] = 0.f; hr[i] = 0.f;
                    if (tl + i < T) {
                        float qup = 0.f;
                        if (hasup) qup = (s[i] * dt) / den;
                        const float hr_imd = hlr + qup;
                        hlr = hr_imd * a;
                        const float qrout = hr_imd - hlr;
                        q[i] = (qt[i] + qrout * f) * dx * dx * 0.001f / dt;
                        hr[i] = hr_imd;
                    }
                }
                const float4 q4 = make_float4(q[0], q[1], q[2], q[3]);
                pub[j] = q4;
                if (xout >= 0) *reinterpret_cast<float4*>(A.xT + (size_t)xout * Tc + tl) = q4;
                if (TAPE) *reinterpret_cast<float4*>(A.hrT + (size_t)cell * Tc + tl) = make_float4(hr[0], hr[1], hr[2], hr[3]);
                if (gid >= 0) {
#pragma unroll
                    for (int i = 0; i < SX_BT; ++i)
                        if (tl + i < T) A.qg[(size_t)gid * A.nt + t0 + tl + i] = q[i];
                }
            } else {
                pub[j] = cur4;
            }
        }
        __syncthreads();
    }
    if (valid && cell >= 0) A.hlr[cell] = hlr;
}

// ------------------------------------------------------------------------------------------------
// routing adjoint: same groups, roots first, time descending.  Reverse of the q update, LINEAR_ROUTING_B
// (forward_db.f90:6628-6652) and UPSTREAM_DISCHARGE_B (:6520-6564), as in GR_x_FORWARD_B :8649-8672.
// Reads hrT (hr_imd of the recomputed forward) and the gauge seeds; writes qt_b over qtT.
// ------------------------------------------------------------------------------------------------
__global__ void sx_k_route_adj(SxDeviceArrays A, int g0, int t0, int T) {
    extern __shared__ __attribute__((aligned(16))) float4 sx_lds[];
    const int g = g0 + blockIdx.x;
    const int sb = A.g_slot_begin[g], m = A.g_slot_begin[g + 1] - sb, dmax = A.g_dmax[g];
    const int j = threadIdx.x, M = blockDim.x;
    const bool valid = j < m;
    const int nb = (T + SX_BT - 1) / SX_BT;

    int cell = -1, rstage = 0, par = -1, xout = -1, xin = -1, gid = -1;
    float a = 0.f, f = 0.f, den = 1.f, lr = 1.f, hr_b = 0.f, lr_b = 0.f;
    bool hasup = false;
    if (valid) {
        const int c = A.s_cell[sb + j];
        rstage = dmax - A.s_stage[sb + j];
        par = A.s_parent[sb + j];
        if (c >= 0) {
            cell = c;
            xout = A.s_xout[sb + j];
            a = A.rt_a[c]; f = A.rt_f[c]; den = A.rt_denb[c]; lr = A.lr[c];
            hr_b = A.hlr_b[c]; lr_b = A.lr_b[c];
            hasup = A.flwacc[c] > 1;
            gid = A.cell_gauge[c];
        } else {
            xin = -1 - c;
        }
    }
    const float dt = A.dt, dx = A.dx;
    const size_t Tc = (size_t)A.Tc;
    const int nsuper = nb + dmax;
    for (int w = 0; w < nsuper; ++w) {
        const int tbr = w - rstage;
        const bool act = valid && tbr >= 0 && tbr < nb;
        float4* pub = sx_lds + (size_t)(w & 1) * M;
        const float4* prev = sx_lds + (size_t)((w + 1) & 1) * M;
        if (act) {
            const int tb = nb - 1 - tbr;
            const int tl = tb * SX_BT;
            float4 in4 = make_float4(0.f, 0.f, 0.f, 0.f);   // contribution of the downstream cell
            if (par >= 0) in4 = prev[par];
            else if (cell >= 0 && xout >= 0) in4 = *reinterpret_cast<const float4*>(A.xT + (size_t)xout * Tc + tl);
            if (cell >= 0) {
                const float4 hr4 = *reinterpret_cast<const float4*>(A.hrT + (size_t)cell * Tc + tl);
                const float hrv[SX_BT] = {hr4.x, hr4.y, hr4.z, hr4.w};
                const float inv[SX_BT] = {in4.x, in4.y, in4.z, in4.w};
                float pb[SX_BT], qtb[SX_BT];
#pragma unroll
                for (int i = SX_BT - 1; i >= 0; --i) {
                    pb[i] = 0.f; qtb[i] = 0.f;
                    if (tl + i < T) {
                        float q_b = 0.f;
                        if (gid >= 0) q_b = q_b + A.qgb[(size_t)gid * A.nt + t0 + tl + i];
                        q_b = q_b + inv[i];
                        const float temp_b = (dx * dx) * 0.001f * q_b / dt;
                        const float qrout_b = f * temp_b;
                        hr_b = hr_b - qrout_b;
                        const float hr_imd_b = qrout_b + a * hr_b;
                        const float arg1_b = a * hrv[i] * hr_b;
                        lr_b = lr_b + dt * arg1_b / ((lr * lr) * 60.f);
                        hr_b = hr_imd_b;
                        if (hasup) pb[i] = dt * hr_imd_b / den;
                        qtb[i] = temp_b;
                    }
                }
                pub[j] = make_float4(pb[0], pb[1], pb[2], pb[3]);
                *reinterpret_cast<float4*>(A.qtT + (size_t)cell * Tc + tl) = make_float4(qtb[0], qtb[1], qtb[2], qtb[3]);
            } else {
                // inlet pseudo-cell: hand the receiver's contribution to the subtree rooted upstream
                *reinterpret_cast<float4*>(A.xT + (size_t)xin * Tc + tl) = in4;
            }
        }
        __syncthreads();
    }
    if (valid && cell >= 0) { A.hlr_b[cell] = hr_b; A.lr_b[cell] = lr_b; }
}

// ------------------------------------------------------------------------------------------------
// vertical adjoint: one thread per cell marches the chunk backwards, consuming qt_b (in qtT) and the
// taped pre-step levels; parameter gradients accumulate per cell in reverse time order like
// parameters_b%x(row,col) does in the reference (forward_db.f90:8699-8702).
// ------------------------------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(SX_VBLOCK) void sx_k_vert_adj(SxDeviceArrays A, int t0, int T) {
    const int k = blockIdx.x * SX_VBLOCK + threadIdx.x;
    if (k >= A.n) return;
    const size_t npad = (size_t)A.npad;
    SxCellParams P;
    P.ci = (ST == 2 || ST == 3) ? A.ci[k] : 1.f;
    P.cp = A.cp[k];
    P.inv_cp = 1.f / P.cp;
    P.cft = A.cft[k];
    P.cst = (ST == 3) ? A.cst[k] : 1.f;
    P.exc = (ST != 4) ? A.exc[k] : 0.f;
    float cft_m5, cst_m5 = 1.f;
    sx_pow_m4_m5(P.cft, &P.cft_m4, &cft_m5);
    P.cst_m4 = 1.f;
    if (ST == 3) sx_pow_m4_m5(P.cst, &P.cst_m4, &cst_m5);
    SxCellGrads G;
    G.ci_b = (ST == 2 || ST == 3) ? A.ci_b[k] : 0.f;
    G.cp_b = A.cp_b[k];
    G.cft_b = A.cft_b[k];
    G.cst_b = (ST == 3) ? A.cst_b[k] : 0.f;
    G.exc_b = (ST != 4) ? A.exc_b[k] : 0.f;
    G.hi_b = (ST == 2 || ST == 3) ? A.hi_b[k] : 0.f;
    G.hp_b = A.hp_b[k];
    G.hft_b = A.hft_b[k];
    G.hst_b = (ST == 3) ? A.hst_b[k] : 0.f;
    const float* qtb = A.qtT + (size_t)k * A.Tc;
    for (int tq = (T - 1) / 4; tq >= 0; --tq) {
        const float4 q4 = *reinterpret_cast<const float4*>(qtb + tq * 4);
        const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            const int tt = tq * 4 + i;
            if (tt < T) {
                const size_t o = (size_t)tt * npad + k;
                const float prcp = A.prcp[(size_t)(t0 + tt) * npad + k], pet = A.pet[(size_t)(t0 + tt) * npad + k];
                const float hi = (ST == 2 || ST == 3) ? A.tape_hi[o] : 0.f;
                const float hp = A.tape_hp[o], hft = A.tape_hft[o];
                const float hst = (ST == 3) ? A.tape_hst[o] : 0.f;
                sx_vertical_step_b<ST>(P, cft_m5, cst_m5, prcp, pet, hi, hp, hft, hst, qv[i], G);
            }
        }
    }
    if (ST == 2 || ST == 3) { A.ci_b[k] = G.ci_b; A.hi_b[k] = G.hi_b; }
    A.cp_b[k] = G.cp_b;
    A.cft_b[k] = G.cft_b;
    if (ST == 3) { A.cst_b[k] = G.cst_b; A.hst_b[k] = G.hst_b; }
    if (ST != 4) A.exc_b[k] = G.exc_b;
    A.hp_b[k] = G.hp_b;
    A.hft_b[k] = G.hft_b;
}
