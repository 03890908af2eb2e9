// sx_jreg.h -- regularisation term of the cost and its adjoint on full (nrow, ncol) planes (gfx950).
//
//   compute_jreg   smash/solver/optimize/mwd_cost.f90:159-245   (COMPUTE_JREG_B forward_db.f90:2927-3092)
//   reg_prior      mwd_cost.f90:1180-1221                       (REG_PRIOR_B     forward_db.f90:5756-5799)
//   reg_smoothing  mwd_cost.f90:1100-1178                       (REG_SMOOTHING_B forward_db.f90:5504-5657)
//
// The reference accumulates every regulariser in ONE fp32 running sum over (field, column, row).  A sum of
// 10^6 fp32 terms carries ~1e-5 of its own rounding, so a tree reduction -- however accurate -- lands outside
// the 1e-6 parity bar; the sum is therefore taken in the reference's order: the terms are produced in parallel
// (sx_k_prior_terms / sx_k_smooth_terms), then ONE wavefront per running sum adds them one after the other
// (sx_k_seq_sum: 64 coalesced terms per load, lane broadcast + dependent v_add, ~8 cycles per term).  The chains
// are independent of the simulation and run on their own stream underneath the sweep.
// The adjoint is a gather per cell that applies the reference's scatter statements in the order its reverse
// sweep reaches them, so every plane is bit-identical to the Tapenade code's.
#pragma once

#include <hip/hip_runtime.h>

#define SX_JREG_MAXCHAIN 8      // njr (<= 4) x {parameters, states}
#define SX_JREG_MAXFIELD 16

struct SxJregChains {
    int nchain;
    int nplane[SX_JREG_MAXCHAIN];   // optimised fields of the chain; their term planes are consecutive
    long first[SX_JREG_MAXCHAIN];   // offset (in planes) of the chain's first term plane
};

__global__ void sx_k_prior_terms(float* t, const float* x, const float* xb, long n2) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= n2) return;
    const float d = x[c] - xb[c];
    t[c] = d * d;
}

// clamped neighbours of (row, col): the mask edge and inactive neighbours fall back on the cell itself
// (mwd_cost.f90:1139-1166)
__device__ __forceinline__ void sx_smooth_bounds(const int* active, int nrow, int ncol, int row, int col, int& mnc, int& mxc,
                                                 int& mnr, int& mxr) {
    mnc = max(col - 1, 0); mxc = min(col + 1, ncol - 1);
    mnr = max(row - 1, 0); mxr = min(row + 1, nrow - 1);
    if (active[row + (long)mnc * nrow] == 0) mnc = col;
    if (active[row + (long)mxc * nrow] == 0) mxc = col;
    if (active[mnr + (long)col * nrow] == 0) mnr = row;
    if (active[mxr + (long)col * nrow] == 0) mxr = row;
}
__device__ __forceinline__ float sx_smooth_mat(const float* x, const float* xb, int rel, long i) { return rel ? x[i] - xb[i] : x[i]; }

__global__ void sx_k_smooth_terms(float* t, const float* x, const float* xb, int rel, const int* active, int nrow, int ncol) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= (long)nrow * ncol) return;
    float v = 0.f;          // an inactive cell adds nothing: "+ 0" leaves the running sum unchanged
    if (active[c] == 1) {
        const int row = (int)(c % nrow), col = (int)(c / nrow);
        int mnc, mxc, mnr, mxr;
        sx_smooth_bounds(active, nrow, ncol, row, col, mnc, mxc, mnr, mxr);
        const float m0 = sx_smooth_mat(x, xb, rel, c);
        const float dr = sx_smooth_mat(x, xb, rel, mxr + (long)col * nrow) - 2.f * m0 + sx_smooth_mat(x, xb, rel, mnr + (long)col * nrow);
        const float dc = sx_smooth_mat(x, xb, rel, row + (long)mxc * nrow) - 2.f * m0 + sx_smooth_mat(x, xb, rel, row + (long)mnc * nrow);
        v = dr * dr + dc * dc;
    }
    t[c] = v;
}

// one wavefront per chain: strict left-to-right fp32 sum of nplane * n2 terms
__global__ __launch_bounds__(64) void sx_k_seq_sum(const float* terms, SxJregChains ch, long n2, float* out) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    const float* t = terms + ch.first[chain] * n2;
    const long n = (long)ch.nplane[chain] * n2;
    float acc = 0.f;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    auto load = [&](long base) {
        v0 = (base + lane < n) ? t[base + lane] : 0.f;
        v1 = (base + 64 + lane < n) ? t[base + 64 + lane] : 0.f;
        v2 = (base + 128 + lane < n) ? t[base + 128 + lane] : 0.f;
        v3 = (base + 192 + lane < n) ? t[base + 192 + lane] : 0.f;
    };
    load(0);
    for (long base = 0; base < n; base += 256) {
        const float a0 = v0, a1 = v1, a2 = v2, a3 = v3;
        if (base + 256 < n) load(base + 256);       // next batch in flight under the 256 dependent additions
#pragma unroll
        for (int l = 0; l < 64; ++l) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), l));
#pragma unroll
        for (int l = 0; l < 64; ++l) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), l));
#pragma unroll
        for (int l = 0; l < 64; ++l) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), l));
#pragma unroll
        for (int l = 0; l < 64; ++l) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a3), l));
    }
    if (lane == 0) out[chain] = acc;
}

// ---------------------------------------------------------------- adjoint
__global__ void sx_k_prior_b(float* g, const float* x, const float* xb, float res_b, long n2) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= n2) return;
    g[c] = g[c] + 2.f * (x[c] - xb[c]) * res_b;
}

// REG_SMOOTHING_B as a gather: the reverse sweep (col, row descending) visits the sources of cell X in the order
// (r, c+1), (r+1, c), X, (r-1, c), (r, c-1); each source applies its six statements in program order
__global__ void sx_k_smooth_b(float* g, const float* x, const float* xb, int rel, const int* active, int nrow, int ncol, float res_b) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= (long)nrow * ncol) return;
    float mb = 0.f;
    if (active[c] == 1) {
        const int row = (int)(c % nrow), col = (int)(c / nrow);
        const int sr[5] = {row, row + 1, row, row - 1, row};
        const int sc[5] = {col + 1, col, col, col, col - 1};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int r = sr[s], q = sc[s];
            if (r < 0 || r >= nrow || q < 0 || q >= ncol) continue;
            const long si = r + (long)q * nrow;
            if (active[si] != 1) continue;
            int mnc, mxc, mnr, mxr;
            sx_smooth_bounds(active, nrow, ncol, r, q, mnc, mxc, mnr, mxr);
            const float m0 = sx_smooth_mat(x, xb, rel, si);
            const float tb = 2.f * (sx_smooth_mat(x, xb, rel, mxr + (long)q * nrow) - 2.f * m0 + sx_smooth_mat(x, xb, rel, mnr + (long)q * nrow)) * res_b;
            const float tb0 = 2.f * (sx_smooth_mat(x, xb, rel, r + (long)mxc * nrow) - 2.f * m0 + sx_smooth_mat(x, xb, rel, r + (long)mnc * nrow)) * res_b;
            if (r == row && mxc == col) mb = mb + tb0;
            if (r == row && q == col) mb = mb - 2.f * tb0;
            if (r == row && mnc == col) mb = mb + tb0;
            if (mxr == row && q == col) mb = mb + tb;
            if (r == row && q == col) mb = mb - 2.f * tb;
            if (mnr == row && q == col) mb = mb + tb;
        }
    }
    g[c] = g[c] + mb;
}

// NORMALIZE_*_B (forward_db.f90:809-889, 1877-1900): gradient w.r.t. the denormalised value
__global__ void sx_k_plane_div(float* g, float d, long n2) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c < n2) g[c] = g[c] / d;
}
__global__ void sx_k_plane_scale(float* dst, const float* src, float scale, int scaled, long n2) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c < n2) dst[c] = scaled ? scale * src[c] : src[c];
}

// ---------------------------------------------------------------- tangent (REG_PRIOR_D, REG_SMOOTHING_D)
// terms of the running sums res_d of forward_db.f90:5720-5750 / 5382-5500; xd = tangent of the (normalised) field
__global__ void sx_k_prior_terms_d(float* t, const float* x, const float* xb, const float* xd, long n2) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= n2) return;
    t[c] = 2.f * (x[c] - xb[c]) * xd[c];
}
__global__ void sx_k_smooth_terms_d(float* t, const float* x, const float* xb, const float* xd, int rel, const int* active, int nrow, int ncol) {
    const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (c >= (long)nrow * ncol) return;
    float v = 0.f;
    if (active[c] == 1) {
        const int row = (int)(c % nrow), col = (int)(c / nrow);
        int mnc, mxc, mnr, mxr;
        sx_smooth_bounds(active, nrow, ncol, row, col, mnc, mxc, mnr, mxr);
        const float m0 = sx_smooth_mat(x, xb, rel, c);
        const float dr = sx_smooth_mat(x, xb, rel, mxr + (long)col * nrow) - 2.f * m0 + sx_smooth_mat(x, xb, rel, mnr + (long)col * nrow);
        const float dc = sx_smooth_mat(x, xb, rel, row + (long)mxc * nrow) - 2.f * m0 + sx_smooth_mat(x, xb, rel, row + (long)mnc * nrow);
        const float dr_d = xd[mxr + (long)col * nrow] - 2.f * xd[c] + xd[mnr + (long)col * nrow];
        const float dc_d = xd[row + (long)mxc * nrow] - 2.f * xd[c] + xd[row + (long)mnc * nrow];
        v = 2.f * dr * dr_d + 2.f * dc * dc_d;
    }
    t[c] = v;
}
