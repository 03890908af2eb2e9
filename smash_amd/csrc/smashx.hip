// smashx.hip -- C ABI implementation (include/smashx.h): plan, HBM residency, sweep orchestration.
// gfx950 only.  No CPU compute path: every entry point that needs the device fails without one.
#include "../../include/smashx.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the functions are resolved at run time (rccl() below)

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sx_cost.h"
#include "sx_jreg.h"
#include "sx_kernels.h"
#include "sx_plan.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(SMASHX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

// Which of the 16 / 8 fields each structure reads (smash/core/_constant.py:15-29; ci via gr_interception), by device
// slot.  Parameter slots: 0 ci, 1 cp, 2 cft, 3 cst, 4 exc, 5 lr, 6..8 px[0..2]; vic-a puts b, cusl1, cusl2, clsl, ks in
// slots 0..4 and ds, dsm, ws in 6..8.  State slots: 0 hi, 1 hp, 2 hft, 3 hst, 4 hlr; vic-a: husl1, husl2, hlsl in 0..2.
const int NPS = 9, NSS = 5;
int param_field(int st, int slot) {
    if (st == 5) {
        const int f[NPS] = {SMASHX_P_B, SMASHX_P_CUSL1, SMASHX_P_CUSL2, SMASHX_P_CLSL, SMASHX_P_KS, SMASHX_P_LR,
                            SMASHX_P_DS, SMASHX_P_DSM, SMASHX_P_WS};
        return f[slot];
    }
    switch (slot) {
        case 0: return (st == 2 || st == 3) ? SMASHX_P_CI : -1;
        case 1: return SMASHX_P_CP;
        case 2: return SMASHX_P_CFT;
        case 3: return st == 3 ? SMASHX_P_CST : -1;
        case 4: return st != 4 ? SMASHX_P_EXC : -1;
        case 5: return SMASHX_P_LR;
        default: return -1;
    }
}
int state_field(int st, int slot) {
    if (st == 5) { const int f[NSS] = {SMASHX_S_HUSL1, SMASHX_S_HUSL2, SMASHX_S_HLSL, -1, SMASHX_S_HLR}; return f[slot]; }
    switch (slot) {
        case 0: return (st == 2 || st == 3) ? SMASHX_S_HI : -1;
        case 1: return SMASHX_S_HP;
        case 2: return SMASHX_S_HFT;
        case 3: return st == 3 ? SMASHX_S_HST : -1;
        default: return SMASHX_S_HLR;
    }
}
int param_slot_of(int st, int f) { for (int i = 0; i < NPS; ++i) if (param_field(st, i) == f) return i; return -1; }
int state_slot_of(int st, int f) { for (int i = 0; i < NSS; ++i) if (state_field(st, i) == f) return i; return -1; }

// ---- small elementwise kernels on full (nrow*ncol) fields and on the cell vectors -------------------
__global__ void k_denormalize(float* a, long n, float lb, float ub) {   // mwd_parameters_manipulation.f90:199
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * (ub - lb) + lb;
}
__global__ void k_normalize(float* a, long n, float lb, float ub) {     // mwd_parameters_manipulation.f90:172
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (a[i] - lb) / (ub - lb);
}
__global__ void k_gather(float* dst, const float* src, const int* idx, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[idx[k]];
}
__global__ void k_gather_rows(float* dst, const float* src, const int* idx, int n, int npad, long plane, int rows) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (k < n && r < rows) dst[(size_t)r * npad + k] = src[(size_t)r * plane + idx[k]];
}
__global__ void k_scatter(float* dst, const float* src, const int* idx, int n, float scale, int scaled) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[idx[k]] = scaled ? scale * src[k] : src[k];
}

// ---- compact forcing: encode + verify (smashx_set_forcing_layout) ---------------------------------------------
// src: fp32 rows, row r = time step t0 + r, cell k at src[r * ld + (idx ? idx[k] : k)].  A value is accepted only if the kernels'
// decode reproduces its bits; status[0] counts rejects, status[1] holds the bits of the one gap value (0 = none seen yet).
__global__ void k_encode_prcp(unsigned short* dst, const float* src, const int* idx, int n, int npad, long ld, int rows, float c,
                              unsigned* status) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (k >= n || r >= rows) return;
    const float v = src[(size_t)r * ld + (idx ? idx[k] : k)];
    unsigned code = 65535u;
    if (v < 0.f) {
        const unsigned bits = __float_as_uint(v);
        const unsigned old = atomicCAS(status + 1, 0u, bits);
        if (old != 0u && old != bits) atomicAdd(status, 1u);             // a second kind of negative value: not representable
    } else {
        const float kf = rintf(v / c);
        code = (kf >= 0.f && kf < 65535.f) ? (unsigned)kf : 65535u;
        if (code == 65535u || __float_as_uint((float)code * c) != __float_as_uint(v)) atomicAdd(status, 1u);
    }
    dst[(size_t)r * npad + k] = (unsigned short)code;
}
// one thread per (cell, day touched by the block): finds the daily value D with D * ratio(h) == pet bit for bit for every hour
// of the day inside the block (or D < 0 == every hour: a gap day); a day already set by an earlier block is only verified.
// petd holds NaN while a day is undetermined (only night hours seen so far).
__global__ void k_encode_pet(float* petd, const float* src, const int* idx, int n, int npad, long ld, int t0, int rows, int hour0,
                             const float* ratio, unsigned* status) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int d0 = (t0 + hour0) / 24, d = d0 + blockIdx.y;
    if (k >= n) return;
    const int ta = max(t0, d * 24 - hour0), tb = min(t0 + rows, (d + 1) * 24 - hour0);     // steps of day d in this block
    if (ta >= tb) return;
    const size_t col = idx ? idx[k] : k;
    float* slot = petd + (size_t)d * npad + k;
    float D = *slot;
    // two "not determined yet" marks: nothing seen (all ones) / only night hours seen, all of them +0 (then the day cannot be a gap day)
    const unsigned PENDING_ZEROS = 0x7fc00001u;
    const bool zeros_seen = __float_as_uint(D) == PENDING_ZEROS;
    auto fits = [&](float Dc) {
        for (int t = ta; t < tb; ++t) {
            const float v = src[(size_t)(t - t0) * ld + col];
            const float w = Dc < 0.f ? Dc : Dc * ratio[(t + hour0) % 24];
            if (__float_as_uint(w) != __float_as_uint(v)) return false;
        }
        return true;
    };
    if (D == D) { if (!fits(D)) atomicAdd(status, 1u); return; }
    // the hour with the largest share pins D to within two ulps; a negative value must be the whole day's
    int tbest = -1; float rbest = 0.f;
    bool allzero = true;
    for (int t = ta; t < tb; ++t) {
        const float v = src[(size_t)(t - t0) * ld + col], r = ratio[(t + hour0) % 24];
        if (__float_as_uint(v) != 0u) allzero = false;
        if (v < 0.f) { if (!zeros_seen && fits(v)) *slot = v; else atomicAdd(status, 1u); return; }   // a gap day is -99 at EVERY hour
        if (r > rbest) { rbest = r; tbest = t; }
    }
    if (tbest < 0) {                                                          // night hours only: any D >= 0 fits, stays open
        if (!allzero) atomicAdd(status, 1u); else *slot = __uint_as_float(PENDING_ZEROS);
        return;
    }
    const float v = src[(size_t)(tbest - t0) * ld + col];
    const unsigned b0 = __float_as_uint(v / rbest);
    for (int j = 0; j <= 6; ++j) {
        const int off = (j & 1) ? (j + 1) / 2 : -(j / 2);                       // 0, +1, -1, +2, -2, +3, -3 ulps
        const float Dc = __uint_as_float(b0 + (unsigned)off);
        if (Dc >= 0.f && Dc == Dc && fits(Dc)) { *slot = Dc; return; }
    }
    atomicAdd(status, 1u);
}
__global__ void k_close_petd(float* petd, size_t n) {      // days that only ever showed night hours: any value works, take 0
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && !(petd[i] == petd[i])) petd[i] = 0.f;
}

// ---- control vector of the calibration (mw_optimize.f90:679-777) ------------------------------------------------------------
// control_to_var: x[off + pos[k]] (fp64, optimiser space) -> cell vector (denormalised like k_denormalize), the full plane the
// downloads read, and -- when a regulariser is on -- the plane compute_jreg sees (normalise(denormalise(v)), mwd_cost.f90:284-291)
__global__ void k_control_set(float* cellv, float* full, float* jx, const double* x, const int* pos, const int* flat, int n, long off,
                              float lb, float ub, int denorm) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float v = (float)x[off + pos[k]];
    const float d = denorm ? v * (ub - lb) + lb : v;
    if (cellv) { cellv[k] = d; full[flat[k]] = d; }       // null: a flagged field the structure does not read (only the regulariser sees it)
    if (jx) jx[flat[k]] = denorm ? (d - lb) / (ub - lb) : v;
}
// var_to_control (grad = 0): the optimiser-space value of the field; gradient (grad = 1): cell gradient x (ub - lb) under
// denormalize_forward (DENORMALIZE_*_B, forward_db.f90:967-1057)
__global__ void k_control_get(double* x, const float* src, const int* pos, const int* flat, int n, long off, float lb, float ub,
                              int denorm, int grad, int from_full) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float v = src[from_full ? flat[k] : k];
    float r;
    if (grad) r = denorm ? (ub - lb) * v : v;
    else r = denorm ? (v - lb) / (ub - lb) : v;
    x[off + pos[k]] = (double)r;
}

// boundary series <-> dense message buffer [edge][Tq] float4
__global__ void k_halo_pack(float4* buf, const float4* x4, const int* slots, int nedge, int nx, int Tq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nedge * Tq) return;
    const int e = i / Tq, tb = i % Tq;
    buf[(size_t)e * Tq + tb] = x4[(size_t)tb * nx + slots[e]];
}
__global__ void k_halo_unpack(float4* x4, const float4* buf, const int* slots, int nedge, int nx, int Tq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nedge * Tq) return;
    const int e = i / Tq, tb = i % Tq;
    x4[(size_t)tb * nx + slots[e]] = buf[(size_t)e * Tq + tb];
}

// self-test of sx_fdiv against the IEEE division on pseudo-random operands: a = +-2^[-20,10) x [1,2), b in [blo, bhi)
__global__ void k_selftest_div(unsigned long long n, unsigned seed, float blo, float bhi, unsigned long long* out) {
    unsigned long long bad = 0, bad2 = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
        const unsigned m1 = (unsigned)h & 0x7fffffu, m2 = (unsigned)(h >> 23) & 0x7fffffu;
        const int e1 = (int)((h >> 46) % 30) - 20;
        const float a0 = ldexpf(__uint_as_float(0x3f800000u | m1), e1);
        const float a = ((h >> 63) & 1) ? -a0 : a0;
        const float b = blo + (bhi - blo) * ((float)m2 * (1.0f / 8388608.0f));
        const float q = sx_fdiv(a, b), ref = a / b;
        if (q != ref) { ++bad; if (fabsf(q - ref) > fabsf(ref) * 2.4e-7f) ++bad2; }
    }
    if (bad) atomicAdd(out, bad);
    if (bad2) atomicAdd(out + 1, bad2);
}

// self-test of the per-wavefront straight-line paths of sx_math.h against the branchy forms they shortcut: out[0] = tanh results that
// differ (arguments a wavefront at a time: all small -> the fast path runs; every 7th wavefront mixed with large ones -> it must
// not), out[1] = x^y results that differ (positive normal bases, ordinary exponents; every 7th wavefront mixed with 0, 1, inf, < 0)
__global__ void k_selftest_paths(unsigned long long n, unsigned seed, unsigned long long* out) {
    unsigned long long bad_t = 0, bad_p = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
        const bool mixed = ((i >> 6) % 7ull) == 6ull;                    // wave-uniform
        const unsigned m1 = (unsigned)h & 0x7fffffu, m2 = (unsigned)(h >> 23) & 0x7fffffu;
        // tanh argument: 2^[-70, -3) x [1, 2), either sign (|x| < ln2 / 4 = 0.173); mixed wavefronts: up to 2^5
        const int e1 = (int)((h >> 46) % (mixed ? 75u : 67u)) - 70;
        float x = ldexpf(__uint_as_float(0x3f800000u | m1), e1);
        if ((h >> 63) & 1) x = -x;
        if ((h >> 60) % 97ull == 0) x = 0.f;
        if (__float_as_uint(sx_tanhf(x, true)) != __float_as_uint(sx_tanhf(x, false))) ++bad_t;
        // power: base 2^[-30, 30) x [1, 2), exponent in (-6, 6)
        const int e2 = (int)((h >> 52) % 60u) - 30;
        float b = ldexpf(__uint_as_float(0x3f800000u | m2), e2);
        float y = ((float)(m1 >> 3) * (1.0f / 1048576.0f) - 0.5f) * 12.0f;
        if (mixed) { const unsigned k = (unsigned)(h >> 40) & 7u; if (k == 0) b = 0.f; else if (k == 1) b = 1.f; else if (k == 2) b = sx_inff(); else if (k == 3) b = -b; else if (k == 4) y = 0.f; else if (k == 5) b = 1e-41f; }
#if !SX_EXACT_LIBM      // (the exact-libm build raises powers through glibc's own algorithm: no such path)
        const float pf = sx_pow_from(sx_log2_d(b, true), b, y, true), pg = sx_pow_from(sx_log2_d(b, false), b, y, false);
        if (__float_as_uint(pf) != __float_as_uint(pg) && !(pf != pf && pg != pg)) ++bad_p;
#else
        (void)b; (void)y;
#endif
    }
    if (bad_t) atomicAdd(out, bad_t);
    if (bad_p) atomicAdd(out + 1, bad_p);
}

// whole-domain outputs: T4 chunk buffer -> planes of `plane` floats per time step, cell k at idx[k]
__global__ void k_domain_export(float* stage, const float* src4, const int* idx, int n, int npad, long plane, int tl0, int nb) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int tt = blockIdx.y;
    if (k >= n || tt >= nb) return;
    const int tl = tl0 + tt;
    stage[(size_t)tt * plane + idx[k]] = src4[((size_t)(tl >> 2) * npad + k) * 4 + (tl & 3)];
}
__global__ void k_fill(float* dst, float v, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

// qsim_d(g, t) from the per-gauge-cell tangent series
__global__ void k_gauge_rows(float* dst, const float* src, const int* gid, int ng, int nt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (t < nt && g < ng) dst[(size_t)g * nt + t] = src[(size_t)gid[g] * nt + t];
}


// ---- RCCL, resolved at run time ------------------------------------------------------------------------------
// The library is not linked: a single-GPU host never loads it, and a host process that already holds an RCCL (PyTorch
// ships its own librccl.so.1) must not get a second copy -- dlopen by soname returns the copy already mapped.
struct RcclApi {
    void* h = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;        // optional (diagnostics: smashx_comm_info)
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok() const { return h != nullptr && err.empty(); }
};
RcclApi& rccl() {
    static RcclApi R;
    static bool tried = false;
    if (tried) return R;
    tried = true;
    const char* names[] = {getenv("SMASHX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !n[0]) continue;
        R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (R.h) break;
    }
    if (!R.h) { R.err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : ""); return R; }
#define SX_SYM(f) do { R.f = (decltype(R.f))dlsym(R.h, "nccl" #f); if (!R.f) R.err = "librccl lacks nccl" #f; } while (0)
    SX_SYM(GetUniqueId); SX_SYM(CommInitRank); SX_SYM(CommDestroy); SX_SYM(Send); SX_SYM(Recv); SX_SYM(AllReduce);
    SX_SYM(GroupStart); SX_SYM(GroupEnd); SX_SYM(GetErrorString);
#undef SX_SYM
    R.CommCount = (decltype(R.CommCount))dlsym(R.h, "ncclCommCount");
    R.GetVersion = (decltype(R.GetVersion))dlsym(R.h, "ncclGetVersion");
    return R;
}
#define NCCLCHK(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess)                                                                          \
            return fail(SMASHX_E_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

struct SxComm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = -1;
    hipStream_t stream = nullptr;    // collectives outside the sweeps (agreement check, cost sum)
    double* d_buf = nullptr;         // 64 doubles
    std::vector<smashx_plan*> plans; // plans whose exchange runs on this communicator: unhooked when it is destroyed
};
struct PeerSeg { int rank, first, count; };   // a run of boundary edges that share the peer rank

struct Launch { hipEvent_t a, b; int kind; double cellsteps; };

}  // namespace

struct smashx_plan {
    smashx_config cfg{};
    SxSchedule sch;
    int n = 0, npad = 0, nt = 0, ng = 0, st = 0;
    long n2 = 0;
    int M = 0;
    int Tc = 0, nchunks = 0;
    bool have_forcing = false, have_options = false, have_qobs = false, uploaded = false;
    bool chunk_ready = false, adj_ready = false;
    smashx_options opt{};
    std::vector<float> wgauge;
    hipStream_t stream = nullptr;    // vertical kernels, uploads/downloads ("V stream")
    hipStream_t stream_r = nullptr;  // routing + cost kernels ("R stream"); overlaps the V stream chunk by chunk
    int Tp = 0;                      // pipeline sub-chunk length inside a storage chunk
    bool hi_tape = true;             // gr-b / gr-c: full tape of the interception level (false: sparse checkpoints, rebuilt in the reverse kernel)
    int chain_from = 1;              // first chained round
    // staging rows of the chained groups (sx_kernels.h "Staging rows"): tables of the transposition kernels, rows beyond the time blocks,
    // LDS of a wave-block's FIFOs; SMASHX_CHAIN_STAGE=0 keeps the plain rows (A/B)
    bool chain_stage = false; SxStageTables stg{}; int stg_blocks = 0, stg_rows_extra = 0; size_t stg_lds = 0;
    size_t vlds_fwd = 0, vlds_adj = 0;   // experiments only (SMASHX_DEBUG_VLDS = n or nfwd,nadj): bytes of unused dynamic LDS per vertical workgroup, which
                                     // caps the vertical workgroups resident on a compute unit (occupancy experiments)
    std::vector<hipEvent_t> buf_free;    // per pipeline sub-chunk: the R stream has finished with this part of the chunk buffers
    std::vector<double> round_ncells;  // cells per routing round
    bool chain_used = false;         // a chained launch ran in the current sweep: check the stall flag afterwards
    bool chain = true;               // all routing rounds in one launch (progress counters), see sx_kernels.h
    // tile boundary exchange
    int n_out = 0, n_in = 0;
    int *d_out_x = nullptr, *d_in_x = nullptr;
    float *halo_out = nullptr, *halo_in = nullptr;   // caller-owned device buffers
    smashx_halo_fn halo_fn = nullptr; void* halo_user = nullptr;
    // native exchange (smashx_set_exchange): edges regrouped by peer rank, plan-owned message buffers
    SxComm* xcomm = nullptr;
    std::vector<PeerSeg> out_segs, in_segs;
    int *d_out_xp = nullptr, *d_in_xp = nullptr;     // exchange slots of the out / in edges in peer-grouped order
    float *x_out = nullptr, *x_in = nullptr;         // [edge][Tp/4] float4
    float* x_keep = nullptr; size_t x_keep_cap = 0;  // received inlet series of the storage chunks the reverse sweep recomputes: [chunk][sub-chunk][edge][Tp/4] float4
    std::vector<void*> allocs; std::vector<size_t> alloc_bytes;
    double bytes = 0;
    double hbm_free_at_plan = -1.0, hbm_total = -1.0;    // hipMemGetInfo when the storage-chunk length was chosen (smashx_plan_hbm)
    SxDeviceArrays A{};
    // extra device storage
    int* d_cell_flat = nullptr;      // k -> flat (row + col*nrow)
    int* d_sparse_idx = nullptr;     // k -> index in the sparse (nac) vectors
    float last_jobs_d = 0.f, last_jreg_d = 0.f;      // the two terms of the last smashx_forward_d's cost_d (smashx_tangent_terms)
    long n_sparse = 0;               // length of a sparse vector = the WHOLE grid's active cells along path (= n on an untiled plan)
    float* d_stage = nullptr;        // staging for full planes
    long stage_planes = 0;
    float* d_fullP[SMASHX_GNP] = {nullptr};
    float* d_fullS[SMASHX_GNS] = {nullptr};
    float* st0[5] = {nullptr};       // initial (denormalised) states in cell order
    float* ckpt = nullptr;           // [nchunks][5][npad]
    float* d_prcp = nullptr; float* d_pet = nullptr;
    // control vector (smashx_control_*): position of cell k among the active cells in column-major order, device staging
    int* d_ctrl_pos = nullptr; double* d_ctrl = nullptr; size_t ctrl_cap = 0;
    // compact forcing (smashx_set_forcing_layout)
    smashx_forcing_layout flay{};
    unsigned short* d_prcp16 = nullptr; float* d_petd = nullptr; float* d_ratio = nullptr; unsigned* d_fstatus = nullptr;
    int ndays = 0; bool petd_open = false;
    std::vector<char> block_seen;    // device-block uploads: steps covered so far (the forcing counts as set once every step is)
    // cost
    int ngc = 0;
    std::vector<int> gauge_gid;
    int *d_gauge_gid = nullptr, *d_gauge_flwacc = nullptr;
    float *d_area = nullptr, *d_wgauge = nullptr, *d_qobs = nullptr, *d_qsim_b = nullptr, *d_cost_out = nullptr;
    SxGaugeSums* d_sums = nullptr; SxCostCoef* d_coef = nullptr;
    float* d_med = nullptr; int* d_med_idx = nullptr;    // median over negative-weight gauges
    int med_nslots = 0; int* d_med_slot = nullptr; float* d_medx = nullptr; int* d_medx_idx = nullptr;   // ... of several tiles
    smashx_reduce_fn med_fn = nullptr; void* med_user = nullptr; std::vector<int> med_slot_h;
    float jobs = 0.f;
    // optional whole-domain outputs of forward sweeps (host arrays owned by the caller)
    float* h_qsim_domain = nullptr; float* h_net_prcp_domain = nullptr; int dom_sparse = 0;
    bool dom_q_active = false;       // the running sweep stores every cell's discharge (forward sweeps only)
    // tangent sweep (smashx_forward_d)
    bool tan_ready = false;
    float* d_qsim_d = nullptr;       // [ng][nt]
    float* d_tanplane[SMASHX_GNP + SMASHX_GNS] = {nullptr};   // full planes of the (denormalised) direction, jreg only
    // regularisation (sx_jreg.h): planes 0..15 = parameters, 16..23 = states
    bool tiled = false;
    int* d_active = nullptr;
    hipStream_t stream_j = nullptr;
    hipEvent_t ev_j = nullptr;
    float* d_jx[SMASHX_GNP + SMASHX_GNS] = {nullptr};     // values as compute_cost sees them (normalised when denormalize_forward)
    float* d_jb[SMASHX_GNP + SMASHX_GNS] = {nullptr};     // background
    float* d_jg[SMASHX_GNP + SMASHX_GNS] = {nullptr};     // d(wjreg jreg)/d(field): what COMPUTE_COST_B leaves in parameters_b / states_b
    float* d_jterm = nullptr; size_t jterm_planes = 0;
    float* d_jsum = nullptr;
    SxJregChains jchains{};
    bool jr_ready = false;                                 // planes of the current upload are on the device
    // timing
    std::vector<Launch> launches;
    std::vector<hipEvent_t> pool; size_t pool_used = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    smashx_timing timing{};
    int last_adjoint = 0;

    template <class T> int dmalloc(T** p, size_t count) {
        void* q = nullptr;
        const size_t b = std::max<size_t>(count, 1) * sizeof(T);
        hipError_t e = hipMalloc(&q, b);
        if (e != hipSuccess) return fail(SMASHX_E_HIP, std::string("hipMalloc(") + std::to_string(b) + " B): " + hipGetErrorString(e));
        allocs.push_back(q); alloc_bytes.push_back(b); bytes += (double)b; *p = (T*)q;
        return 0;
    }
    void dfree(void* q) {
        if (!q) return;
        for (size_t i = 0; i < allocs.size(); ++i)
            if (allocs[i] == q) { bytes -= (double)alloc_bytes[i]; allocs.erase(allocs.begin() + i); alloc_bytes.erase(alloc_bytes.begin() + i); break; }
        (void)hipFree(q);
    }
    template <class T> int upload_vec(T** p, const std::vector<T>& v) {
        int rc = dmalloc(p, v.size()); if (rc) return rc;
        if (!v.empty()) HIPCHK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
    hipEvent_t event() {
        if (pool_used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
        return pool[pool_used++];
    }
    hipStream_t cur = nullptr;       // stream of the launch being marked
    void mark_begin(int kind, hipStream_t st, double cellsteps = 0.0) {
        cur = st; Launch l; l.a = event(); l.b = nullptr; l.kind = kind; l.cellsteps = cellsteps;
        if (l.a) (void)hipEventRecord(l.a, st);
        launches.push_back(l);
    }
    void mark_end() { Launch& l = launches.back(); l.b = event(); if (l.b) (void)hipEventRecord(l.b, cur); }
};

namespace {

int set_device(const smashx_plan* p) {
    if (p->cfg.device >= 0) HIPCHK(hipSetDevice(p->cfg.device));
    return 0;
}

// Levels of the river tree a routing wavefront resolves inside ONE super-step (sx_plan.h "components"): SMASHX_SUBLEVELS for the
// rounds >= 1, SMASHX_SUBLEVELS_R0 for round 0.  Default 1 = one level per super-step.  Measured (profiles/r3_sublevels.json): with 4
// levels the groups are 3.5 x shallower (335 -> 96 stages) and every result is bit-identical, but a super-step then runs its body four
// times -- 0.7 us + 0.45 us per pass, against 1.15 us for one level -- so what the fill gains the steady state loses: 1024^2 x 8760
// 157.5 -> 160.7 ms, a 2048 x 1024 tile 354 -> 375 ms; in round 0 (vector-bound) it costs 22 ms.  Kept as a switch.
int sublevels_env() {
    const char* e = getenv("SMASHX_SUBLEVELS");
    const char* e0 = getenv("SMASHX_SUBLEVELS_R0");
    return (e ? std::max(1, atoi(e)) : 1) | ((e0 ? std::max(1, atoi(e0)) : 1) << 8);
}

int chunk_len(const smashx_plan* p, int c) { return std::min(p->Tc, p->nt - c * p->Tc); }

// allocate the time-chunk buffers; Tc from cfg or from free HBM
int ensure_chunk_buffers(smashx_plan* p, bool adjoint) {
    const int st = p->st;
    // taped levels per cell-step: hp, hft (+ hst in gr-c, + husl1 in vic-a) and, for gr-b / gr-c, the interception level hi.
    // hi depends on the forcing only: when its tape is what keeps the period from fitting in one storage chunk, the plan
    // keeps one checkpoint per SX_HIK steps instead and the reverse kernel rebuilds the levels block by block
    // (slower per step, but no recomputation of whole chunks: gr-c at 1024^2 x 8760 230 -> 204 ms).  SMASHX_HI_TAPE=0/1 forces it.
    const bool has_hi = (st == 2 || st == 3);
    const double ntape_full = (st == 5 ? 3.0 : st == 3 ? 4.0 : st == 2 ? 3.0 : 2.0);
    const double ntape_lean = has_hi ? ntape_full - 1.0 + 1.0 / SX_HIK : ntape_full;
    double ntape = ntape_full;
    if (!p->chunk_ready) {
        // a tile's neighbours must cut time exactly like it does (one message per sub-chunk): lengths sized from this
        // rank's own free HBM would differ between ranks and end in mismatched send/recv sizes
        if ((p->n_out > 0 || p->n_in > 0) && p->cfg.chunk_steps <= 0)
            return fail(SMASHX_E_ARG, "a tile with boundary series needs an explicit chunk_steps (identical on every rank; pipe_steps = 0 then means no sub-chunks)");
        const int nt16 = (p->nt + 15) / 16 * 16;
        int Tc = p->cfg.chunk_steps > 0 ? (p->cfg.chunk_steps + 15) / 16 * 16 : 0;
        {
            size_t fr = 0, tot = 0;
            HIPCHK(hipMemGetInfo(&fr, &tot));
            p->hbm_free_at_plan = (double)fr; p->hbm_total = (double)tot;
        }
        if (Tc == 0) {
            size_t fr = 0, tot = 0;
            HIPCHK(hipMemGetInfo(&fr, &tot));
            const double avail = (double)fr * 0.85 - 1.0e9;
            // floats per time step: qt, hr_imd and the taped levels of every cell, the exchange series (the staging rows of the chained
            // groups are optional and come out of the reserve: below)
            auto fit = [&](double nt_) { return (long)(avail / (4.0 * ((double)p->npad * (2 + nt_) + (double)std::max(p->sch.nxslots, 1)))) / 16 * 16; };
            long t = fit(ntape_full);
            if (has_hi && t < nt16) {
                // Dropping the hi tape costs the reverse kernel ~14 % (levels rebuilt block by block: 78.5 against 69 ms per 9.2e9
                // cell-steps, profiles/r3_*_2048*), one more storage chunk costs 1/C of a forward pass (~0.67 of a reverse pass): lean
                // only where it saves enough recomputation -- always when it lets the whole period fit one chunk (gr-c at 1024^2:
                // 230 -> 204 ms), not at 2048^2 where 4 lean chunks lose to 5 taped ones (784 -> see DESIGN.md 8)
                const long tl = fit(ntape_lean);
                auto nch = [&](long tc) { tc = std::max<long>(16, std::min<long>(nt16, tc)); return (double)((p->nt + tc - 1) / tc); };
                if (0.67 * (1.0 / nch(tl) - 1.0 / nch(t)) > 0.14) { p->hi_tape = false; t = tl; }
            }
            Tc = (int)std::max<long>(16, std::min<long>(nt16, t));
        }
        if (const char* e = getenv("SMASHX_HI_TAPE")) p->hi_tape = atoi(e) != 0;
        if (!has_hi) p->hi_tape = true;
        ntape = p->hi_tape ? ntape_full : ntape_lean;
        (void)ntape;
        Tc = std::min(Tc, nt16);
        {   // balance the chunks: same count, equal lengths
            const int nch = (p->nt + Tc - 1) / Tc;
            Tc = std::min(Tc, ((p->nt + nch - 1) / nch + 15) / 16 * 16);
        }
        p->Tc = Tc;
        p->A.Tc = Tc;
        p->A.nx = std::max(p->sch.nxslots, 1);
        p->nchunks = (p->nt + Tc - 1) / Tc;
        {   // optional pipeline sub-chunks (V stream || R stream), equal lengths, multiples of 16.  Default: none --
            // measured at 1024^2 x 8760: overlapping the latency-bound routing groups with the vertical kernel
            // costs more (routing fill per sub-chunk, wave slots taken from the vertical kernel) than it hides.
            int want = p->cfg.pipe_steps > 0 ? p->cfg.pipe_steps : Tc;
            const int nsub = std::max(1, (Tc + want / 2) / want);
            p->Tp = std::min(Tc, ((Tc + nsub - 1) / nsub + 15) / 16 * 16);
        }
        int rc;
        if ((rc = p->dmalloc(&p->A.qtT, (size_t)p->npad * Tc))) return rc;
        if ((rc = p->dmalloc(&p->A.xT, (size_t)std::max(p->sch.nxslots, 1) * Tc))) return rc;
        HIPCHK(hipMemsetAsync(p->A.qtT, 0, (size_t)p->npad * Tc * 4, p->stream));
        HIPCHK(hipMemsetAsync(p->A.xT, 0, (size_t)std::max(p->sch.nxslots, 1) * Tc * 4, p->stream));
        p->chunk_ready = true;
    }
    if (p->h_qsim_domain && !p->A.qdT) {
        int rc;
        if ((rc = p->dmalloc(&p->A.qdT, (size_t)p->npad * p->Tc))) return rc;
    }
    if (adjoint && !p->adj_ready) {
        int rc;
        const size_t cs = (size_t)p->npad * p->Tc;
        if (p->A.qsk) { p->dfree(p->A.qsk); p->A.qsk = nullptr; }      // (taken by a forward-only sweep before: the tapes go first)
        {   // the hr_imd tape is written and read by the routing kernels only: its rows are shifted by the cell's stage when the
            // extra rows (the deepest group's stages) fit beside everything else -- decided per plan, never changes results
            const size_t extra = (size_t)p->sch.max_stage * p->npad * 4;
            const double ntp = 3.0 + ((st == 5 || ((st == 2 || st == 3) && p->hi_tape)) ? 1.0 : (st == 2 || st == 3) ? 1.0 / SX_HIK : 0.0) + (st == 3 ? 1.0 : 0.0);
            size_t fr = 0, tot = 0;
            HIPCHK(hipMemGetInfo(&fr, &tot));
            const double need = 4.0 * ((double)cs * ntp + (double)extra + (double)p->npad * (14 + (p->nchunks > 1 ? 5.0 * p->nchunks : 0.0))
                                       + 2.0 * (double)std::max(p->ngc, 1) * p->nt) + 3.0e9;
            const char* e = getenv("SMASHX_HR_SKEW");
            p->A.hr_skew = (e ? atoi(e) != 0 : true) && (double)fr > need;
            if ((rc = p->dmalloc(&p->A.hrT, cs + (p->A.hr_skew ? extra : 0)))) return rc;
        }
        if ((rc = p->dmalloc(&p->A.tape_hp, cs))) return rc;
        if ((rc = p->dmalloc(&p->A.tape_hft, cs))) return rc;
        if (st == 5 || ((st == 2 || st == 3) && p->hi_tape)) { if ((rc = p->dmalloc(&p->A.tape_hi, cs))) return rc; }
        else if (st == 2 || st == 3) { if ((rc = p->dmalloc(&p->A.ckpt_hi, cs / SX_HIK))) return rc; }
        if (st == 3) { if ((rc = p->dmalloc(&p->A.tape_hst, cs))) return rc; }
        if (p->nchunks > 1) { if ((rc = p->dmalloc(&p->ckpt, (size_t)p->nchunks * 5 * p->npad))) return rc; }
        float** g[14] = {&p->A.ci_b, &p->A.cp_b, &p->A.cft_b, &p->A.cst_b, &p->A.exc_b, &p->A.lr_b,
                         &p->A.hi_b, &p->A.hp_b, &p->A.hft_b, &p->A.hst_b, &p->A.hlr_b, &p->A.px_b[0], &p->A.px_b[1], &p->A.px_b[2]};
        for (auto q : g) if ((rc = p->dmalloc(q, (size_t)p->npad))) return rc;
        if ((rc = p->dmalloc(&p->A.qgb, (size_t)std::max(p->ngc, 1) * p->nt))) return rc;
        if ((rc = p->dmalloc(&p->d_qsim_b, (size_t)std::max(p->ng, 1) * p->nt))) return rc;
        p->adj_ready = true;
    }
    if (p->A.ncs > 0 && !p->A.qsk) {
        // staging rows of the chained groups, [Tc / 4 + deepest chained group + 1][ncs] float4: a copy that makes the chained launches
        // faster, nothing depends on it -- it takes what the reserve holds once everything else is in place (2048^2: 7.4 GB of the
        // 34 GB the chunk length leaves free) and is left out when that is not there (the chained launches keep the plain rows)
        const size_t need = (size_t)p->A.ncs * 16 * ((size_t)p->Tc / 4 + p->stg_rows_extra + 1);
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        if ((double)fr > (double)need + 2.0e9) { int rc; if ((rc = p->dmalloc(&p->A.qsk, need / 4))) return rc; }
    }
    return 0;
}

// view of the chunk buffers shifted to local step `off` (multiple of 4) of the current storage chunk
SxDeviceArrays view_at(const smashx_plan* p, int off) {
    SxDeviceArrays B = p->A;
    B.k0 = 0; B.k1 = p->n;
    const size_t q = (size_t)(off / 4);
    B.qtT = p->A.qtT + q * p->npad * 4;
    if (p->A.hrT) B.hrT = p->A.hrT + q * p->npad * 4;
    if (p->A.qdT) B.qdT = p->A.qdT + q * p->npad * 4;
    if (!p->chain) B.qsk = nullptr;      // the launch-per-round fallback reads and writes the plain rows
    B.xT = p->A.xT + q * p->A.nx * 4;
    if (p->A.qtdT) B.qtdT = p->A.qtdT + q * p->npad * 4;      // tangent sweep (smashx_forward_d)
    if (p->A.xdT) B.xdT = p->A.xdT + q * p->A.nx * 4;
    if (p->A.tape_hp) B.tape_hp = p->A.tape_hp + (size_t)off * p->npad;
    if (p->A.tape_hft) B.tape_hft = p->A.tape_hft + (size_t)off * p->npad;
    if (p->A.tape_hi) B.tape_hi = p->A.tape_hi + (size_t)off * p->npad;
    if (p->A.ckpt_hi) B.ckpt_hi = p->A.ckpt_hi + (size_t)(off / SX_HIK) * p->npad;
    if (p->A.tape_hst) B.tape_hst = p->A.tape_hst + (size_t)off * p->npad;
    return B;
}

// vertical launches on the cell range [k0, k1) (cells are numbered in routing-group order: round 0 first)
template <int ST>
void launch_vert_fwd(smashx_plan* p, const SxDeviceArrays& B, bool tape, int t0, int T) {
    const dim3 grid((B.k1 - B.k0 + SX_VBLOCK - 1) / SX_VBLOCK), block(SX_VBLOCK);
    p->mark_begin(0, p->stream, (double)(B.k1 - B.k0) * T);
    const bool cf = B.prcp16 != nullptr;
    const size_t vl = p->vlds_fwd;       // unused dynamic LDS that caps the resident vertical workgroups per compute unit
    if (tape) { if (cf) hipLaunchKernelGGL((sx_k_vert_fwd<ST, true, true>), grid, block, vl, p->stream, B, t0, T);
                else    hipLaunchKernelGGL((sx_k_vert_fwd<ST, true, false>), grid, block, vl, p->stream, B, t0, T); }
    else      { if (cf) hipLaunchKernelGGL((sx_k_vert_fwd<ST, false, true>), grid, block, vl, p->stream, B, t0, T);
                else    hipLaunchKernelGGL((sx_k_vert_fwd<ST, false, false>), grid, block, vl, p->stream, B, t0, T); }
    p->mark_end();
}
void vert_fwd(smashx_plan* p, int off, bool tape, int t0, int T, int k0 = 0, int k1 = -1) {
    SxDeviceArrays B = view_at(p, off);
    B.k0 = k0; B.k1 = k1 < 0 ? p->n : k1;
    if (B.k1 <= B.k0) return;
    switch (p->st) {
        case 1: launch_vert_fwd<1>(p, B, tape, t0, T); break;
        case 2: launch_vert_fwd<2>(p, B, tape, t0, T); break;
        case 3: launch_vert_fwd<3>(p, B, tape, t0, T); break;
        case 5: {
            const dim3 grid((B.k1 - B.k0 + SX_VBLOCK - 1) / SX_VBLOCK), block(SX_VBLOCK);
            p->mark_begin(0, p->stream, (double)(B.k1 - B.k0) * T);
            const bool cf = B.prcp16 != nullptr;
            if (tape) { if (cf) hipLaunchKernelGGL((sx_k_vert_fwd_vic<true, true>), grid, block, 0, p->stream, B, t0, T);
                        else    hipLaunchKernelGGL((sx_k_vert_fwd_vic<true, false>), grid, block, 0, p->stream, B, t0, T); }
            else      { if (cf) hipLaunchKernelGGL((sx_k_vert_fwd_vic<false, true>), grid, block, 0, p->stream, B, t0, T);
                        else    hipLaunchKernelGGL((sx_k_vert_fwd_vic<false, false>), grid, block, 0, p->stream, B, t0, T); }
            p->mark_end();
        } break;
        default: launch_vert_fwd<4>(p, B, tape, t0, T); break;
    }
}
void vert_adj(smashx_plan* p, int off, int t0, int T, int k0 = 0, int k1 = -1) {
    SxDeviceArrays B = view_at(p, off);
    B.k0 = k0; B.k1 = k1 < 0 ? p->n : k1;
    if (B.k1 <= B.k0) return;
    const dim3 grid((B.k1 - B.k0 + SX_VBLOCK - 1) / SX_VBLOCK), block(SX_VBLOCK);
    p->mark_begin(3, p->stream, (double)(B.k1 - B.k0) * T);
    const bool cf = B.prcp16 != nullptr;
    const size_t vl = p->vlds_adj;
#define SX_VADJ(ST) do { if (cf) hipLaunchKernelGGL((sx_k_vert_adj<ST, true>), grid, block, vl, p->stream, B, t0, T); \
                         else hipLaunchKernelGGL((sx_k_vert_adj<ST, false>), grid, block, vl, p->stream, B, t0, T); } while (0)
    switch (p->st) {
        case 1: SX_VADJ(1); break;
        case 2: SX_VADJ(2); break;
        case 3: SX_VADJ(3); break;
        case 5: if (cf) hipLaunchKernelGGL(sx_k_vert_adj_vic<true>, grid, block, 0, p->stream, B, t0, T);
                else hipLaunchKernelGGL(sx_k_vert_adj_vic<false>, grid, block, 0, p->stream, B, t0, T);
                break;
        default: SX_VADJ(4); break;
    }
#undef SX_VADJ
    p->mark_end();
}
double round_cells(const smashx_plan* p, int r0, int r1) {   // cells (inlets excluded) of the groups of rounds [r0, r1)
    double c = 0.0;
    for (int r = r0; r < r1; ++r) c += p->round_ncells[r];
    return c;
}
int chain_first(const smashx_plan* p) {       // first chained round (nrounds: none)
    const int nr = p->sch.nrounds;
    return (p->chain && nr - p->chain_from >= 2) ? p->chain_from : nr;
}
// counters of a chained launch: the groups' progress and the ticket counter (the stall flag lives for the whole sweep)
void reset_chain_counters(smashx_plan* p, hipStream_t st) {
    (void)hipMemsetAsync(p->A.prog, 0, (size_t)p->sch.ngroups * sizeof(int), st);
    (void)hipMemsetAsync(p->A.prog + p->sch.ngroups + 1, 0, sizeof(int), st);
}
// the un-chained rounds [0, chain_first) of one forward pass over [t0, t0 + T): one launch per round on stream st
void route_fwd_rounds(smashx_plan* p, int off, bool tape, int t0, int T, hipStream_t st) {
    SxDeviceArrays B = view_at(p, off);
    if (!p->dom_q_active) B.qdT = nullptr;
    const size_t lds = (size_t)2 * p->M * sizeof(float4);
    const int cf = chain_first(p);
    for (int r = 0; r < cf; ++r) {
        const int g0 = p->sch.round_group_begin[r], ngr = p->sch.round_group_begin[r + 1] - g0;
        p->mark_begin(1, st, round_cells(p, r, r + 1) * T);
        if (tape) hipLaunchKernelGGL((sx_k_route_fwd<true, false>), dim3(ngr), dim3(p->M), lds, st, B, g0, g0 + ngr, t0, T);
        else      hipLaunchKernelGGL((sx_k_route_fwd<false, false>), dim3(ngr), dim3(p->M), lds, st, B, g0, g0 + ngr, t0, T);
        p->mark_end();
    }
}
// the chained rounds in ONE launch (tickets: sx_kernels.h)
// inputs_read (optional): recorded on st as soon as the pass has read the last of qtT -- after the copy into the staging rows when the
// chained launch runs on those (it then touches no buffer of the vertical kernels), else left alone (the caller records after the pass)
bool route_fwd_chained(smashx_plan* p, int off, bool tape, int t0, int T, hipStream_t st, hipEvent_t inputs_read = nullptr) {
    const int nr = p->sch.nrounds, cf = chain_first(p);
    if (cf >= nr) return false;
    SxDeviceArrays B = view_at(p, off);
    if (!p->dom_q_active) B.qdT = nullptr;
    const size_t lds = (size_t)2 * p->M * sizeof(float4);
    const int g0 = p->sch.round_group_begin[cf], g1 = p->sch.ngroups;
    reset_chain_counters(p, st);
    const int grid = g1 - g0;
    p->mark_begin(5, st, round_cells(p, cf, nr) * T);       // (the chained launch's time includes its copy pass)
    if (B.qsk)       // the chained groups' inputs -- their cells' runoff, the series handed up to them -- into the staging rows
        hipLaunchKernelGGL((sx_k_chain_transpose<true>), dim3((p->stg_blocks + SX_STG_WAVES - 1) / SX_STG_WAVES), dim3(64 * SX_STG_WAVES), p->stg_lds * SX_STG_WAVES, st, B, p->stg, g0, (T + SX_BT - 1) / SX_BT, p->stg_blocks, (int)p->stg_lds);
    const bool early = B.qsk && inputs_read && !B.qdT;
    if (early) (void)hipEventRecord(inputs_read, st);
    if (tape) hipLaunchKernelGGL((sx_k_route_fwd<true, true>), dim3(grid), dim3(p->M), lds, st, B, g0, g1, t0, T);
    else      hipLaunchKernelGGL((sx_k_route_fwd<false, true>), dim3(grid), dim3(p->M), lds, st, B, g0, g1, t0, T);
    p->mark_end();
    p->chain_used = true;
    return early;
}
// Routing launches of one pass.  Rounds below chain_first keep one launch per round (they are wide and
// HBM-bound); the narrow, latency-bound rounds from there on run chained inside a single launch
// (sx_kernels.h "rounds chained inside one launch"), which turns their sum into roughly the longest of them.
bool route_fwd(smashx_plan* p, int off, bool tape, int t0, int T, hipEvent_t inputs_read = nullptr) {
    route_fwd_rounds(p, off, tape, t0, T, p->stream_r);
    return route_fwd_chained(p, off, tape, t0, T, p->stream_r, inputs_read);
}
void route_adj_chained(smashx_plan* p, int off, int t0, int T, hipStream_t st) {
    const int nr = p->sch.nrounds, cf = chain_first(p);
    if (cf >= nr) return;
    const SxDeviceArrays B = view_at(p, off);
    const size_t lds = (size_t)2 * p->M * sizeof(float4);
    const int g0 = p->sch.round_group_begin[cf], g1 = p->sch.ngroups;
    reset_chain_counters(p, st);
    const int grid = g1 - g0;
    p->mark_begin(6, st, round_cells(p, cf, nr) * T);
    hipLaunchKernelGGL((sx_k_route_adj<true>), dim3(grid), dim3(p->M), lds, st, B, g0, g1, t0, T);
    if (B.qsk)       // qt_b of the chained cells and the adjoint series that leave the chain: from the staging rows to where the vertical kernel,
                     // round 0 and the exchange expect them
        hipLaunchKernelGGL((sx_k_chain_transpose<false>), dim3((p->stg_blocks + SX_STG_WAVES - 1) / SX_STG_WAVES), dim3(64 * SX_STG_WAVES), p->stg_lds * SX_STG_WAVES, st, B, p->stg, g0, (T + SX_BT - 1) / SX_BT, p->stg_blocks, (int)p->stg_lds);
    p->mark_end();
    p->chain_used = true;
}
void route_adj_rounds(smashx_plan* p, int off, int t0, int T, hipStream_t st) {
    const SxDeviceArrays B = view_at(p, off);
    const size_t lds = (size_t)2 * p->M * sizeof(float4);
    const int cf = chain_first(p);
    for (int r = cf - 1; r >= 0; --r) {
        const int g0 = p->sch.round_group_begin[r], ngr = p->sch.round_group_begin[r + 1] - g0;
        p->mark_begin(2, st, round_cells(p, r, r + 1) * T);
        hipLaunchKernelGGL((sx_k_route_adj<false>), dim3(ngr), dim3(p->M), lds, st, B, g0, g0 + ngr, t0, T);
        p->mark_end();
    }
}
void route_adj(smashx_plan* p, int off, int t0, int T) {
    route_adj_chained(p, off, t0, T, p->stream_r);
    route_adj_rounds(p, off, t0, T, p->stream_r);
}

SxCostArgs cost_args(smashx_plan* p, float jobs_b) {
    SxCostArgs C{};
    C.ng = p->ng; C.nt = p->nt; C.s0 = p->opt.optimize_start_step - 1; C.njf = p->opt.njf;
    for (int j = 0; j < SX_MAXJF; ++j) { C.jobs_fun[j] = p->opt.jobs_fun[j]; C.wjobs_fun[j] = p->opt.wjobs_fun[j]; }
    C.dt = p->cfg.dt; C.dx = p->cfg.dx;
    C.qg = p->A.qg; C.qgb = p->A.qgb; C.ngc = p->ngc;
    C.gauge_gid = p->d_gauge_gid; C.gauge_flwacc = p->d_gauge_flwacc; C.area = p->d_area; C.wgauge = p->d_wgauge;
    C.qobs = p->d_qobs; C.qsim_b = p->d_qsim_b; C.sums = p->d_sums; C.coef = p->d_coef; C.out = p->d_cost_out;
    C.med = p->d_med; C.med_idx = p->d_med_idx;
    C.nslots = p->med_nslots; C.slot = p->d_med_slot; C.medx = p->d_medx; C.medx_idx = p->d_medx_idx;
    C.jobs_b = jobs_b;
    return C;
}

int run_cost(smashx_plan* p, int adjoint, float cost_b) {
    if (p->ng == 0 && p->med_nslots == 0) return 0;      // (a tile without gauges still takes part in the sum of the median's slots)
    SxCostArgs C = cost_args(p, cost_b);
    p->mark_begin(4, p->stream_r);
    if (p->ng > 0) hipLaunchKernelGGL(sx_k_cost_sums, dim3(p->ng), dim3(64), 0, p->stream_r, C);
    if (p->med_nslots > 0) {
        // the median spans several tiles: local gauge_jobs into their slots, slots summed over the ranks, then the median of all
        HIPCHK(hipMemsetAsync(p->d_medx, 0, (size_t)p->med_nslots * sizeof(float), p->stream_r));
        hipLaunchKernelGGL(sx_k_cost_final, dim3(1), dim3(1), 0, p->stream_r, C, adjoint, 1);
        if (p->xcomm) {
            NCCLCHK(rccl().AllReduce(p->d_medx, p->d_medx, (size_t)p->med_nslots, ncclFloat, ncclSum, p->xcomm->comm, p->stream_r));
        } else if (p->med_fn) {
            std::vector<float> h(p->med_nslots);
            HIPCHK(hipMemcpyAsync(h.data(), p->d_medx, h.size() * sizeof(float), hipMemcpyDeviceToHost, p->stream_r));
            HIPCHK(hipStreamSynchronize(p->stream_r));
            if (p->med_fn(p->med_user, h.data(), p->med_nslots)) return fail(SMASHX_E_ARG, "median reduce callback failed");
            HIPCHK(hipMemcpyAsync(p->d_medx, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, p->stream_r));
            HIPCHK(hipStreamSynchronize(p->stream_r));
        } else return fail(SMASHX_E_STATE, "median over the gauges of several tiles: no exchange to sum the slots (smashx_set_exchange or a reduce_fn)");
        hipLaunchKernelGGL(sx_k_cost_final, dim3(1), dim3(1), 0, p->stream_r, C, adjoint, 2);
    } else
        hipLaunchKernelGGL(sx_k_cost_final, dim3(1), dim3(1), 0, p->stream_r, C, adjoint, 0);
    if (adjoint && p->ng > 0) {
        const dim3 b(256), g1((p->nt + 255) / 256, p->ng), g2((p->nt + 255) / 256, p->ngc);
        hipLaunchKernelGGL(sx_k_cost_seeds, g1, b, 0, p->stream_r, C);
        hipLaunchKernelGGL(sx_k_cost_cellseeds, g2, b, 0, p->stream_r, C);
    }
    p->mark_end();
    return 0;
}

// the reservoir levels live on the V stream, the routing store hlr (index 4) on the R stream
int copy_states(smashx_plan* p, float* const dst[5], float* const src[5]) {
    for (int i = 0; i < 5; ++i)
        if (state_field(p->st, i) >= 0)
            HIPCHK(hipMemcpyAsync(dst[i], src[i], (size_t)p->npad * 4, hipMemcpyDeviceToDevice, i == 4 ? p->stream_r : p->stream));
    return 0;
}
int restore_states(smashx_plan* p, float* const src[5]) {
    float* dst[5] = {p->A.hi, p->A.hp, p->A.hft, p->A.hst, p->A.hlr};
    return copy_states(p, dst, src);
}

}  // namespace

extern "C" {

const char* smashx_last_error(void) { return g_err.c_str(); }

int smashx_abi_sizes(int sizes[7]) {
    if (sizes) {
        sizes[0] = (int)sizeof(smashx_config); sizes[1] = (int)sizeof(smashx_mesh); sizes[2] = (int)sizeof(smashx_options);
        sizes[3] = (int)sizeof(smashx_parameters); sizes[4] = (int)sizeof(smashx_states); sizes[5] = (int)sizeof(smashx_costs);
        sizes[6] = (int)sizeof(smashx_timing);
    }
    return SMASHX_ABI_VERSION;
}

int smashx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int smashx_plan_create(const smashx_config* cfg, const smashx_mesh* mesh, smashx_plan** out) {
    if (!cfg || !mesh || !out) return fail(SMASHX_E_ARG, "null argument");
    *out = nullptr;
    if (cfg->structure < 1 || cfg->structure > 5) return fail(SMASHX_E_UNSUPPORTED, "structure must be gr-a/b/c/d or vic-a");
    if (cfg->nrow <= 0 || cfg->ncol <= 0 || cfg->nt <= 0 || cfg->ng < 0 || !(cfg->dt > 0.f) || !(cfg->dx > 0.f))
        return fail(SMASHX_E_ARG, "bad sizes in smashx_config");
    if (!mesh->flwdir || !mesh->flwacc || !mesh->active_cell || (cfg->ng > 0 && (!mesh->gauge_pos || !mesh->area)))
        return fail(SMASHX_E_ARG, "null mesh array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SMASHX_E_NODEVICE, "no HIP device: libsmashx has no CPU fallback");
    smashx_plan* p = new smashx_plan();
    p->cfg = *cfg;
    if (set_device(p)) { delete p; return SMASHX_E_HIP; }
    p->M = cfg->group_size > 0 ? cfg->group_size : 512;
    if (p->M % 64 != 0 || p->M > SX_MAXGROUP || p->M < 64) { delete p; return fail(SMASHX_E_ARG, "group_size must be a multiple of 64 in [64, 512]"); }
    const bool rect_tile = cfg->tile[1] > cfg->tile[0] && cfg->tile[3] > cfg->tile[2];
    const bool tiled = rect_tile || mesh->owner_mask;
    const int rc0 = sx_build_schedule(cfg->nrow, cfg->ncol, mesh->flwdir, mesh->active_cell, cfg->ng, mesh->gauge_pos, p->M,
                                      rect_tile ? cfg->tile : nullptr, p->sch, mesh->owner_mask, sublevels_env());
    if (rc0 != 0) { std::string e = p->sch.error; delete p; return fail(rc0 == -5 ? SMASHX_E_MESH : SMASHX_E_ARG, e); }
    p->tiled = tiled;
    p->n = p->sch.n; p->npad = (p->n + SX_VBLOCK - 1) / SX_VBLOCK * SX_VBLOCK;
    p->nt = cfg->nt; p->ng = cfg->ng; p->st = cfg->structure; p->n2 = (long)cfg->nrow * cfg->ncol;
    int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { smashx_plan_destroy(p); return rc; } } while (0)
    if (hipStreamCreate(&p->stream) != hipSuccess || hipStreamCreate(&p->stream_r) != hipSuccess) { delete p; return fail(SMASHX_E_HIP, "hipStreamCreate failed"); }
    (void)hipEventCreate(&p->ev0); (void)hipEventCreate(&p->ev1);
    if (hipStreamCreate(&p->stream_j) != hipSuccess || hipEventCreate(&p->ev_j) != hipSuccess) { smashx_plan_destroy(p); return fail(SMASHX_E_HIP, "hipStreamCreate failed"); }
    SxDeviceArrays& A = p->A;
    A.n = p->n; A.npad = p->npad; A.nt = p->nt; A.dt = cfg->dt; A.dx = cfg->dx; A.Tc = 0;
    if (getenv("SMASHX_VERBOSE"))
        for (int r = 0; r < p->sch.nrounds; ++r) {       // diagnostics: shape of the routing schedule
            long slots = 0, inlets = 0, dsum = 0; int dmx = 0;
            const int ga = p->sch.round_group_begin[r], gb = p->sch.round_group_begin[r + 1];
            for (int g = ga; g < gb; ++g) {
                dsum += p->sch.g_dmax[g]; dmx = std::max(dmx, p->sch.g_dmax[g]);
                for (int q = p->sch.g_slot_begin[g]; q < p->sch.g_slot_begin[g + 1]; ++q) { if (p->sch.s_cell[q] == INT_MIN) continue; ++slots; if (p->sch.s_cell[q] < 0) ++inlets; }
            }
            fprintf(stderr, "smashx: routing round %d: %d groups, %ld slots (%ld inlets), depth mean %.1f max %d\n", r, gb - ga, slots, inlets,
                    gb > ga ? (double)dsum / (gb - ga) : 0.0, dmx);
        }
    // schedule tables
    int *d1, *d2, *d3, *d4, *d5, *d6, *d7, *d8;
    TRY(p->upload_vec(&d1, p->sch.g_slot_begin)); A.g_slot_begin = d1;
    TRY(p->upload_vec(&d2, p->sch.g_dmax)); A.g_dmax = d2;
    TRY(p->upload_vec(&d3, p->sch.s_cell)); A.s_cell = d3;
    TRY(p->upload_vec(&d4, p->sch.s_stage)); A.s_stage = d4;
    TRY(p->upload_vec(&d5, p->sch.s_child)); A.s_child = d5;
    TRY(p->upload_vec(&d6, p->sch.s_ccount)); A.s_ccount = d6;
    { int *e1, *e2; TRY(p->upload_vec(&e1, p->sch.s_sub)); A.s_sub = e1; TRY(p->upload_vec(&e2, p->sch.s_wsub)); A.s_wsub = e2; }
    TRY(p->upload_vec(&d7, p->sch.s_parent)); A.s_parent = d7;
    TRY(p->upload_vec(&d8, p->sch.s_xout)); A.s_xout = d8;
    {   // chained rounds (the rounds from SMASHX_CHAIN_FROM, default 1, share one launch); SMASHX_CHAIN_ROUNDS=0
        // restores the launch-per-round schedule for A/B measurements and bisection
        int *d9, *d10;
        TRY(p->upload_vec(&d9, p->sch.x_prod_group)); A.x_prod = d9;
        TRY(p->upload_vec(&d10, p->sch.x_cons_group)); A.x_cons = d10;
        TRY(p->dmalloc(&A.prog, (size_t)p->sch.ngroups + SX_PROG_EXTRA));
        if (hipMemset(A.prog, 0, ((size_t)p->sch.ngroups + SX_PROG_EXTRA) * sizeof(int)) != hipSuccess) { smashx_plan_destroy(p); return fail(SMASHX_E_HIP, "hipMemset"); }
        A.ngroups = p->sch.ngroups;
        const char* e = getenv("SMASHX_CHAIN_ROUNDS");
        p->chain = !(e && e[0] == '0');
        if (const char* dv = getenv("SMASHX_DEBUG_VLDS")) {       // "n" or "nfwd,nadj"
            p->vlds_fwd = p->vlds_adj = (size_t)std::max(0, atoi(dv));
            if (const char* c2 = strchr(dv, ',')) p->vlds_adj = (size_t)std::max(0, atoi(c2 + 1));
        }
        p->round_ncells.assign(p->sch.nrounds, 0.0);
        for (int r = 0; r < p->sch.nrounds; ++r)
            for (int g = p->sch.round_group_begin[r]; g < p->sch.round_group_begin[r + 1]; ++g)
                for (int q = p->sch.g_slot_begin[g]; q < p->sch.g_slot_begin[g + 1]; ++q) p->round_ncells[r] += p->sch.s_cell[q] >= 0;
        const char* cfm = getenv("SMASHX_CHAIN_FROM");
        p->chain_from = cfm ? std::max(0, atoi(cfm)) : 1;
        {   // staging rows of the chained groups: wave-blocks of 64 consecutive slots of one group, FIFO offsets per slot and direction
            A.qsk = nullptr; A.cs0 = 0; A.ncs = 0;
            const int cf = chain_first(p);
            {   // The copy pays where the chained launch is bound by its memory instructions: several waves of groups per compute unit
                // (2048^2: 1293 groups, sweep -12 ms).  A launch the device holds at once is bound by the latency of its chain of stages
                // instead, gains less than the two passes cost (1024^2: 316 groups, chained launches -5 ms, passes +6.5 ms) and keeps the
                // plain rows.  SMASHX_CHAIN_STAGE=0/1 overrides.
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                const int nchained = cf < p->sch.nrounds ? p->sch.ngroups - p->sch.round_group_begin[cf] : 0;
                p->chain_stage = nchained >= 2 * cus;
                if (const char* cs = getenv("SMASHX_CHAIN_STAGE")) p->chain_stage = atoi(cs) != 0;
            }
            if (p->chain_stage && cf < p->sch.nrounds) {
                const int gc = p->sch.round_group_begin[cf];
                const int cs0 = p->sch.g_slot_begin[gc], ncs = p->sch.nslots - cs0;
                std::vector<int> slot0, cnt, smn, spr, fg((size_t)std::max(ncs, 1), 0), fs((size_t)std::max(ncs, 1), 0);
                size_t lds = 0;
                int extra = 0;
                for (int g = gc; g < p->sch.ngroups; ++g) {
                    extra = std::max(extra, p->sch.g_dmax[g]);
                    const int sb = p->sch.g_slot_begin[g], m = p->sch.g_slot_begin[g + 1] - sb;
                    for (int o = 0; o < m; o += 64) {
                        const int nl = std::min(64, m - o);
                        int lo = INT_MAX, hi = 0;
                        for (int q = sb + o; q < sb + o + nl; ++q) { lo = std::min(lo, p->sch.s_stage[q]); hi = std::max(hi, p->sch.s_stage[q]); }
                        size_t og = 0, os = 0;
                        for (int q = sb + o; q < sb + o + nl; ++q) {
                            const int d = p->sch.s_stage[q] - lo;
                            fg[q - cs0] = (int)og; og += (size_t)d + 1;
                            fs[q - cs0] = (int)os; os += (size_t)(hi - lo - d) + 1;
                        }
                        lds = std::max(lds, std::max(og, os) * sizeof(float4));
                        slot0.push_back(sb + o); cnt.push_back(nl); smn.push_back(lo); spr.push_back(hi - lo);
                    }
                }
                if (lds * SX_STG_WAVES <= 160 * 1024 && lds <= 64 * 1024 && !slot0.empty()) {      // (a schedule whose wave-blocks span more stages than that keeps the plain rows)
                    int *d0, *d1, *d2, *d3, *d4, *d5;
                    TRY(p->upload_vec(&d0, slot0)); TRY(p->upload_vec(&d1, cnt)); TRY(p->upload_vec(&d2, smn)); TRY(p->upload_vec(&d3, spr));
                    TRY(p->upload_vec(&d4, fg)); TRY(p->upload_vec(&d5, fs));
                    p->stg = SxStageTables{d0, d1, d2, d3, d4, d5};
                    p->stg_blocks = (int)slot0.size(); p->stg_rows_extra = extra; p->stg_lds = (lds + 15) / 16 * 16;
                    // the copy kernels may use all of a CU's LDS (160 KiB; the attribute belongs to the kernel, not to the plan: the same
                    // value for every plan of the process).  Where that is refused, a plan that needs more than the default keeps the plain rows
                    static const bool lds_ok = []() {
                        return hipFuncSetAttribute(reinterpret_cast<const void*>(&sx_k_chain_transpose<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                            && hipFuncSetAttribute(reinterpret_cast<const void*>(&sx_k_chain_transpose<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                    }();
                    if (lds_ok || p->stg_lds * SX_STG_WAVES <= 48 * 1024) { A.cs0 = cs0; A.ncs = ncs; }
                    else (void)hipGetLastError();
                }
            }
        }
        A.spin_limit = SX_SPIN_LIMIT; A.mute_group = -1;
        if (const char* sl = getenv("SMASHX_SPIN_LIMIT")) A.spin_limit = std::max(1, atoi(sl));
        if (const char* mg = getenv("SMASHX_DEBUG_MUTE_GROUP")) A.mute_group = atoi(mg);     // tests of the stall path only
        if (A.mute_group < -1) A.mute_group = p->sch.round_group_begin[std::min(p->chain_from, p->sch.nrounds - 1)];   // "the first chained group"
        A.gtime = nullptr;
        const char* tr = getenv("SMASHX_TRACE_GROUPS");
        if (tr && tr[0] == '1') {
            TRY(p->dmalloc(&A.gtime, (size_t)4 * p->sch.ngroups));
            if (hipMemset(A.gtime, 0, (size_t)4 * p->sch.ngroups * sizeof(long long)) != hipSuccess) { smashx_plan_destroy(p); return fail(SMASHX_E_HIP, "hipMemset"); }
        }
    }
    TRY(p->upload_vec(&p->d_cell_flat, p->sch.cell_flat));
    TRY(p->upload_vec(&p->d_active, std::vector<int>(mesh->active_cell, mesh->active_cell + p->n2)));
    TRY(p->dmalloc(&p->d_jsum, (size_t)SX_JREG_MAXCHAIN));
    p->n_out = (int)p->sch.out_x.size(); p->n_in = (int)p->sch.in_x.size();
    TRY(p->upload_vec(&p->d_out_x, p->sch.out_x.empty() ? std::vector<int>(1, 0) : p->sch.out_x));
    TRY(p->upload_vec(&p->d_in_x, p->sch.in_x.empty() ? std::vector<int>(1, 0) : p->sch.in_x));
    // per-cell mesh data
    std::vector<int> facc(p->npad, 1), cg(p->npad, -1);
    for (int k = 0; k < p->n; ++k) facc[k] = mesh->flwacc[p->sch.cell_flat[k]];
    p->gauge_gid.assign(p->ng, -1);
    std::vector<int> gfl(std::max(p->ng, 1), 1);
    for (int g = 0; g < p->ng; ++g) {
        const int k = p->sch.gauge_k[g];
        if (cg[k] < 0) cg[k] = p->ngc++;
        p->gauge_gid[g] = cg[k];
        gfl[g] = mesh->flwacc[p->sch.cell_flat[k]];
    }
    TRY(p->upload_vec(&A.flwacc, facc));
    TRY(p->upload_vec(&A.cell_gauge, cg));
    // sparse forcing index: k -> position along path over active cells (mw_sparse_storage.f90:12-49)
    if (mesh->path) {
        std::vector<int> sp(p->n, -1);
        int ind = 0;
        for (long i = 0; i < p->n2; ++i) {
            const int row = mesh->path[2 * i], col = mesh->path[2 * i + 1];
            if (row < 0 || col < 0 || row >= cfg->nrow || col >= cfg->ncol) continue;
            const long c = row + (long)col * cfg->nrow;
            if (mesh->active_cell[c] == 1) { const int k = p->sch.k_of_flat[c]; if (k >= 0 && sp[k] < 0) sp[k] = ind; ++ind; }
        }
        bool ok = true;
        for (int k = 0; k < p->n; ++k) ok &= sp[k] >= 0;
        if (ok) { TRY(p->upload_vec(&p->d_sparse_idx, sp)); p->n_sparse = ind; }
    }
    // parameters, states, routing invariants
    float** pf[NPS] = {&A.ci, &A.cp, &A.cft, &A.cst, &A.exc, &A.lr, &A.px[0], &A.px[1], &A.px[2]};
    for (auto q : pf) TRY(p->dmalloc(q, (size_t)p->npad));
    float** sf[5] = {&A.hi, &A.hp, &A.hft, &A.hst, &A.hlr};
    for (auto q : sf) TRY(p->dmalloc(q, (size_t)p->npad));
    for (int i = 0; i < 5; ++i) TRY(p->dmalloc(&p->st0[i], (size_t)p->npad));
    float** rf[4] = {&A.rt_a, &A.rt_f, &A.rt_denf, &A.rt_denb};
    for (auto q : rf) TRY(p->dmalloc(q, (size_t)p->npad));
    for (int i = 0; i < NPS; ++i) if (param_field(p->st, i) >= 0) TRY(p->dmalloc(&p->d_fullP[param_field(p->st, i)], (size_t)p->n2));
    for (int i = 0; i < NSS; ++i) if (state_field(p->st, i) >= 0) TRY(p->dmalloc(&p->d_fullS[state_field(p->st, i)], (size_t)p->n2));
    p->stage_planes = std::max<long>(1, std::min<long>(64, (256L << 20) / (p->n2 * 4)));
    TRY(p->dmalloc(&p->d_stage, (size_t)p->n2 * p->stage_planes));
    // gauges / cost
    TRY(p->dmalloc(&A.qg, (size_t)std::max(p->ngc, 1) * p->nt));
    TRY(p->upload_vec(&p->d_gauge_gid, p->gauge_gid.empty() ? std::vector<int>(1, 0) : p->gauge_gid));
    TRY(p->upload_vec(&p->d_gauge_flwacc, gfl));
    std::vector<float> area(std::max(p->ng, 1), 1.f);
    for (int g = 0; g < p->ng; ++g) area[g] = mesh->area[g];
    TRY(p->upload_vec(&p->d_area, area));
    TRY(p->dmalloc(&p->d_wgauge, (size_t)std::max(p->ng, 1)));
    TRY(p->dmalloc(&p->d_qobs, (size_t)std::max(p->ng, 1) * p->nt));
    TRY(p->dmalloc(&p->d_sums, (size_t)std::max(p->ng, 1)));
    TRY(p->dmalloc(&p->d_coef, (size_t)std::max(p->ng, 1) * SX_MAXJF));
    TRY(p->dmalloc(&p->d_med, (size_t)2 * std::max(p->ng, 1)));
    TRY(p->dmalloc(&p->d_med_idx, (size_t)2 * std::max(p->ng, 1)));
    TRY(p->dmalloc(&p->d_cost_out, 4));
    if (hipMemset(p->d_qobs, 0, (size_t)std::max(p->ng, 1) * p->nt * 4) != hipSuccess) { smashx_plan_destroy(p); return fail(SMASHX_E_HIP, "hipMemset"); }
    // default options: plain Model.run(): njf = 0 (mwd_setup.f90:236)
    std::memset(&p->opt, 0, sizeof(p->opt));
    p->opt.optimize_start_step = 1;
    p->wgauge.assign(std::max(p->ng, 1), p->ng > 0 ? 1.f / p->ng : 0.f);
    if (hipMemcpy(p->d_wgauge, p->wgauge.data(), p->wgauge.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { smashx_plan_destroy(p); return fail(SMASHX_E_HIP, "hipMemcpy"); }
    p->have_options = true;
#undef TRY
    *out = p;
    return 0;
}

int smashx_plan_destroy(smashx_plan* p) {
    if (!p) return 0;
    (void)set_device(p);
    if (p->xcomm) { auto& v = p->xcomm->plans; v.erase(std::remove(v.begin(), v.end(), p), v.end()); }
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->stream_r) (void)hipStreamSynchronize(p->stream_r);
    for (void* q : p->allocs) (void)hipFree(q);
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->stream_r) (void)hipStreamDestroy(p->stream_r);
    if (p->stream_j) { (void)hipStreamSynchronize(p->stream_j); (void)hipStreamDestroy(p->stream_j); }
    if (p->ev_j) (void)hipEventDestroy(p->ev_j);
    delete p;
    return 0;
}

int smashx_plan_ncells(const smashx_plan* p) { return p ? p->n : SMASHX_E_ARG; }

int smashx_plan_cell_order(const smashx_plan* p, int* rows, int* cols) {
    if (!p || !rows || !cols) return fail(SMASHX_E_ARG, "null argument");
    for (int k = 0; k < p->n; ++k) { rows[k] = p->sch.cell_flat[k] % p->cfg.nrow; cols[k] = p->sch.cell_flat[k] / p->cfg.nrow; }
    return 0;
}

static int alloc_forcing(smashx_plan* p) {
    if (p->d_prcp || p->d_prcp16) return 0;
    int rc;
    p->block_seen.clear();           // new rows: nothing is covered yet
    if (p->flay.compact) {
        p->ndays = (p->nt + p->flay.pet_hour0 + 23) / 24;
        if ((rc = p->dmalloc(&p->d_prcp16, (size_t)p->nt * p->npad))) return rc;
        if ((rc = p->dmalloc(&p->d_petd, (size_t)p->ndays * p->npad))) return rc;
        if ((rc = p->dmalloc(&p->d_ratio, 24))) return rc;
        if ((rc = p->dmalloc(&p->d_fstatus, 2))) return rc;
        HIPCHK(hipMemset(p->d_prcp16, 0, (size_t)p->nt * p->npad * sizeof(unsigned short)));
        HIPCHK(hipMemset(p->d_petd, 0xFF, (size_t)p->ndays * p->npad * sizeof(float)));      // NaN = "day not determined yet"
        HIPCHK(hipMemset(p->d_fstatus, 0, 2 * sizeof(unsigned)));
        HIPCHK(hipMemcpy(p->d_ratio, p->flay.pet_ratio, 24 * sizeof(float), hipMemcpyHostToDevice));
        p->A.prcp16 = p->d_prcp16; p->A.petd = p->d_petd; p->A.pet_ratio = p->d_ratio;
        p->A.prcp_c = p->flay.prcp_factor; p->A.prcp_gap = -99.f; p->A.hour0 = p->flay.pet_hour0;
        p->A.prcp = nullptr; p->A.pet = nullptr;
        return 0;
    }
    if ((rc = p->dmalloc(&p->d_prcp, (size_t)p->nt * p->npad))) return rc;
    if ((rc = p->dmalloc(&p->d_pet, (size_t)p->nt * p->npad))) return rc;
    p->A.prcp = p->d_prcp; p->A.pet = p->d_pet;
    p->A.prcp16 = nullptr; p->A.petd = nullptr; p->A.pet_ratio = nullptr;
    return 0;
}

static void drop_compact(smashx_plan* p) {
    p->dfree(p->d_prcp16); p->dfree(p->d_petd); p->dfree(p->d_ratio); p->dfree(p->d_fstatus);
    p->d_prcp16 = nullptr; p->d_petd = nullptr; p->d_ratio = nullptr; p->d_fstatus = nullptr;
    p->A.prcp16 = nullptr; p->A.petd = nullptr; p->A.pet_ratio = nullptr;
    p->flay.compact = 0;
}

// encode + verify rows [t0, t0 + rows) from fp32 device rows (row stride ld, cell k at idx[k] or k); *ok = every value is
// reproduced bit for bit by the kernels' decode
static int encode_block(smashx_plan* p, int t0, int rows, const float* d_prcp_src, const float* d_pet_src, const int* idx, long ld,
                        hipStream_t st, bool* ok) {
    const int h0 = p->flay.pet_hour0;
    const int d0 = (t0 + h0) / 24, d1 = (t0 + rows - 1 + h0) / 24;
    for (int r0 = 0; r0 < rows; r0 += 32768) {      // grid.y limit
        const int rr = std::min(32768, rows - r0);
        hipLaunchKernelGGL(k_encode_prcp, dim3((p->n + 255) / 256, rr), dim3(256), 0, st, p->d_prcp16 + (size_t)(t0 + r0) * p->npad,
                           d_prcp_src + (size_t)r0 * ld, idx, p->n, p->npad, ld, rr, p->flay.prcp_factor, p->d_fstatus);
    }
    hipLaunchKernelGGL(k_encode_pet, dim3((p->n + 255) / 256, d1 - d0 + 1), dim3(256), 0, st, p->d_petd, d_pet_src, idx, p->n, p->npad, ld,
                       t0, rows, h0, p->d_ratio, p->d_fstatus);
    unsigned status[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(status, p->d_fstatus, sizeof(status), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    *ok = status[0] == 0;
    if (status[1]) { float g; std::memcpy(&g, &status[1], 4); p->A.prcp_gap = g; }
    p->petd_open = true;
    return 0;
}

int smashx_set_forcing_layout(smashx_plan* p, const smashx_forcing_layout* lay) {
    if (!p || !lay) return fail(SMASHX_E_ARG, "null argument");
    if (p->d_prcp || p->d_prcp16) {
        if (p->have_forcing && !p->d_prcp16 == !lay->compact) return fail(SMASHX_E_STATE, "the forcing is already resident in this layout");
        int rc = set_device(p); if (rc) return rc;                     // reset: drop what is resident, the caller sends the forcing again
        HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipStreamSynchronize(p->stream_r));
        drop_compact(p);
        p->dfree(p->d_prcp); p->dfree(p->d_pet); p->d_prcp = p->d_pet = nullptr; p->A.prcp = p->A.pet = nullptr;
        p->have_forcing = false;
        p->block_seen.clear();
    }
    if (lay->compact) {
        if (p->cfg.dt != 3600.f) return fail(SMASHX_E_UNSUPPORTED, "compact forcing: the daily-PET form is defined for dt = 3600 s");
        if (!(lay->prcp_factor > 0.f) || lay->pet_hour0 < 0 || lay->pet_hour0 > 23) return fail(SMASHX_E_ARG, "bad prcp_factor / pet_hour0");
        for (int h = 0; h < 24; ++h) if (!(lay->pet_ratio[h] >= 0.f)) return fail(SMASHX_E_ARG, "pet_ratio must be >= 0");
    }
    p->flay = *lay;
    return 0;
}

int smashx_forcing_info(const smashx_plan* p, int* compact, double* bytes_per_cellstep) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    const bool c = p->d_prcp16 != nullptr || (!p->d_prcp && p->flay.compact);
    if (compact) *compact = c ? 1 : 0;
    if (bytes_per_cellstep) *bytes_per_cellstep = c ? 2.0 + 4.0 * ((p->nt + p->flay.pet_hour0 + 23) / 24) / (double)p->nt : 8.0;
    return 0;
}

int smashx_set_forcing(smashx_plan* p, const float* prcp, const float* pet, int sparse) {
    if (!p || !prcp || !pet) return fail(SMASHX_E_ARG, "null argument");
    int rc = set_device(p); if (rc) return rc;
    if (sparse && !p->d_sparse_idx) return fail(SMASHX_E_ARG, "sparse forcing needs mesh.path at plan creation");
    // a sparse vector is numbered over the whole grid's active cells: on a tiled plan it is longer than the part (the part's cells
    // keep their whole-grid positions in it)
    const long plane = sparse ? p->n_sparse : p->n2;
    const int* idx = sparse ? p->d_sparse_idx : p->d_cell_flat;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = alloc_forcing(p))) return rc;
        const bool compact = p->d_prcp16 != nullptr;
        bool ok = true;
        if (compact) {
            // staged in blocks of whole days, so that a day's PET is determined from all of its hours at once; both variables of a
            // block share the staging buffer
            const long half = std::max<long>(1, p->stage_planes / 2);
            HIPCHK(hipMemsetAsync(p->d_fstatus, 0, 2 * sizeof(unsigned), p->stream));      // a fresh data set: no gap value known yet
            HIPCHK(hipMemsetAsync(p->d_petd, 0xFF, (size_t)p->ndays * p->npad * sizeof(float), p->stream));
            for (int t = 0; t < p->nt && ok;) {
                int rows = (int)std::min<long>(half, p->nt - t);
                const int to_day_end = 24 - (t + p->flay.pet_hour0) % 24;
                if (rows > to_day_end) rows = to_day_end + (rows - to_day_end) / 24 * 24;
                float* sp = p->d_stage; float* se = p->d_stage + (size_t)half * p->n2;
                HIPCHK(hipMemcpyAsync(sp, prcp + (size_t)t * plane, (size_t)rows * plane * 4, hipMemcpyHostToDevice, p->stream));
                HIPCHK(hipMemcpyAsync(se, pet + (size_t)t * plane, (size_t)rows * plane * 4, hipMemcpyHostToDevice, p->stream));
                if ((rc = encode_block(p, t, rows, sp, se, idx, plane, p->stream, &ok))) return rc;
                t += rows;
            }
            if (ok) break;
            // not of the reader's form: never approximate -- keep the fp32 rows
            HIPCHK(hipStreamSynchronize(p->stream));
            drop_compact(p);
            continue;
        }
        const float* src[2] = {prcp, pet};
        float* dst[2] = {p->d_prcp, p->d_pet};
        for (int v = 0; v < 2; ++v)
            for (int t = 0; t < p->nt; t += (int)p->stage_planes) {
                const int rows = (int)std::min<long>(p->stage_planes, p->nt - t);
                HIPCHK(hipMemcpyAsync(p->d_stage, src[v] + (size_t)t * plane, (size_t)rows * plane * 4, hipMemcpyHostToDevice, p->stream));
                hipLaunchKernelGGL(k_gather_rows, dim3((p->n + 255) / 256, rows), dim3(256), 0, p->stream,
                                   dst[v] + (size_t)t * p->npad, p->d_stage, idx, p->n, p->npad, plane, rows);
                HIPCHK(hipStreamSynchronize(p->stream));
            }
        break;
    }
    p->have_forcing = true;
    if (getenv("SMASHX_VERBOSE"))
        fprintf(stderr, "smashx: forcing resident as %s (%.2f GB for %d cells x %d steps)\n", p->d_prcp16 ? "compact uint16 rain counts + daily PET" : "fp32 rows",
                (p->d_prcp16 ? 2.0 * p->nt + 4.0 * p->ndays : 8.0 * p->nt) * p->npad * 1e-9, p->n, p->nt);
    return 0;
}

// The forcing only counts as set once device blocks have covered every step of [0, nt): a sweep started earlier would close the
// daily PET of days it has not seen (k_close_petd pins them to 0) and later blocks for those days would then fail verification.
// Coverage is only forgotten where the data it stands for is: when the rows are (re)allocated (alloc_forcing, a new layout) and -- compact
// layout only -- when a block starting at step 0 opens a new upload cycle (its daily PET field is reset there, see below).  fp32 rows may
// be sent in any order and refreshed block by block.
static void mark_block(smashx_plan* p, int t0, int t1, bool new_cycle) {
    if (new_cycle) p->block_seen.assign(p->nt, 0);
    p->block_seen.resize(p->nt, 0);
    std::fill(p->block_seen.begin() + t0, p->block_seen.begin() + t1, 1);
    p->have_forcing = std::find(p->block_seen.begin(), p->block_seen.end(), 0) == p->block_seen.end();
}

int smashx_set_forcing_device_block(smashx_plan* p, int t0, int t1, const float* d_prcp, const float* d_pet) {
    if (!p || !d_prcp || !d_pet || t0 < 0 || t1 > p->nt || t0 >= t1) return fail(SMASHX_E_ARG, "bad block");
    int rc = set_device(p); if (rc) return rc;
    if ((rc = alloc_forcing(p))) return rc;
    if (p->d_prcp16) {
        bool ok = true;
        if (t0 == 0) {   // the forcing is being sent again from its start: forget the previous data set's gap value and daily field
            HIPCHK(hipMemsetAsync(p->d_fstatus, 0, 2 * sizeof(unsigned), p->stream));
            HIPCHK(hipMemsetAsync(p->d_petd, 0xFF, (size_t)p->ndays * p->npad * sizeof(float), p->stream));
        }
        if ((rc = encode_block(p, t0, t1 - t0, d_prcp, d_pet, nullptr, (long)p->n, p->stream, &ok))) return rc;
        if (!ok) {
            HIPCHK(hipMemset(p->d_fstatus, 0, sizeof(unsigned)));
            return fail(SMASHX_E_UNSUPPORTED, "forcing block [" + std::to_string(t0) + ", " + std::to_string(t1) + ") is not of the compact form "
                        "(prcp = k * factor with k < 65535 or one gap value; pet = daily * ratio(hour)): reset the layout to fp32 and send the forcing again");
        }
        mark_block(p, t0, t1, t0 == 0);
        return 0;
    }
    HIPCHK(hipMemcpy2DAsync(p->d_prcp + (size_t)t0 * p->npad, (size_t)p->npad * 4, d_prcp, (size_t)p->n * 4, (size_t)p->n * 4, t1 - t0, hipMemcpyDeviceToDevice, p->stream));
    HIPCHK(hipMemcpy2DAsync(p->d_pet + (size_t)t0 * p->npad, (size_t)p->npad * 4, d_pet, (size_t)p->n * 4, (size_t)p->n * 4, t1 - t0, hipMemcpyDeviceToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    mark_block(p, t0, t1, false);
    return 0;
}

int smashx_set_qobs(smashx_plan* p, const float* qobs) {
    if (!p || (!qobs && p->ng > 0)) return fail(SMASHX_E_ARG, "null argument");
    int rc = set_device(p); if (rc) return rc;
    std::vector<float> tr((size_t)std::max(p->ng, 1) * p->nt);
    for (int g = 0; g < p->ng; ++g)
        for (int t = 0; t < p->nt; ++t) tr[(size_t)g * p->nt + t] = qobs[g + (size_t)p->ng * t];   // (ng,nt) column-major -> [g][t]
    HIPCHK(hipMemcpy(p->d_qobs, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
    p->have_qobs = true;
    return 0;
}

int smashx_set_options(smashx_plan* p, const smashx_options* o) {
    if (!p || !o) return fail(SMASHX_E_ARG, "null argument");
    int rc = set_device(p); if (rc) return rc;
    if (o->njf < 0 || o->njf > SX_MAXJF || o->njr < 0 || o->njr > 4) return fail(SMASHX_E_ARG, "njf/njr out of range");
    if (o->optimize_start_step < 1 || o->optimize_start_step > p->nt) return fail(SMASHX_E_ARG, "optimize_start_step out of range");
    for (int j = 0; j < o->njf; ++j)
        if (o->jobs_fun[j] < SMASHX_NSE || o->jobs_fun[j] > SMASHX_LOGARITHMIC)
            return fail(SMASHX_E_UNSUPPORTED, "signature-based jobs_fun are outside the hot path (SURVEY.md 8a a9)");
    for (int j = 0; j < o->njr; ++j)
        if (o->jreg_fun[j] < SMASHX_PRIOR || o->jreg_fun[j] > SMASHX_HARD_SMOOTHING)
            return fail(SMASHX_E_UNSUPPORTED, "unknown jreg_fun (prior / smoothing / hard_smoothing, mwd_cost.f90:199-224)");
    p->opt = *o;
    p->jr_ready = false;
    p->opt.wgauge = nullptr;
    for (int g = 0; g < p->ng; ++g) {
        const float w = o->wgauge ? o->wgauge[g] : 1.f / p->ng;
        if (w < 0.f && p->tiled && p->med_nslots == 0)
            return fail(SMASHX_E_UNSUPPORTED, "negative wgauge (median over gauges) on a tiled plan: declare the slots of the whole decomposition first (smashx_set_median_slots)");
        if (p->med_nslots > 0 && (w < 0.f) != (p->med_slot_h[g] >= 0))
            return fail(SMASHX_E_ARG, "smashx_set_median_slots and wgauge disagree: exactly the negative-weight gauges have a slot");
        p->wgauge[g] = w;
    }
    HIPCHK(hipMemcpy(p->d_wgauge, p->wgauge.data(), p->wgauge.size() * 4, hipMemcpyHostToDevice));
    p->have_options = true;
    return 0;
}

namespace {
int jreg_optim(const smashx_plan* p, int idx) {
    return idx < SMASHX_GNP ? p->opt.optim_parameters[idx] : p->opt.optim_states[idx - SMASHX_GNP];
}
float jreg_span(const smashx_plan* p, int idx) {
    return idx < SMASHX_GNP ? p->opt.ub_parameters[idx] - p->opt.lb_parameters[idx]
                            : p->opt.ub_states[idx - SMASHX_GNP] - p->opt.lb_states[idx - SMASHX_GNP];
}
// compute_jreg (and COMPUTE_JREG_B when adjoint) on the regularisation stream: independent of the simulation
int run_jreg(smashx_plan* p, int adjoint, float cost_b) {
    if (!p->jr_ready) return fail(SMASHX_E_STATE, "jreg is on but the current upload carried no background fields");
    hipStream_t sJ = p->stream_j;
    const dim3 b(256), gfull((unsigned)((p->n2 + 255) / 256));
    const int nrow = p->cfg.nrow, ncol = p->cfg.ncol;
    SxJregChains& ch = p->jchains;
    ch.nchain = 2 * p->opt.njr;
    long planes = 0;
    for (int i = 0; i < p->opt.njr; ++i)
        for (int grp = 0; grp < 2; ++grp) {
            const int c = 2 * i + grp, lo = grp ? SMASHX_GNP : 0, hi = grp ? SMASHX_GNP + SMASHX_GNS : SMASHX_GNP;
            ch.first[c] = planes; ch.nplane[c] = 0;
            for (int idx = lo; idx < hi; ++idx) if (jreg_optim(p, idx) > 0) { ch.nplane[c]++; planes++; }
        }
    if ((size_t)planes > p->jterm_planes) {
        int rc = p->dmalloc(&p->d_jterm, (size_t)planes * p->n2); if (rc) return rc;   // (the smaller buffer stays owned by the plan)
        p->jterm_planes = (size_t)planes;
    }
    for (int i = 0; i < p->opt.njr; ++i)
        for (int grp = 0; grp < 2; ++grp) {
            const int c = 2 * i + grp, lo = grp ? SMASHX_GNP : 0, hi = grp ? SMASHX_GNP + SMASHX_GNS : SMASHX_GNP;
            long slot = ch.first[c];
            for (int idx = lo; idx < hi; ++idx) {
                if (jreg_optim(p, idx) <= 0) continue;
                float* t = p->d_jterm + (size_t)slot++ * p->n2;
                if (p->opt.jreg_fun[i] == SMASHX_PRIOR)
                    hipLaunchKernelGGL(sx_k_prior_terms, gfull, b, 0, sJ, t, p->d_jx[idx], p->d_jb[idx], p->n2);
                else
                    hipLaunchKernelGGL(sx_k_smooth_terms, gfull, b, 0, sJ, t, p->d_jx[idx], p->d_jb[idx],
                                       p->opt.jreg_fun[i] == SMASHX_SMOOTHING ? 1 : 0, p->d_active, nrow, ncol);
            }
        }
    hipLaunchKernelGGL(sx_k_seq_sum, dim3(ch.nchain), dim3(64), 0, sJ, p->d_jterm, ch, p->n2, p->d_jsum);
    if (adjoint) {
        const float jreg_b = p->opt.wjreg * cost_b;                       // COMPUTE_COST_B: cost = jobs + wjreg * jreg
        for (int idx = 0; idx < SMASHX_GNP + SMASHX_GNS; ++idx)
            if (jreg_optim(p, idx) > 0) HIPCHK(hipMemsetAsync(p->d_jg[idx], 0, (size_t)p->n2 * 4, sJ));
        for (int i = p->opt.njr - 1; i >= 0; --i) {
            const float w = p->opt.wjreg_fun[i];
            const bool prior = p->opt.jreg_fun[i] == SMASHX_PRIOR;
            const float res_b = prior ? w * jreg_b : (w * w) * jreg_b;    // wjreg_fun(i)**2 for the smoothing terms
            for (int idx = SMASHX_GNP + SMASHX_GNS - 1; idx >= 0; --idx) {
                if (jreg_optim(p, idx) <= 0) continue;
                if (prior) hipLaunchKernelGGL(sx_k_prior_b, gfull, b, 0, sJ, p->d_jg[idx], p->d_jx[idx], p->d_jb[idx], res_b, p->n2);
                else hipLaunchKernelGGL(sx_k_smooth_b, gfull, b, 0, sJ, p->d_jg[idx], p->d_jx[idx], p->d_jb[idx],
                                        p->opt.jreg_fun[i] == SMASHX_SMOOTHING ? 1 : 0, p->d_active, nrow, ncol, res_b);
            }
        }
        if (p->opt.denormalize_forward)
            for (int idx = 0; idx < SMASHX_GNP + SMASHX_GNS; ++idx)
                if (jreg_optim(p, idx) > 0) hipLaunchKernelGGL(sx_k_plane_div, gfull, b, 0, sJ, p->d_jg[idx], jreg_span(p, idx), p->n2);
    }
    HIPCHK(hipEventRecord(p->ev_j, sJ));
    return 0;
}
}  // namespace

int smashx_upload(smashx_plan* p, const smashx_parameters* params, const smashx_parameters* params_bgd,
                  const smashx_states* states, const smashx_states* states_bgd) {
    if (!p || !params || !states) return fail(SMASHX_E_ARG, "null argument");
    int rc = set_device(p); if (rc) return rc;
    const int st = p->st;
    const dim3 b(256), gfull((unsigned)((p->n2 + 255) / 256)), gk((p->n + 255) / 256);
    float* pdst[NPS] = {p->A.ci, p->A.cp, p->A.cft, p->A.cst, p->A.exc, p->A.lr, p->A.px[0], p->A.px[1], p->A.px[2]};
    for (int i = 0; i < NPS; ++i) {
        const int f = param_field(st, i);
        if (f < 0) continue;
        if (!params->f[f]) {     // NULL = unchanged since the last upload (calibration loops move only the optimised fields)
            if (!p->uploaded) return fail(SMASHX_E_ARG, "a parameter field the structure uses is NULL");
            continue;
        }
        HIPCHK(hipMemcpyAsync(p->d_fullP[f], params->f[f], (size_t)p->n2 * 4, hipMemcpyHostToDevice, p->stream));
        if (p->opt.denormalize_forward)
            hipLaunchKernelGGL(k_denormalize, gfull, b, 0, p->stream, p->d_fullP[f], p->n2, p->opt.lb_parameters[f], p->opt.ub_parameters[f]);
        hipLaunchKernelGGL(k_gather, gk, b, 0, p->stream, pdst[i], p->d_fullP[f], p->d_cell_flat, p->n);
    }
    for (int i = 0; i < NSS; ++i) {
        const int f = state_field(st, i);
        if (f < 0) continue;
        if (!states->f[f]) {
            if (!p->uploaded) return fail(SMASHX_E_ARG, "a state field the structure uses is NULL");
            continue;
        }
        HIPCHK(hipMemcpyAsync(p->d_fullS[f], states->f[f], (size_t)p->n2 * 4, hipMemcpyHostToDevice, p->stream));
        if (p->opt.denormalize_forward)
            hipLaunchKernelGGL(k_denormalize, gfull, b, 0, p->stream, p->d_fullS[f], p->n2, p->opt.lb_states[f], p->opt.ub_states[f]);
        hipLaunchKernelGGL(k_gather, gk, b, 0, p->stream, p->st0[i], p->d_fullS[f], p->d_cell_flat, p->n);
    }
    hipLaunchKernelGGL(sx_k_prep_routing, gk, b, 0, p->stream, p->A);
    p->jr_ready = false;
    if (p->opt.njr > 0) {
        // compute_jreg reads the control vector as compute_cost holds it: normalised again after the forward run
        // when denormalize_forward is set (mwd_cost.f90:284-291), plus the background fields
        if (!params_bgd || !states_bgd) return fail(SMASHX_E_ARG, "jreg needs parameters_bgd and states_bgd");
        for (int idx = 0; idx < SMASHX_GNP + SMASHX_GNS; ++idx) {
            if (jreg_optim(p, idx) <= 0) continue;
            const bool isp = idx < SMASHX_GNP;
            const float* h = isp ? params->f[idx] : states->f[idx - SMASHX_GNP];
            const float* hb = isp ? params_bgd->f[idx] : states_bgd->f[idx - SMASHX_GNP];
            if (!h || !hb) return fail(SMASHX_E_ARG, "an optimised field (optim_parameters / optim_states) or its background is NULL");
            if (!p->d_jx[idx]) {
                if ((rc = p->dmalloc(&p->d_jx[idx], (size_t)p->n2))) return rc;
                if ((rc = p->dmalloc(&p->d_jb[idx], (size_t)p->n2))) return rc;
                if ((rc = p->dmalloc(&p->d_jg[idx], (size_t)p->n2))) return rc;
            }
            const float lb = isp ? p->opt.lb_parameters[idx] : p->opt.lb_states[idx - SMASHX_GNP];
            const float ub = isp ? p->opt.ub_parameters[idx] : p->opt.ub_states[idx - SMASHX_GNP];
            HIPCHK(hipMemcpyAsync(p->d_jx[idx], h, (size_t)p->n2 * 4, hipMemcpyHostToDevice, p->stream));
            if (p->opt.denormalize_forward) {
                hipLaunchKernelGGL(k_denormalize, gfull, b, 0, p->stream, p->d_jx[idx], p->n2, lb, ub);
                hipLaunchKernelGGL(k_normalize, gfull, b, 0, p->stream, p->d_jx[idx], p->n2, lb, ub);
            }
            HIPCHK(hipMemcpyAsync(p->d_jb[idx], hb, (size_t)p->n2 * 4, hipMemcpyHostToDevice, p->stream));
        }
        p->jr_ready = true;
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    p->uploaded = true;
    return 0;
}

static int sweep_once(smashx_plan* p, int adjoint, float cost_b, bool* stalled_out);
static int close_forcing(smashx_plan* p) {       // compact PET: days that never showed a daytime hour take the value 0
    if (!p->petd_open || !p->d_petd) return 0;
    const size_t n = (size_t)p->ndays * p->npad;
    hipLaunchKernelGGL(k_close_petd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, p->d_petd, n);
    p->petd_open = false;
    return 0;
}

// A chained routing launch draws its groups by ticket (sx_kernels.h), so a group only waits for groups that are resident or done: every
// input from outside the launch (vertical kernel, round 0, series received from other ranks) is complete before the launch starts.
// Should a wait nevertheless exhaust its poll limit, a flag is raised and the sweep's results are void: the plan then drops to one launch per
// round for good and the sweep is run again; the ranks of a decomposition take that decision together (one all-reduce per sweep).
int smashx_sweep(smashx_plan* p, int adjoint, float cost_b) {
    bool stalled = false;
    int rc = sweep_once(p, adjoint, cost_b, &stalled);
    if (rc) return rc;
    const bool agree = p->xcomm != nullptr && p->chain;      // a decomposition decides together: same condition on every rank
    if (agree) {
        double v = stalled ? 1.0 : 0.0;
        if ((rc = smashx_comm_allreduce_sum(p->xcomm, &v, 1))) return rc;
        stalled = v > 0.0;
    }
    if (!stalled) return 0;
    if (p->halo_fn && !p->xcomm)     // host-callback exchange: the neighbours of a tile cannot be made to repeat their sweep from here
        return fail(SMASHX_E_HIP, "chained routing launch stalled waiting for an upstream group (results invalid); set SMASHX_CHAIN_ROUNDS=0");
    fprintf(stderr, "smashx: a chained routing launch stalled (poll limit reached); this plan now runs one launch per routing round%s\n",
            p->xcomm ? " (every rank of the decomposition does)" : "");
    p->chain = false;
    p->A.mute_group = -1;
    rc = sweep_once(p, adjoint, cost_b, &stalled);
    if (rc) return rc;
    return stalled ? fail(SMASHX_E_HIP, "routing stalled without chained rounds (internal error)") : 0;
}

// ---- boundary-series exchange of a tiled plan (used by the sweep and by the tangent model) ---------------------------------
// move the boundary series of one sub-chunk between the exchange rows (xarr: xT, or xdT for the tangents) and the message buffers
// (the caller's, or the plan's own in peer-grouped edge order when RCCL carries them)
static void sx_halo_move(smashx_plan* p, bool native, float* xarr, bool pack, bool out_edges, int off, int T, hipStream_t st) {
    const int nedge = out_edges ? p->n_out : p->n_in;
    if (nedge == 0) return;
    const int Tq = (T + 3) / 4;
    float4* x4 = reinterpret_cast<float4*>(xarr) + (size_t)(off / 4) * p->A.nx;
    float4* buf = reinterpret_cast<float4*>(native ? (out_edges ? p->x_out : p->x_in) : (out_edges ? p->halo_out : p->halo_in));
    const int* slots = native ? (out_edges ? p->d_out_xp : p->d_in_xp) : (out_edges ? p->d_out_x : p->d_in_x);
    const dim3 g((nedge * Tq + 255) / 256), b(256);
    if (pack) hipLaunchKernelGGL(k_halo_pack, g, b, 0, st, buf, x4, slots, nedge, p->A.nx, Tq);
    else hipLaunchKernelGGL(k_halo_unpack, g, b, 0, st, x4, buf, slots, nedge, p->A.nx, Tq);
}
// one grouped send or recv per sub-chunk: a message per peer = its edges x the sub-chunk's steps, stream-ordered on st
static int sx_halo_xfer(smashx_plan* p, bool send, bool out_edges, int T, hipStream_t st) {
    const std::vector<PeerSeg>& segs = out_edges ? p->out_segs : p->in_segs;
    if (segs.empty()) return 0;
    RcclApi& R = rccl();
    float* buf = out_edges ? p->x_out : p->x_in;
    const size_t per_edge = (size_t)((T + 3) / 4) * 4;
    static const bool trace = getenv("SMASHX_TRACE_XFER") != nullptr;     // debugging aid: one line per posted group, completion awaited
    if (trace) {
        std::string peers;
        for (const PeerSeg& sg : segs) peers += " " + std::to_string(sg.rank) + ":" + std::to_string(sg.count);
        fprintf(stderr, "[smashx rank %d] %s %s edges, %d steps, peers(rank:edges)%s\n", p->xcomm->rank, send ? "send" : "recv",
                out_edges ? "out" : "in", T, peers.c_str());
    }
    NCCLCHK(R.GroupStart());
    for (const PeerSeg& sg : segs) {
        float* b = buf + (size_t)sg.first * per_edge;
        const ncclResult_t r = send ? R.Send(b, (size_t)sg.count * per_edge, ncclFloat, sg.rank, p->xcomm->comm, st)
                                    : R.Recv(b, (size_t)sg.count * per_edge, ncclFloat, sg.rank, p->xcomm->comm, st);
        if (r != ncclSuccess) { (void)R.GroupEnd(); return fail(SMASHX_E_HIP, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(r)); }
    }
    NCCLCHK(R.GroupEnd());
    if (trace) {
        HIPCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[smashx rank %d]   ... completed\n", p->xcomm->rank);
    }
    return 0;
}
// phases as in smashx_halo_fn: 0 FWD_RECV (in), 1 FWD_SEND (out), 2 ADJ_RECV (out), 3 ADJ_SEND (in)
static int sx_halo_hook(smashx_plan* p, bool native, int phase, int t0, int T, hipStream_t st) {
    if (native) return sx_halo_xfer(p, phase == 1 || phase == 3, phase == 1 || phase == 2, T, st);
    // host callback: the packed buffer must be complete before the host moves it; the unpack of the previous message must have
    // consumed the receive buffer before the host fills it again
    HIPCHK(hipStreamSynchronize(st));
    const int rc2 = p->halo_fn(p->halo_user, phase, t0, T);
    return rc2 ? fail(SMASHX_E_ARG, "halo callback failed") : 0;
}

static int sweep_once(smashx_plan* p, int adjoint, float cost_b, bool* stalled_out) {
    *stalled_out = false;
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    if (!p->have_forcing) return fail(SMASHX_E_STATE, "forcing not set");
    if (!p->uploaded) return fail(SMASHX_E_STATE, "parameters/states not uploaded");
    int rc = set_device(p); if (rc) return rc;
    if ((rc = ensure_chunk_buffers(p, adjoint != 0))) return rc;
    if ((p->n_out > 0 || p->n_in > 0) && !p->halo_fn && !p->xcomm)
        return fail(SMASHX_E_STATE, "tile has boundary series but no exchange is set (smashx_set_exchange / smashx_set_halo)");
    p->launches.clear(); p->pool_used = 0;
    p->buf_free.clear();             // (the previous sweep ended with both streams drained)
    hipStream_t sV = p->stream, sR = p->stream_r;
    if ((rc = close_forcing(p))) return rc;
    HIPCHK(hipEventRecord(p->ev0, sV));
    HIPCHK(hipStreamWaitEvent(sR, p->ev0, 0));
    p->chain_used = false;
    HIPCHK(hipMemsetAsync(p->A.prog + p->sch.ngroups, 0, sizeof(int), sR));   // stall flag of the chained launches
    HIPCHK(hipMemsetAsync(p->A.prog + p->sch.ngroups + 3, 0, 4 * sizeof(int), sR));   // ... and its diagnostics
    if (p->opt.njr > 0) {
        HIPCHK(hipStreamWaitEvent(p->stream_j, p->ev0, 0));
        if ((rc = run_jreg(p, adjoint, cost_b))) return rc;
    }
    if ((rc = restore_states(p, p->st0))) return rc;
    const int C = p->nchunks;
    auto nsub_of = [&](int T) { return (T + p->Tp - 1) / p->Tp; };
    // forward over one storage chunk: V(j) on the V stream, R(j) on the R stream as soon as V(j) is done
    const bool native = p->xcomm != nullptr;
    const bool halo = (p->halo_fn || native) && (p->n_out > 0 || p->n_in > 0);
    auto halo_move = [&](bool pack, bool out_edges, int off, int T, hipStream_t st) { sx_halo_move(p, native, p->A.xT, pack, out_edges, off, T, st); };
    auto hook = [&](int phase, int t0, int T, hipStream_t st) -> int { return sx_halo_hook(p, native, phase, t0, T, st); };
    // Recomputation of a storage chunk (reverse sweep) needs no neighbour: the inlet series a rank received for that chunk in the first
    // pass are kept (n_in edges x Tc steps x 4 B per chunk) and unpacked again, and nothing is sent -- the ranks downstream kept theirs.
    // A sweep then changes direction twice between the ranks (forward -> reverse) instead of twice per recomputed chunk more.
    const int ns_max = nsub_of(p->Tc);
    const size_t keep_sub = (size_t)std::max(p->n_in, 1) * (size_t)((p->Tp + 3) / 4) * 4;      // floats of one sub-chunk's message
    const bool keep_in = adjoint && halo && p->n_in > 0 && C > 1;
    if (keep_in && p->x_keep_cap < keep_sub * ns_max * (size_t)(C - 1)) {
        if (p->x_keep) p->dfree(p->x_keep);
        p->x_keep = nullptr; p->x_keep_cap = 0;
        if ((rc = p->dmalloc(&p->x_keep, keep_sub * ns_max * (size_t)(C - 1)))) return rc;
        p->x_keep_cap = keep_sub * ns_max * (size_t)(C - 1);
    }
    // the series of the in edges for sub-chunk jb of chunk c reach the exchange rows: received (and kept), or taken from the first pass
    auto inlet_series = [&](int c, int jb, int off, int t0, int T, bool recompute, hipStream_t st) -> int {
        float* msg = native ? p->x_in : p->halo_in;
        float* kept = (keep_in && c < C - 1) ? p->x_keep + ((size_t)c * ns_max + jb) * keep_sub : nullptr;
        const size_t bytes = (size_t)p->n_in * (size_t)((T + 3) / 4) * 4 * sizeof(float);
        if (recompute && kept) {
            HIPCHK(hipMemcpyAsync(msg, kept, bytes, hipMemcpyDeviceToDevice, st));
        } else {
            if ((rc = hook(0, t0, T, st))) return rc;      // the message buffer now holds the upstream tiles' series
            if (kept) HIPCHK(hipMemcpyAsync(kept, msg, bytes, hipMemcpyDeviceToDevice, st));
        }
        halo_move(false, false, off, T, st);
        return 0;
    };
    static const bool early_release = []() { const char* e = getenv("SMASHX_EARLY_RELEASE"); return !(e && e[0] == '0'); }();
    auto forward_chunk = [&](int c, bool tape, bool recompute = false) -> int {
        const int t0c = c * p->Tc, Tcur = chunk_len(p, c), ns = nsub_of(Tcur);
        std::vector<hipEvent_t> ev(ns);
        auto launch_v = [&](int jb) -> int {
            const int off = jb * p->Tp, T = std::min(p->Tp, Tcur - off);
            // this part of the chunk buffers was last used by the R stream in the previous pass over them (the reverse sweep ends on
            // the V stream): wait for that sub-chunk only, so that the routing of chunk c's tail runs under chunk c + 1's first kernels
            if (jb < (int)p->buf_free.size() && p->buf_free[jb]) { HIPCHK(hipStreamWaitEvent(sV, p->buf_free[jb], 0)); p->buf_free[jb] = nullptr; }
            vert_fwd(p, off, tape, t0c + off, T);
            ev[jb] = p->event(); HIPCHK(hipEventRecord(ev[jb], sV));
            return 0;
        };
        // every vertical sub-chunk is queued at once: they only depend on each other (state carry, stream order) and write
        // disjoint parts of the chunk buffers, so the V stream never waits for the host, which blocks in the halo hooks
        // of the routing pipeline below (tiles) -- the vertical work then hides the pipeline fill across tiles
        for (int jb = 0; jb < ns; ++jb) if ((rc = launch_v(jb))) return rc;
        for (int jb = 0; jb < ns; ++jb) {
            const int off = jb * p->Tp, T = std::min(p->Tp, Tcur - off);
            if (halo && p->n_in > 0 && (rc = inlet_series(c, jb, off, t0c + off, T, recompute, sR))) return rc;
            HIPCHK(hipStreamWaitEvent(sR, ev[jb], 0));
            // The next pass over this part of the chunk buffers is a vertical kernel that overwrites qt (and, taped, the level tapes --
            // which no forward routing launch reads): it may start as soon as the routing has READ qt.  With the chained launch on staging
            // rows that is after round 0 and the copy pass -- the chained launch, a latency chain on a few CUs, then runs under the next
            // storage chunk's vertical kernel (first pass of a checkpointed sweep: 3 x 3.6 ms at 2048^2); else after the whole pass.
            if ((int)p->buf_free.size() <= jb) p->buf_free.resize(jb + 1, nullptr);
            p->buf_free[jb] = p->event();
            const bool released = route_fwd(p, off, tape, t0c + off, T, early_release ? p->buf_free[jb] : nullptr);
            if (halo && p->n_out > 0 && !recompute) {       // (a recomputed chunk sends nothing: every rank kept what it received)
                halo_move(true, true, off, T, sR);
                if ((rc = hook(1, t0c + off, T, sR))) return rc;
            }
            if (!released) HIPCHK(hipEventRecord(p->buf_free[jb], sR));
        }
        return 0;
    };
    // (before a storage chunk's buffers are overwritten the V stream waits for the R stream sub-chunk by sub-chunk: buf_free in forward_chunk)
    p->dom_q_active = !adjoint && p->h_qsim_domain && p->A.qdT;
    // optional whole-domain stores (md_forward_structure.f90:158-194) of one storage chunk -> the caller's arrays
    auto export_domain = [&](int c) -> int {
        const int t0c = c * p->Tc, Tcur = chunk_len(p, c);
        HIPCHK(hipStreamSynchronize(sV));
        HIPCHK(hipStreamSynchronize(sR));
        const long plane = p->dom_sparse ? p->n_sparse : p->n2;
        const int* idx = p->dom_sparse ? p->d_sparse_idx : p->d_cell_flat;
        const int nbmax = (int)std::max<long>(1, std::min<long>(p->stage_planes * p->n2 / plane, 1 << 15));
        for (int which = 0; which < 2; ++which) {
            float* host = which ? p->h_net_prcp_domain : p->h_qsim_domain;
            const float* src = which ? p->A.qtT : p->A.qdT;
            if (!host || !src) continue;
            for (int tl0 = 0; tl0 < Tcur; tl0 += nbmax) {
                const int nb = std::min(nbmax, Tcur - tl0);
                if (!p->dom_sparse || p->tiled)      // cells this plan does not write: inactive ones (dense form), other parts' (tiles)
                    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)nb * plane + 255) / 256)), dim3(256), 0, sV, p->d_stage, -99.f, (size_t)nb * plane);
                hipLaunchKernelGGL(k_domain_export, dim3((p->n + 255) / 256, nb), dim3(256), 0, sV, p->d_stage, src, idx, p->n, p->npad, plane, tl0, nb);
                HIPCHK(hipMemcpyAsync(host + (size_t)(t0c + tl0) * plane, p->d_stage, (size_t)nb * plane * 4, hipMemcpyDeviceToHost, sV));
                HIPCHK(hipStreamSynchronize(sV));
            }
        }
        return 0;
    };
    if (!adjoint) {
        const bool dom = p->h_qsim_domain || p->h_net_prcp_domain;
        for (int c = 0; c < C; ++c) {
            if ((rc = forward_chunk(c, false))) return rc;
            if (dom && (rc = export_domain(c))) return rc;
        }
        if ((rc = run_cost(p, 0, 0.f))) return rc;
    } else {
        for (int c = 0; c < C; ++c) {
            if (C > 1) {
                float* cur[5] = {p->A.hi, p->A.hp, p->A.hft, p->A.hst, p->A.hlr};
                float* dst[5];
                for (int i = 0; i < 5; ++i) dst[i] = p->ckpt + ((size_t)c * 5 + i) * p->npad;
                if ((rc = copy_states(p, dst, cur))) return rc;
            }
            // the last storage chunk is the first one the reverse sweep needs: it runs with the tape on straight away and
            // is not recomputed
            if ((rc = forward_chunk(c, c == C - 1))) return rc;
        }
        if ((rc = run_cost(p, 1, cost_b))) return rc;
        {   // gradient accumulators start from what COMPUTE_COST_B left in parameters_b / states_b: zero, or the
            // regulariser's gradient for the optimised fields (forward_db.f90:10869-10874)
            float* gv[14] = {p->A.ci_b, p->A.cp_b, p->A.cft_b, p->A.cst_b, p->A.exc_b, p->A.hi_b, p->A.hp_b, p->A.hft_b, p->A.hst_b,
                             p->A.lr_b, p->A.hlr_b, p->A.px_b[0], p->A.px_b[1], p->A.px_b[2]};
            const int ps = p->st;
            auto pfi = [&](int slot) { const int f = param_field(ps, slot); return f; };
            auto sfi = [&](int slot) { const int f = state_field(ps, slot); return f < 0 ? -1 : SMASHX_GNP + f; };
            const int gi[14] = {pfi(0), pfi(1), pfi(2), pfi(3), pfi(4), sfi(0), sfi(1), sfi(2), sfi(3), pfi(5), sfi(4), pfi(6), pfi(7), pfi(8)};
            bool waited[2] = {false, false};
            for (int q = 0; q < 14; ++q) {
                const bool on_r = (q == 9 || q == 10);
                hipStream_t sq = on_r ? sR : sV;
                HIPCHK(hipMemsetAsync(gv[q], 0, (size_t)p->npad * 4, sq));
                if (p->opt.njr > 0 && gi[q] >= 0 && jreg_optim(p, gi[q]) > 0) {
                    if (!waited[on_r ? 1 : 0]) { HIPCHK(hipStreamWaitEvent(sq, p->ev_j, 0)); waited[on_r ? 1 : 0] = true; }
                    hipLaunchKernelGGL(k_gather, dim3((p->n + 255) / 256), dim3(256), 0, sq, gv[q], p->d_jg[gi[q]], p->d_cell_flat, p->n);
                }
            }
        }
        if (p->ng == 0) HIPCHK(hipMemsetAsync(p->A.qgb, 0, (size_t)std::max(p->ngc, 1) * p->nt * 4, sR));
        for (int c = C - 1; c >= 0; --c) {
            const int t0c = c * p->Tc, Tcur = chunk_len(p, c);
            if (C > 1 && c < C - 1) {   // recompute this storage chunk with the tape on
                float* src[5];
                for (int i = 0; i < 5; ++i) src[i] = p->ckpt + ((size_t)c * 5 + i) * p->npad;
                if ((rc = restore_states(p, src))) return rc;
                if ((rc = forward_chunk(c, true, true))) return rc;
            }
            const int nsa = (Tcur + p->Tp - 1) / p->Tp;
            for (int jb = nsa - 1; jb >= 0; --jb) {
                const int off = jb * p->Tp, T = std::min(p->Tp, Tcur - off);
                if (halo && p->n_out > 0) {
                    if ((rc = hook(2, t0c + off, T, sR))) return rc;  // out_buf now holds the downstream tiles' adjoint contributions
                    halo_move(false, true, off, T, sR);
                }
                route_adj(p, off, t0c + off, T);
                hipEvent_t e = p->event();
                HIPCHK(hipEventRecord(e, sR));
                HIPCHK(hipStreamWaitEvent(sV, e, 0));
                vert_adj(p, off, t0c + off, T);
                if (halo && p->n_in > 0) {
                    halo_move(true, false, off, T, sR);
                    if ((rc = hook(3, t0c + off, T, sR))) return rc;
                }
            }
        }
    }
    {
        hipEvent_t e = p->event();
        HIPCHK(hipEventRecord(e, sR));
        HIPCHK(hipStreamWaitEvent(sV, e, 0));
    }
    HIPCHK(hipEventRecord(p->ev1, sV));
    HIPCHK(hipStreamSynchronize(sV));
    HIPCHK(hipStreamSynchronize(sR));
    if (p->opt.njr > 0) HIPCHK(hipStreamSynchronize(p->stream_j));
    HIPCHK(hipGetLastError());
    if (p->chain_used) {
        int stalled = 0;
        HIPCHK(hipMemcpy(&stalled, p->A.prog + p->sch.ngroups, sizeof(int), hipMemcpyDeviceToHost));
        if (stalled) {
            if (getenv("SMASHX_VERBOSE")) {
                int d[7] = {0};
                (void)hipMemcpy(d, p->A.prog + p->sch.ngroups, sizeof(d), hipMemcpyDeviceToHost);
                fprintf(stderr, "smashx: stall: group %d followed the progress of group %d, needed %d blocks, saw %d; tickets drawn %d, chained groups %d..%d, progress:",
                        d[3] - 2, p->sch.ngroups + d[4], d[5], d[6], d[1], p->sch.round_group_begin[chain_first(p)], p->sch.ngroups - 1);
                std::vector<int> pr(p->sch.ngroups);
                (void)hipMemcpy(pr.data(), p->A.prog, pr.size() * sizeof(int), hipMemcpyDeviceToHost);
                for (int g = p->sch.round_group_begin[chain_first(p)]; g < p->sch.ngroups; ++g) fprintf(stderr, " %d", pr[g]);
                fprintf(stderr, " (n_in %d n_out %d)\n", p->n_in, p->n_out);
            }
            *stalled_out = true; return 0;
        }
    }
    // timing
    smashx_timing& tm = p->timing;
    std::memset(&tm, 0, sizeof(tm));
    (void)hipEventElapsedTime(&tm.sweep_ms, p->ev0, p->ev1);
    for (const Launch& l : p->launches) {
        float ms = 0.f;
        if (l.a && l.b) (void)hipEventElapsedTime(&ms, l.a, l.b);
        switch (l.kind) {
            case 0: tm.vert_fwd_ms += ms; tm.vert_fwd_launches++; tm.cellsteps[0] += l.cellsteps; break;
            case 1: tm.route_fwd_ms += ms; tm.route_fwd_launches++; tm.cellsteps[1] += l.cellsteps; break;
            case 2: tm.route_adj_ms += ms; tm.route_adj_launches++; tm.cellsteps[2] += l.cellsteps; break;
            case 3: tm.vert_adj_ms += ms; tm.vert_adj_launches++; tm.cellsteps[3] += l.cellsteps; break;
            case 5: tm.route_fwd_ms += ms; tm.route_fwd_launches++; tm.cellsteps[1] += l.cellsteps;
                    tm.route_fwd_chained_ms += ms; tm.route_fwd_chained_launches++; break;
            case 6: tm.route_adj_ms += ms; tm.route_adj_launches++; tm.cellsteps[2] += l.cellsteps;
                    tm.route_adj_chained_ms += ms; tm.route_adj_chained_launches++; break;
            default: tm.cost_ms += ms; break;
        }
    }
    tm.n_chunks = p->nchunks; tm.chunk_steps = p->Tc; tm.pipe_steps = p->Tp; tm.n_rounds = p->sch.nrounds; tm.n_groups = p->sch.ngroups;
    tm.device_bytes = p->bytes;
    tm.max_stage = p->sch.max_stage;
    tm.chain_staged = (p->chain && p->A.qsk != nullptr) ? 1 : 0;
    tm.n_chained_groups = chain_first(p) < p->sch.nrounds ? p->sch.ngroups - p->sch.round_group_begin[chain_first(p)] : 0;
    p->last_adjoint = adjoint;
    return 0;
}

// Output_DT%qsim_domain / net_prcp_domain (or their sparse_ forms): host arrays the next FORWARD sweeps fill
int smashx_set_domain_outputs(smashx_plan* p, float* qsim_domain, float* net_prcp_domain, int sparse) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    if (sparse && !p->d_sparse_idx && (qsim_domain || net_prcp_domain))
        return fail(SMASHX_E_ARG, "sparse domain outputs need mesh.path (the sparse cell numbering)");
    // a tiled plan fills the cells of its own part and leaves -99 everywhere else (the caller overlays the parts): dense form
    // (nrow, ncol, nt), or sparse form (nac, nt) with nac the WHOLE grid's active cells -- a part's cells keep their whole-grid numbers
    p->h_qsim_domain = qsim_domain; p->h_net_prcp_domain = net_prcp_domain; p->dom_sparse = sparse ? 1 : 0;
    return 0;
}

// diagnostics: start/end ticks (100 MHz wall clock) of every routing group in the last forward (pass 0) and
// adjoint (pass 1) routing launches; needs SMASHX_TRACE_GROUPS=1 at plan creation.  out: [2][ngroups][2].
int smashx_debug_group_times(smashx_plan* p, long long* out, int* round_of_group) {
    if (!p || !out) return fail(SMASHX_E_ARG, "null argument");
    if (!p->A.gtime) return fail(SMASHX_E_STATE, "group tracing is off (SMASHX_TRACE_GROUPS=1)");
    int rc = set_device(p); if (rc) return rc;
    HIPCHK(hipMemcpy(out, p->A.gtime, (size_t)4 * p->sch.ngroups * sizeof(long long), hipMemcpyDeviceToHost));
    if (round_of_group)
        for (int r = 0; r < p->sch.nrounds; ++r)
            for (int g = p->sch.round_group_begin[r]; g < p->sch.round_group_begin[r + 1]; ++g) round_of_group[g] = r;
    return 0;
}

// device self-test: counts calls where sx_fdiv (sx_math.h) differs from the IEEE quotient a/b for n pseudo-random
// operand pairs with b in [blo, bhi); out[0] = mismatches, out[1] = mismatches larger than one ulp.
int smashx_selftest_math(int device, long long n, unsigned seed, float blo, float bhi, long long* out) {
    if (!out || n <= 0) return fail(SMASHX_E_ARG, "bad argument");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest_div, dim3(2048), dim3(256), 0, 0, (unsigned long long)n, seed, blo, bhi, d);
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SMASHX_E_HIP, hipGetErrorString(e));
    out[0] = (long long)h[0]; out[1] = (long long)h[1];
    return 0;
}

int smashx_selftest_paths(int device, long long n, unsigned seed, long long* out) {
    if (!out || n <= 0) return fail(SMASHX_E_ARG, "bad argument");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest_paths, dim3(2048), dim3(256), 0, 0, (unsigned long long)n, seed, d);
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SMASHX_E_HIP, hipGetErrorString(e));
    out[0] = (long long)h[0]; out[1] = (long long)h[1];
    return 0;
}

int smashx_tile_probe(const smashx_config* cfg, const smashx_mesh* mesh, int* info, int* out_src, int* out_dst,
                      int* in_src, int* in_dst, int cap) {
    if (!cfg || !mesh || !info) return fail(SMASHX_E_ARG, "null argument");
    SxSchedule sch;
    const bool tiled = cfg->tile[1] > cfg->tile[0] && cfg->tile[3] > cfg->tile[2];
    const int M = cfg->group_size > 0 ? cfg->group_size : 512;
    const int rc = sx_build_schedule(cfg->nrow, cfg->ncol, mesh->flwdir, mesh->active_cell, cfg->ng, mesh->gauge_pos, M,
                                     tiled ? cfg->tile : nullptr, sch, mesh->owner_mask, sublevels_env());
    if (rc) return fail(rc == -5 ? SMASHX_E_MESH : SMASHX_E_ARG, sch.error);
    const int no = (int)sch.out_x.size(), ni = (int)sch.in_x.size();
    const int v[8] = {sch.n, sch.nrounds, sch.ngroups, sch.nslots, sch.nxslots, sch.max_stage, no, ni};
    for (int i = 0; i < 8; ++i) info[i] = v[i];
    if (no > cap || ni > cap) { if (out_src || in_src) return fail(SMASHX_E_ARG, "edge capacity too small"); return 0; }
    for (int i = 0; i < no; ++i) { if (out_src) out_src[i] = sch.out_src[i]; if (out_dst) out_dst[i] = sch.out_dst[i]; }
    for (int i = 0; i < ni; ++i) { if (in_src) in_src[i] = sch.in_src[i]; if (in_dst) in_dst[i] = sch.in_dst[i]; }
    return 0;
}

int smashx_halo_counts(const smashx_plan* p, int* n_out, int* n_in) {
    if (!p || !n_out || !n_in) return fail(SMASHX_E_ARG, "null argument");
    *n_out = p->n_out; *n_in = p->n_in;
    return 0;
}

int smashx_halo_edges(const smashx_plan* p, int* out_src, int* out_dst, int* in_src, int* in_dst) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    for (int i = 0; i < p->n_out; ++i) { if (out_src) out_src[i] = p->sch.out_src[i]; if (out_dst) out_dst[i] = p->sch.out_dst[i]; }
    for (int i = 0; i < p->n_in; ++i) { if (in_src) in_src[i] = p->sch.in_src[i]; if (in_dst) in_dst[i] = p->sch.in_dst[i]; }
    return 0;
}

int smashx_plan_chunking(smashx_plan* p, int* chunk_steps, int* pipe_steps) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    int rc = set_device(p); if (rc) return rc;
    if ((rc = ensure_chunk_buffers(p, true))) return rc;
    if (chunk_steps) *chunk_steps = p->Tc;
    if (pipe_steps) *pipe_steps = p->Tp;
    return 0;
}

// out = {free HBM when the storage-chunk length was chosen (bytes; -1 before that), total HBM of the device, bytes the plan holds now}
int smashx_plan_hbm(const smashx_plan* p, double out[3]) {
    if (!p || !out) return fail(SMASHX_E_ARG, "null argument");
    out[0] = p->hbm_free_at_plan; out[1] = p->hbm_total; out[2] = p->bytes;
    return 0;
}

int smashx_set_halo(smashx_plan* p, float* d_out_buf, float* d_in_buf, smashx_halo_fn fn, void* user) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    if (fn && ((p->n_out > 0 && !d_out_buf) || (p->n_in > 0 && !d_in_buf))) return fail(SMASHX_E_ARG, "halo buffers missing");
    p->halo_out = d_out_buf; p->halo_in = d_in_buf; p->halo_fn = fn; p->halo_user = user;
    return 0;
}

int smashx_set_median_slots(smashx_plan* p, int nslots, const int* slot_of_gauge, smashx_reduce_fn fn, void* user) {
    if (!p || nslots < 0 || (nslots > 0 && !slot_of_gauge)) return fail(SMASHX_E_ARG, "bad argument");
    int rc = set_device(p); if (rc) return rc;
    p->med_nslots = 0; p->med_fn = fn; p->med_user = user;
    if (nslots == 0) return 0;
    std::vector<int> sl(std::max(p->ng, 1), -1);
    for (int g = 0; g < p->ng; ++g) {
        sl[g] = slot_of_gauge[g];
        if (sl[g] < -1 || sl[g] >= nslots) return fail(SMASHX_E_ARG, "slot_of_gauge out of range");
    }
    if (p->d_med_slot) { p->dfree(p->d_med_slot); p->dfree(p->d_medx); p->dfree(p->d_medx_idx); }
    if ((rc = p->upload_vec(&p->d_med_slot, sl))) return rc;
    p->med_slot_h = sl;
    if ((rc = p->dmalloc(&p->d_medx, (size_t)3 * nslots))) return rc;
    if ((rc = p->dmalloc(&p->d_medx_idx, (size_t)nslots))) return rc;
    p->med_nslots = nslots;
    return 0;
}

// ---- native exchange over RCCL -----------------------------------------------------------------------------------
int smashx_comm_unique_id(unsigned char id[SMASHX_COMM_ID_BYTES]) {
    if (!id) return fail(SMASHX_E_ARG, "null argument");
    RcclApi& R = rccl();
    if (!R.ok()) return fail(SMASHX_E_UNSUPPORTED, "RCCL unavailable: " + R.err);
    static_assert(sizeof(ncclUniqueId) == SMASHX_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCLCHK(R.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

int smashx_comm_create(const unsigned char id[SMASHX_COMM_ID_BYTES], int rank, int nranks, int device, void** comm) {
    if (!id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(SMASHX_E_ARG, "bad argument");
    *comm = nullptr;
    RcclApi& R = rccl();
    if (!R.ok()) return fail(SMASHX_E_UNSUPPORTED, "RCCL unavailable: " + R.err);
    if (device >= 0) HIPCHK(hipSetDevice(device));
    SxComm* c = new SxComm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = R.CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) { delete c; return fail(SMASHX_E_HIP, std::string("ncclCommInitRank: ") + R.GetErrorString(r)); }
    if (hipStreamCreate(&c->stream) != hipSuccess || hipMalloc((void**)&c->d_buf, 64 * sizeof(double)) != hipSuccess) {
        (void)R.CommDestroy(c->comm); delete c; return fail(SMASHX_E_HIP, "hipStreamCreate / hipMalloc failed");
    }
    *comm = c;
    return 0;
}

static void smashx_forget_comm(smashx_plan* p, SxComm* c) { if (p->xcomm == c) p->xcomm = nullptr; }
// a plan leaves the list of the communicator it pointed to whenever that pointer changes (unset, switched, plan destroyed): a
// communicator's list only ever holds live plans that still use it
static void smashx_detach_plan(smashx_plan* p) {
    if (SxComm* c = p->xcomm) c->plans.erase(std::remove(c->plans.begin(), c->plans.end(), p), c->plans.end());
    p->xcomm = nullptr;
}

int smashx_comm_destroy(void* comm) {
    SxComm* c = (SxComm*)comm;
    if (!c) return 0;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    for (smashx_plan* q : c->plans) smashx_forget_comm(q, c);      // no plan keeps a pointer to a communicator that is gone
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return 0;
}

// what the communicator itself says about the run: ranks it spans (ncclCommCount) and the library's version code (ncclGetVersion)
int smashx_comm_info(void* comm, int* nranks, int* version) {
    SxComm* c = (SxComm*)comm;
    if (!c) return fail(SMASHX_E_ARG, "null communicator");
    RcclApi& R = rccl();
    int n = c->nranks, v = 0;
    if (R.CommCount) NCCLCHK(R.CommCount(c->comm, &n));
    if (R.GetVersion) NCCLCHK(R.GetVersion(&v));
    if (nranks) *nranks = n;
    if (version) *version = v;
    return 0;
}

int smashx_comm_allreduce_sum(void* comm, double* values, int n) {
    SxComm* c = (SxComm*)comm;
    if (!c || !values || n < 1 || n > 64) return fail(SMASHX_E_ARG, "bad argument (1 <= n <= 64)");
    if (c->device >= 0) HIPCHK(hipSetDevice(c->device));
    RcclApi& R = rccl();
    HIPCHK(hipMemcpyAsync(c->d_buf, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(R.AllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(values, c->d_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int smashx_set_exchange(smashx_plan* p, void* comm, const int* out_peer, const int* in_peer) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    SxComm* c = (SxComm*)comm;
    if (!c) { smashx_detach_plan(p); return 0; }
    // The call is collective (an agreement all-reduce, then one grouped hello exchange): a rank that left early would leave its
    // neighbours waiting inside ncclGroupEnd for ever.  So everything that can fail locally -- argument checks, regrouping, allocations --
    // happens FIRST and only feeds a status word; the ranks then agree on the status together with the chunking, and enter the hello
    // group only if every rank reported success.
    int rc = 0, lrc = 0;
    std::string lerr;
    auto local = [&](int code, const std::string& msg) { if (!lrc) { lrc = code; lerr = msg; } };
    if ((p->n_out > 0 && !out_peer) || (p->n_in > 0 && !in_peer)) local(SMASHX_E_ARG, "peer lists missing");
    if ((rc = set_device(p))) return rc;            // (a wrong device id is a caller bug on this rank alone: nothing collective has started)
    if (!lrc && (rc = ensure_chunk_buffers(p, false))) local(rc, g_err);
    // regroup the boundary edges by peer rank; inside a peer they keep the order both sides share (sorted by source cell)
    std::vector<PeerSeg> out_segs, in_segs;
    std::vector<int> out_xp, in_xp;
    auto regroup = [&](int n, const int* peer, const std::vector<int>& xs, std::vector<PeerSeg>& segs, std::vector<int>& xp) {
        segs.clear();
        std::vector<int> order(n);
        xp.assign(std::max(n, 1), 0);
        for (int i = 0; i < n; ++i) {
            if (peer[i] < 0 || peer[i] >= c->nranks || peer[i] == c->rank) { local(SMASHX_E_ARG, "boundary edge with a bad peer rank"); return; }
            order[i] = i;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return peer[a] < peer[b]; });
        for (int i = 0; i < n; ++i) {
            xp[i] = xs[order[i]];
            if (segs.empty() || segs.back().rank != peer[order[i]]) segs.push_back(PeerSeg{peer[order[i]], i, 0});
            segs.back().count++;
        }
    };
    if (!lrc) regroup(p->n_out, out_peer, p->sch.out_x, out_segs, out_xp);
    if (!lrc) regroup(p->n_in, in_peer, p->sch.in_x, in_segs, in_xp);
    int *d_oxp = nullptr, *d_ixp = nullptr;
    float* hello = nullptr;
    std::vector<int> peers;
    if (!lrc) {
        for (const PeerSeg& sg : out_segs) peers.push_back(sg.rank);
        for (const PeerSeg& sg : in_segs) if (std::find(peers.begin(), peers.end(), sg.rank) == peers.end()) peers.push_back(sg.rank);
        std::sort(peers.begin(), peers.end());
        if ((rc = p->upload_vec(&d_oxp, out_xp)) || (rc = p->upload_vec(&d_ixp, in_xp)) || (rc = p->dmalloc(&hello, 2 * std::max<size_t>(peers.size(), 1))))
            local(rc, g_err);
        if (!lrc && !p->x_out) {
            if ((rc = p->dmalloc(&p->x_out, (size_t)std::max(p->n_out, 1) * p->Tp)) || (rc = p->dmalloc(&p->x_in, (size_t)std::max(p->n_in, 1) * p->Tp)))
                local(rc, g_err);
        }
    }
    auto drop_new = [&]() { if (d_oxp) p->dfree(d_oxp); if (d_ixp) p->dfree(d_ixp); if (hello) p->dfree(hello); };
    // every rank must cut time identically (the messages are per pipeline sub-chunk), and every rank must have got this far
    {
        RcclApi& R = rccl();
        double v[7] = {(double)p->Tc, -(double)p->Tc, (double)p->Tp, -(double)p->Tp, (double)p->nt, -(double)p->nt, lrc ? 1.0 : 0.0};
        HIPCHK(hipMemcpyAsync(c->d_buf, v, sizeof(v), hipMemcpyHostToDevice, c->stream));
        NCCLCHK(R.AllReduce(c->d_buf, c->d_buf, 7, ncclDouble, ncclMax, c->comm, c->stream));
        HIPCHK(hipMemcpyAsync(v, c->d_buf, sizeof(v), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (lrc) { drop_new(); return fail(lrc, lerr); }
        if (v[6] != 0.0) { drop_new(); return fail(SMASHX_E_STATE, "smashx_set_exchange failed on another rank of the decomposition"); }
        if (v[0] != -v[1] || v[2] != -v[3] || v[4] != -v[5]) {
            drop_new();
            return fail(SMASHX_E_ARG, "the ranks of the decomposition disagree on chunk_steps / pipe_steps / nt (this rank: " +
                                      std::to_string(p->Tc) + " / " + std::to_string(p->Tp) + " / " + std::to_string(p->nt) + ")");
        }
    }
    // Connect now, symmetrically: RCCL sets a point-to-point connection up the first time a pair is used, inside ncclGroupEnd on the
    // HOST, and both ends must be inside a group naming each other at that moment.  During a sweep the ranks reach their groups in
    // dependency order (a tile only sends after it has received and routed), so first use there makes rank A wait for B's NEXT
    // group while B waits for A's current one (observed: 3 sub-catchment parts hang in their first sub-chunk).  Here every rank
    // exchanges one float with each neighbour in both directions within ONE group -- the all-to-all shape RCCL's schedule is built
    // for -- so no connection is ever set up inside a sweep.
    if (!peers.empty()) {
        RcclApi& R = rccl();
        HIPCHK(hipMemsetAsync(hello, 0, 2 * peers.size() * sizeof(float), c->stream));
        NCCLCHK(R.GroupStart());
        for (size_t i = 0; i < peers.size(); ++i) {
            ncclResult_t r1 = R.Send(hello + 2 * i, 1, ncclFloat, peers[i], c->comm, c->stream);
            ncclResult_t r2 = R.Recv(hello + 2 * i + 1, 1, ncclFloat, peers[i], c->comm, c->stream);
            if (r1 != ncclSuccess || r2 != ncclSuccess) { (void)R.GroupEnd(); return fail(SMASHX_E_HIP, "ncclSend/ncclRecv (connection set-up) failed"); }
        }
        NCCLCHK(R.GroupEnd());
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    p->dfree(hello);
    if (p->d_out_xp) p->dfree(p->d_out_xp);          // a second call replaces the tables of the first
    if (p->d_in_xp) p->dfree(p->d_in_xp);
    p->d_out_xp = d_oxp; p->d_in_xp = d_ixp;
    p->out_segs.swap(out_segs); p->in_segs.swap(in_segs);
    if (p->xcomm != c) smashx_detach_plan(p);        // re-targeted: the previous communicator forgets this plan
    p->xcomm = c;
    if (std::find(c->plans.begin(), c->plans.end(), p) == c->plans.end()) c->plans.push_back(p);
    return 0;
}

// ---- control vector -----------------------------------------------------------------------------------------------------------
namespace {
int ctrl_fields(const smashx_plan* p, int* idx) {     // flagged fields the structure uses: parameters (0..15) then states (16..23)
    int nf = 0;
    for (int f = 0; f < SMASHX_GNP; ++f) if (p->opt.optim_parameters[f] > 0) idx[nf++] = f;
    for (int f = 0; f < SMASHX_GNS; ++f) if (p->opt.optim_states[f] > 0) idx[nf++] = SMASHX_GNP + f;
    return nf;
}
int ctrl_prepare(smashx_plan* p, int nf) {
    int rc;
    if (!p->d_ctrl_pos) {
        // rank of every plan cell among the active cells in (col outer, row inner) order = ascending flat index row + col * nrow
        std::vector<int> order(p->n), pos(p->n);
        for (int k = 0; k < p->n; ++k) order[k] = k;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return p->sch.cell_flat[a] < p->sch.cell_flat[b]; });
        for (int i = 0; i < p->n; ++i) pos[order[i]] = i;
        if ((rc = p->upload_vec(&p->d_ctrl_pos, pos))) return rc;
    }
    const size_t need = (size_t)nf * p->n;
    if (need > p->ctrl_cap) {
        if (p->d_ctrl) p->dfree(p->d_ctrl);
        if ((rc = p->dmalloc(&p->d_ctrl, need))) return rc;
        p->ctrl_cap = need;
    }
    return 0;
}
struct CtrlField { float* cellv; float* full; float lb, ub; bool used; };
CtrlField ctrl_field(smashx_plan* p, int idx, bool grad) {
    CtrlField F{nullptr, nullptr, 0.f, 1.f, false};
    float* pv[NPS] = {p->A.ci, p->A.cp, p->A.cft, p->A.cst, p->A.exc, p->A.lr, p->A.px[0], p->A.px[1], p->A.px[2]};
    float* pg[NPS] = {p->A.ci_b, p->A.cp_b, p->A.cft_b, p->A.cst_b, p->A.exc_b, p->A.lr_b, p->A.px_b[0], p->A.px_b[1], p->A.px_b[2]};
    float* sg[5] = {p->A.hi_b, p->A.hp_b, p->A.hft_b, p->A.hst_b, p->A.hlr_b};
    if (idx < SMASHX_GNP) {
        const int i = param_slot_of(p->st, idx);
        F.lb = p->opt.lb_parameters[idx]; F.ub = p->opt.ub_parameters[idx];
        if (i >= 0) { F.cellv = grad ? pg[i] : pv[i]; F.full = p->d_fullP[idx]; F.used = true; }
    } else {
        const int f = idx - SMASHX_GNP, i = state_slot_of(p->st, f);
        F.lb = p->opt.lb_states[f]; F.ub = p->opt.ub_states[f];
        if (i >= 0) { F.cellv = grad ? sg[i] : p->st0[i]; F.full = p->d_fullS[f]; F.used = true; }
    }
    return F;
}
}  // namespace

int smashx_control_size(smashx_plan* p) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    int idx[SMASHX_GNP + SMASHX_GNS];
    return ctrl_fields(p, idx) * p->n;
}

int smashx_control_set(smashx_plan* p, const double* x) {
    if (!p || !x) return fail(SMASHX_E_ARG, "null argument");
    if (!p->uploaded) return fail(SMASHX_E_STATE, "smashx_control_set needs one complete smashx_upload first (the fields that are not optimised)");
    if (p->tiled) return fail(SMASHX_E_UNSUPPORTED, "control vector on a tiled plan");
    int rc = set_device(p); if (rc) return rc;
    int idx[SMASHX_GNP + SMASHX_GNS];
    const int nf = ctrl_fields(p, idx);
    if (nf == 0) return fail(SMASHX_E_STATE, "no field is flagged in optim_parameters / optim_states");
    if ((rc = ctrl_prepare(p, nf))) return rc;
    HIPCHK(hipMemcpyAsync(p->d_ctrl, x, (size_t)nf * p->n * sizeof(double), hipMemcpyHostToDevice, p->stream));
    const dim3 b(256), gk((p->n + 255) / 256);
    for (int j = 0; j < nf; ++j) {
        const CtrlField F = ctrl_field(p, idx[j], false);
        float* jx = (p->opt.njr > 0 && p->d_jx[idx[j]]) ? p->d_jx[idx[j]] : nullptr;
        if (!F.used && !jx) continue;                // flagged but read by nothing: the reference carries it along, nothing here depends on it
        hipLaunchKernelGGL(k_control_set, gk, b, 0, p->stream, F.used ? F.cellv : nullptr, F.full, jx, p->d_ctrl, p->d_ctrl_pos, p->d_cell_flat,
                           p->n, (long)j * p->n, F.lb, F.ub, p->opt.denormalize_forward);
    }
    hipLaunchKernelGGL(sx_k_prep_routing, gk, b, 0, p->stream, p->A);
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

static int control_read(smashx_plan* p, double* x, int grad) {
    if (!p || !x) return fail(SMASHX_E_ARG, "null argument");
    if (!p->uploaded) return fail(SMASHX_E_STATE, "nothing uploaded");
    if (grad && !(p->adj_ready && p->last_adjoint)) return fail(SMASHX_E_STATE, "the last sweep was not an adjoint sweep");
    int rc = set_device(p); if (rc) return rc;
    int idx[SMASHX_GNP + SMASHX_GNS];
    const int nf = ctrl_fields(p, idx);
    if (nf == 0) return fail(SMASHX_E_STATE, "no field is flagged in optim_parameters / optim_states");
    if ((rc = ctrl_prepare(p, nf))) return rc;
    HIPCHK(hipMemsetAsync(p->d_ctrl, 0, (size_t)nf * p->n * sizeof(double), p->stream));
    const dim3 b(256), gk((p->n + 255) / 256);
    for (int j = 0; j < nf; ++j) {
        const CtrlField F = ctrl_field(p, idx[j], grad != 0);
        if (!F.used) {
            // a flagged field the structure does not read: only the regulariser knows it -- its plane (already in the optimiser's
            // space) or its gradient plane (COMPUTE_JREG_B, scaled like the download scales it)
            const float* plane = p->opt.njr > 0 ? (grad ? p->d_jg[idx[j]] : p->d_jx[idx[j]]) : nullptr;
            if (!plane) continue;
            hipLaunchKernelGGL(k_control_get, gk, b, 0, p->stream, p->d_ctrl, plane, p->d_ctrl_pos, p->d_cell_flat, p->n, (long)j * p->n,
                               F.lb, F.ub, grad ? p->opt.denormalize_forward : 0, grad, 1);
            continue;
        }
        hipLaunchKernelGGL(k_control_get, gk, b, 0, p->stream, p->d_ctrl, F.cellv, p->d_ctrl_pos, p->d_cell_flat, p->n, (long)j * p->n,
                           F.lb, F.ub, p->opt.denormalize_forward, grad, 0);
    }
    HIPCHK(hipMemcpyAsync(x, p->d_ctrl, (size_t)nf * p->n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    return 0;
}
int smashx_control_get(smashx_plan* p, double* x) { return control_read(p, x, 0); }
int smashx_control_gradient(smashx_plan* p, double* g) { return control_read(p, g, 1); }

int smashx_get_timing(const smashx_plan* p, smashx_timing* out) {
    if (!p || !out) return fail(SMASHX_E_ARG, "null argument");
    *out = p->timing;
    return 0;
}

int smashx_download(smashx_plan* p, int adjoint, smashx_parameters* params, smashx_states* states, float* qsim,
                    smashx_costs* costs, smashx_states* fstates, smashx_parameters* params_b, smashx_states* states_b) {
    if (!p) return fail(SMASHX_E_ARG, "null plan");
    int rc = set_device(p); if (rc) return rc;
    const int st = p->st;
    const dim3 b(256), gfull((unsigned)((p->n2 + 255) / 256)), gk((p->n + 255) / 256);
    // discharge at gauges: output%qsim(ng,nt)
    if (qsim && p->ng > 0) {
        std::vector<float> qg((size_t)p->ngc * p->nt);
        HIPCHK(hipMemcpy(qg.data(), p->A.qg, qg.size() * 4, hipMemcpyDeviceToHost));
        for (int g = 0; g < p->ng; ++g)
            for (int t = 0; t < p->nt; ++t) qsim[g + (size_t)p->ng * t] = qg[(size_t)p->gauge_gid[g] * p->nt + t];
    }
    if (costs) {
        float jobs = 0.f;
        if (p->ng > 0) HIPCHK(hipMemcpy(&jobs, p->d_cost_out, 4, hipMemcpyDeviceToHost));
        float jreg = 0.f;
        if (p->opt.njr > 0) {                        // mwd_cost.f90:199-228: weighted sums of the chains, then parameters + states
            float sums[SX_JREG_MAXCHAIN] = {0.f};
            HIPCHK(hipMemcpy(sums, p->d_jsum, sizeof(float) * 2 * p->opt.njr, hipMemcpyDeviceToHost));
            float pj = 0.f, sj = 0.f;
            for (int i = 0; i < p->opt.njr; ++i) {
                const float w = p->opt.wjreg_fun[i];
                const float ww = p->opt.jreg_fun[i] == SMASHX_PRIOR ? w : w * w;
                pj = pj + ww * sums[2 * i];
                sj = sj + ww * sums[2 * i + 1];
            }
            jreg = pj + sj;
        }
        costs->cost = jobs + p->opt.wjreg * jreg;    // mwd_cost.f90:300
        costs->cost_jobs = jobs; costs->cost_jreg = jreg;
    }
    // final states: output%fstates = states (forward.f90:71); inactive cells keep their entry values
    if (fstates && !adjoint) {
        float* cur[5] = {p->A.hi, p->A.hp, p->A.hft, p->A.hst, p->A.hlr};
        for (int i = 0; i < NSS; ++i) {
            const int f = state_field(st, i);
            if (f < 0 || !fstates->f[f]) continue;
            HIPCHK(hipMemcpyAsync(p->d_stage, p->d_fullS[f], (size_t)p->n2 * 4, hipMemcpyDeviceToDevice, p->stream));
            hipLaunchKernelGGL(k_scatter, gk, b, 0, p->stream, p->d_stage, cur[i], p->d_cell_flat, p->n, 1.f, 0);
            HIPCHK(hipMemcpyAsync(fstates->f[f], p->d_stage, (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
            HIPCHK(hipStreamSynchronize(p->stream));
        }
    }
    // parameters / states as the reference leaves them (forward.f90:33-38,72; mwd_cost.f90:284-298):
    // denormalised; base_forward additionally sends them through normalise -> denormalise inside compute_cost.
    if (p->opt.denormalize_forward) {
        bool touched = false;
        for (int i = 0; i < NPS; ++i) {
            const int f = param_field(st, i);
            if (f < 0 || !params || !params->f[f]) continue;
            touched = true;
            if (!adjoint) {
                hipLaunchKernelGGL(k_normalize, gfull, b, 0, p->stream, p->d_fullP[f], p->n2, p->opt.lb_parameters[f], p->opt.ub_parameters[f]);
                hipLaunchKernelGGL(k_denormalize, gfull, b, 0, p->stream, p->d_fullP[f], p->n2, p->opt.lb_parameters[f], p->opt.ub_parameters[f]);
            }
            HIPCHK(hipMemcpyAsync(params->f[f], p->d_fullP[f], (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
        }
        for (int i = 0; i < NSS; ++i) {
            const int f = state_field(st, i);
            if (f < 0 || !states || !states->f[f]) continue;
            touched = true;
            if (!adjoint) {
                hipLaunchKernelGGL(k_normalize, gfull, b, 0, p->stream, p->d_fullS[f], p->n2, p->opt.lb_states[f], p->opt.ub_states[f]);
                hipLaunchKernelGGL(k_denormalize, gfull, b, 0, p->stream, p->d_fullS[f], p->n2, p->opt.lb_states[f], p->opt.ub_states[f]);
            }
            HIPCHK(hipMemcpyAsync(states->f[f], p->d_fullS[f], (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
        }
        HIPCHK(hipStreamSynchronize(p->stream));
        if (touched) p->uploaded = false;   // the caller now holds denormalised fields: a full upload is required before the next sweep
    }
    if (adjoint) {
        if (!p->adj_ready) return fail(SMASHX_E_STATE, "no adjoint sweep has run");
        float* gp[NPS] = {p->A.ci_b, p->A.cp_b, p->A.cft_b, p->A.cst_b, p->A.exc_b, p->A.lr_b, p->A.px_b[0], p->A.px_b[1], p->A.px_b[2]};
        float* gs[5] = {p->A.hi_b, p->A.hp_b, p->A.hft_b, p->A.hst_b, p->A.hlr_b};
        // parameters_b / states_b are fully overwritten (forward_db.f90:10869-10870); DENORMALIZE_*_B multiplies by (ub-lb)
        if (params_b)
            for (int f = 0; f < SMASHX_GNP; ++f) {
                if (!params_b->f[f]) continue;
                const int i = param_slot_of(st, f);
                const bool jr = p->opt.njr > 0 && jreg_optim(p, f) > 0;      // inactive cells / unused fields: the regulariser's part
                if (jr) hipLaunchKernelGGL(sx_k_plane_scale, gfull, b, 0, p->stream, p->d_stage, p->d_jg[f],
                                           p->opt.ub_parameters[f] - p->opt.lb_parameters[f], p->opt.denormalize_forward, p->n2);
                else HIPCHK(hipMemsetAsync(p->d_stage, 0, (size_t)p->n2 * 4, p->stream));
                if (i < 0) {
                    if (!jr) { std::memset(params_b->f[f], 0, (size_t)p->n2 * 4); continue; }
                    HIPCHK(hipMemcpyAsync(params_b->f[f], p->d_stage, (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
                    HIPCHK(hipStreamSynchronize(p->stream));
                    continue;
                }
                hipLaunchKernelGGL(k_scatter, gk, b, 0, p->stream, p->d_stage, gp[i], p->d_cell_flat, p->n,
                                   p->opt.ub_parameters[f] - p->opt.lb_parameters[f], p->opt.denormalize_forward);
                HIPCHK(hipMemcpyAsync(params_b->f[f], p->d_stage, (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
                HIPCHK(hipStreamSynchronize(p->stream));
            }
        if (states_b)
            for (int f = 0; f < SMASHX_GNS; ++f) {
                if (!states_b->f[f]) continue;
                const int i = state_slot_of(st, f);
                const bool jr = p->opt.njr > 0 && jreg_optim(p, SMASHX_GNP + f) > 0;
                if (jr) hipLaunchKernelGGL(sx_k_plane_scale, gfull, b, 0, p->stream, p->d_stage, p->d_jg[SMASHX_GNP + f],
                                           p->opt.ub_states[f] - p->opt.lb_states[f], p->opt.denormalize_forward, p->n2);
                else HIPCHK(hipMemsetAsync(p->d_stage, 0, (size_t)p->n2 * 4, p->stream));
                if (i < 0) {
                    if (!jr) { std::memset(states_b->f[f], 0, (size_t)p->n2 * 4); continue; }
                    HIPCHK(hipMemcpyAsync(states_b->f[f], p->d_stage, (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
                    HIPCHK(hipStreamSynchronize(p->stream));
                    continue;
                }
                hipLaunchKernelGGL(k_scatter, gk, b, 0, p->stream, p->d_stage, gs[i], p->d_cell_flat, p->n,
                                   p->opt.ub_states[f] - p->opt.lb_states[f], p->opt.denormalize_forward);
                HIPCHK(hipMemcpyAsync(states_b->f[f], p->d_stage, (size_t)p->n2 * 4, hipMemcpyDeviceToHost, p->stream));
                HIPCHK(hipStreamSynchronize(p->stream));
            }
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int smashx_forward(smashx_plan* p, smashx_parameters* params, const smashx_parameters* params_bgd, smashx_states* states,
                   const smashx_states* states_bgd, float* qsim, smashx_costs* costs, smashx_states* fstates) {
    int rc;
    if ((rc = smashx_upload(p, params, params_bgd, states, states_bgd))) return rc;
    if ((rc = smashx_sweep(p, 0, 0.f))) return rc;
    return smashx_download(p, 0, params, states, qsim, costs, fstates, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// tangent model: base_forward_d (forward_db.f90:10517-10601).  One forward sweep that carries the directional
// derivative along (params_d, states_d): vertical kernel on dual numbers, routing twice (values with the hr_imd
// tape, then tangents), cost_d = <d cost / d qsim, qsim_d> + <d(wjreg jreg) / d theta, theta_d> with the adjoint
// seeds of the cost kernels and of the regulariser (fp64 dot products).
// ---------------------------------------------------------------------------------------------------------
int smashx_forward_d(smashx_plan* p, smashx_parameters* params, const smashx_parameters* params_d, const smashx_parameters* params_bgd,
                     smashx_states* states, const smashx_states* states_d, const smashx_states* states_bgd, float* qsim,
                     float* qsim_d, smashx_costs* costs, float* cost_d) {
    if (!p || !params || !states || !params_d || !states_d || !cost_d) return fail(SMASHX_E_ARG, "null argument");
    // a tiled plan: the boundary series of the value pass and then of the tangent pass travel like the forward sweep's (same hooks,
    // same phases), a message per pipeline sub-chunk; the criteria's tangent covers this part's gauges, the regulariser's the whole grid
    // (smashx_tangent_terms gives the two apart: a decomposition adds the parts' first terms and counts the second once)
    if (p->tiled && p->med_nslots > 0) return fail(SMASHX_E_UNSUPPORTED, "tangent model on a tiled plan with the median over gauges of several parts");
    int rc = smashx_upload(p, params, params_bgd, states, states_bgd); if (rc) return rc;
    if (!p->have_forcing) return fail(SMASHX_E_STATE, "forcing not set");
    if ((rc = set_device(p))) return rc;
    if ((rc = ensure_chunk_buffers(p, true))) return rc;
    const int st = p->st;
    hipStream_t sV = p->stream, sR = p->stream_r;
    const dim3 b(256), gfull((unsigned)((p->n2 + 255) / 256)), gk((p->n + 255) / 256);
    if (!p->tan_ready) {
        if ((rc = p->dmalloc(&p->A.qtdT, (size_t)p->npad * p->Tc))) return rc;
        if ((rc = p->dmalloc(&p->A.xdT, (size_t)std::max(p->sch.nxslots, 1) * p->Tc))) return rc;
        if ((rc = p->dmalloc(&p->A.qgd, (size_t)std::max(p->ngc, 1) * p->nt))) return rc;
        if ((rc = p->dmalloc(&p->d_qsim_d, (size_t)std::max(p->ng, 1) * p->nt))) return rc;
        HIPCHK(hipMemset(p->A.xdT, 0, (size_t)std::max(p->sch.nxslots, 1) * p->Tc * 4));
        p->tan_ready = true;
    }
    // direction -> per-cell arrays (the gradient arrays double as tangent storage); DENORMALIZE_*_D scales by (ub - lb)
    float* tp[NPS] = {p->A.ci_b, p->A.cp_b, p->A.cft_b, p->A.cst_b, p->A.exc_b, p->A.lr_b, p->A.px_b[0], p->A.px_b[1], p->A.px_b[2]};
    float* ts[5] = {p->A.hi_b, p->A.hp_b, p->A.hft_b, p->A.hst_b, p->A.hlr_b};
    for (int i = 0; i < NPS; ++i) {
        const int f = param_field(st, i);
        HIPCHK(hipMemsetAsync(tp[i], 0, (size_t)p->npad * 4, sV));
        if (f < 0) continue;
        if (!params_d->f[f]) return fail(SMASHX_E_ARG, "a tangent field the structure uses is NULL");
        HIPCHK(hipMemcpyAsync(p->d_stage, params_d->f[f], (size_t)p->n2 * 4, hipMemcpyHostToDevice, sV));
        if (p->opt.denormalize_forward)
            hipLaunchKernelGGL(sx_k_plane_scale, gfull, b, 0, sV, p->d_stage, p->d_stage, p->opt.ub_parameters[f] - p->opt.lb_parameters[f], 1, p->n2);
        hipLaunchKernelGGL(k_gather, gk, b, 0, sV, tp[i], p->d_stage, p->d_cell_flat, p->n);
        HIPCHK(hipStreamSynchronize(sV));
    }
    for (int i = 0; i < 5; ++i) {
        const int f = state_field(st, i);
        HIPCHK(hipMemsetAsync(ts[i], 0, (size_t)p->npad * 4, sV));
        if (f < 0) continue;
        if (!states_d->f[f]) return fail(SMASHX_E_ARG, "a tangent field the structure uses is NULL");
        HIPCHK(hipMemcpyAsync(p->d_stage, states_d->f[f], (size_t)p->n2 * 4, hipMemcpyHostToDevice, sV));
        if (p->opt.denormalize_forward)
            hipLaunchKernelGGL(sx_k_plane_scale, gfull, b, 0, sV, p->d_stage, p->d_stage, p->opt.ub_states[f] - p->opt.lb_states[f], 1, p->n2);
        hipLaunchKernelGGL(k_gather, gk, b, 0, sV, ts[i], p->d_stage, p->d_cell_flat, p->n);
        HIPCHK(hipStreamSynchronize(sV));
    }
    // regulariser tangent (COMPUTE_JREG_D, forward_db.f90:2810-2925): the reference's running sums in its order
    float jreg_d = 0.f;
    if (p->opt.njr > 0) {
        HIPCHK(hipEventRecord(p->ev0, sV));
        HIPCHK(hipStreamWaitEvent(p->stream_j, p->ev0, 0));
        if ((rc = run_jreg(p, 0, 0.f))) return rc;                        // jreg itself (cost), fills p->jchains
        HIPCHK(hipStreamSynchronize(p->stream_j));
        float sums_keep[SX_JREG_MAXCHAIN] = {0.f};
        HIPCHK(hipMemcpy(sums_keep, p->d_jsum, sizeof(float) * 2 * p->opt.njr, hipMemcpyDeviceToHost));
        const int nrow = p->cfg.nrow, ncol = p->cfg.ncol;
        // tangent of the control vector as compute_cost sees it: NORMALIZE_D(DENORMALIZE_D(direction))
        for (int idx = 0; idx < SMASHX_GNP + SMASHX_GNS; ++idx) {
            if (jreg_optim(p, idx) <= 0) continue;
            const float* h = idx < SMASHX_GNP ? params_d->f[idx] : states_d->f[idx - SMASHX_GNP];
            if (!h) return fail(SMASHX_E_ARG, "the tangent of an optimised field is NULL");
            if (!p->d_tanplane[idx]) { if ((rc = p->dmalloc(&p->d_tanplane[idx], (size_t)p->n2))) return rc; }
            HIPCHK(hipMemcpyAsync(p->d_tanplane[idx], h, (size_t)p->n2 * 4, hipMemcpyHostToDevice, sV));
            if (p->opt.denormalize_forward) {
                hipLaunchKernelGGL(sx_k_plane_scale, gfull, b, 0, sV, p->d_tanplane[idx], p->d_tanplane[idx], jreg_span(p, idx), 1, p->n2);
                hipLaunchKernelGGL(sx_k_plane_div, gfull, b, 0, sV, p->d_tanplane[idx], jreg_span(p, idx), p->n2);
            }
        }
        const SxJregChains& ch = p->jchains;
        for (int i = 0; i < p->opt.njr; ++i)
            for (int grp = 0; grp < 2; ++grp) {
                const int c = 2 * i + grp, lo = grp ? SMASHX_GNP : 0, hi = grp ? SMASHX_GNP + SMASHX_GNS : SMASHX_GNP;
                long slot = ch.first[c];
                for (int idx = lo; idx < hi; ++idx) {
                    if (jreg_optim(p, idx) <= 0) continue;
                    float* t = p->d_jterm + (size_t)slot++ * p->n2;
                    if (p->opt.jreg_fun[i] == SMASHX_PRIOR)
                        hipLaunchKernelGGL(sx_k_prior_terms_d, gfull, b, 0, sV, t, p->d_jx[idx], p->d_jb[idx], p->d_tanplane[idx], p->n2);
                    else
                        hipLaunchKernelGGL(sx_k_smooth_terms_d, gfull, b, 0, sV, t, p->d_jx[idx], p->d_jb[idx], p->d_tanplane[idx],
                                           p->opt.jreg_fun[i] == SMASHX_SMOOTHING ? 1 : 0, p->d_active, nrow, ncol);
                }
            }
        hipLaunchKernelGGL(sx_k_seq_sum, dim3(ch.nchain), dim3(64), 0, sV, p->d_jterm, ch, p->n2, p->d_jsum);
        float sd[SX_JREG_MAXCHAIN] = {0.f};
        HIPCHK(hipMemcpyAsync(sd, p->d_jsum, sizeof(float) * 2 * p->opt.njr, hipMemcpyDeviceToHost, sV));
        HIPCHK(hipStreamSynchronize(sV));
        HIPCHK(hipMemcpy(p->d_jsum, sums_keep, sizeof(float) * 2 * p->opt.njr, hipMemcpyHostToDevice));   // the download reads jreg from here
        float pj = 0.f, sj = 0.f;
        for (int i = 0; i < p->opt.njr; ++i) {
            const float w = p->opt.wjreg_fun[i];
            const float ww = p->opt.jreg_fun[i] == SMASHX_PRIOR ? w : w * w;
            pj = pj + ww * sd[2 * i];
            sj = sj + ww * sd[2 * i + 1];
        }
        jreg_d = pj + sj;
    }
    // sweep
    p->launches.clear(); p->pool_used = 0;
    if ((rc = close_forcing(p))) return rc;
    HIPCHK(hipEventRecord(p->ev0, sV));
    HIPCHK(hipStreamWaitEvent(sR, p->ev0, 0));
    p->chain_used = false;
    p->dom_q_active = false;
    HIPCHK(hipMemsetAsync(p->A.prog + p->sch.ngroups, 0, sizeof(int), sR));
    if ((rc = restore_states(p, p->st0))) return rc;
    const dim3 vgrid(p->npad / SX_VBLOCK), vblock(SX_VBLOCK);
    const size_t lds = (size_t)2 * p->M * sizeof(float4);
    for (int c = 0; c < p->nchunks; ++c) {
        const int t0c = c * p->Tc, Tcur = chunk_len(p, c);
        if (c > 0) { hipEvent_t e = p->event(); HIPCHK(hipEventRecord(e, sR)); HIPCHK(hipStreamWaitEvent(sV, e, 0)); }
        const SxDeviceArrays B = view_at(p, 0);
        p->mark_begin(0, sV);
        switch (st) {
            case 1: hipLaunchKernelGGL((sx_k_vert_fwd_d<1>), vgrid, vblock, 0, sV, B, t0c, Tcur); break;
            case 2: hipLaunchKernelGGL((sx_k_vert_fwd_d<2>), vgrid, vblock, 0, sV, B, t0c, Tcur); break;
            case 3: hipLaunchKernelGGL((sx_k_vert_fwd_d<3>), vgrid, vblock, 0, sV, B, t0c, Tcur); break;
            case 5: hipLaunchKernelGGL(sx_k_vert_fwd_vic_d, vgrid, vblock, 0, sV, B, t0c, Tcur); break;
            default: hipLaunchKernelGGL((sx_k_vert_fwd_d<4>), vgrid, vblock, 0, sV, B, t0c, Tcur); break;
        }
        p->mark_end();
        hipEvent_t e = p->event(); HIPCHK(hipEventRecord(e, sV)); HIPCHK(hipStreamWaitEvent(sR, e, 0));
        // routing: values (forward_d's primal forms, hr_imd tape on), then tangents; one launch per round.  A plan with boundary series
        // (tiles) cuts each pass into the pipeline sub-chunks its message buffers hold: receive + unpack, route, pack + send
        const bool native = p->xcomm != nullptr;
        const bool halo = (p->halo_fn || native) && (p->n_out > 0 || p->n_in > 0);
        const int Tsub = halo ? p->Tp : Tcur;
        for (int pass = 1; pass <= 2; ++pass)
            for (int off = 0; off < Tcur; off += Tsub) {
                const int T = std::min(Tsub, Tcur - off);
                SxDeviceArrays Bt = view_at(p, off); Bt.qdT = nullptr;
                Bt.qsk = nullptr;            // one launch per round: every series in its plain row
                float* xarr = pass == 1 ? p->A.xT : p->A.xdT;
                if (halo && p->n_in > 0) {
                    if ((rc = sx_halo_hook(p, native, 0, t0c + off, T, sR))) return rc;
                    sx_halo_move(p, native, xarr, false, false, off, T, sR);
                }
                for (int r = 0; r < p->sch.nrounds; ++r) {
                    const int g0 = p->sch.round_group_begin[r], ngr = p->sch.round_group_begin[r + 1] - g0;
                    p->mark_begin(1, sR);
                    if (pass == 1) hipLaunchKernelGGL((sx_k_route_fwd<true, false, 1>), dim3(ngr), dim3(p->M), lds, sR, Bt, g0, g0 + ngr, t0c + off, T);
                    else           hipLaunchKernelGGL((sx_k_route_fwd<false, false, 2>), dim3(ngr), dim3(p->M), lds, sR, Bt, g0, g0 + ngr, t0c + off, T);
                    p->mark_end();
                }
                if (halo && p->n_out > 0) {
                    sx_halo_move(p, native, xarr, true, true, off, T, sR);
                    if ((rc = sx_halo_hook(p, native, 1, t0c + off, T, sR))) return rc;
                }
            }
    }
    // cost (values) and its tangent in the reference's summation order
    if ((rc = run_cost(p, 0, 0.f))) return rc;
    float jobs_d = 0.f;
    if (p->ng > 0) {
        SxCostArgs C = cost_args(p, 0.f);
        hipLaunchKernelGGL(k_gauge_rows, dim3((p->nt + 255) / 256, p->ng), dim3(256), 0, sR, p->d_qsim_d, p->A.qgd, p->d_gauge_gid, p->ng, p->nt);
        hipLaunchKernelGGL(sx_k_cost_tangent, dim3(1), dim3(64), 0, sR, C, p->A.qgd, p->d_cost_out + 1);
        HIPCHK(hipMemcpyAsync(&jobs_d, p->d_cost_out + 1, sizeof(float), hipMemcpyDeviceToHost, sR));
    }
    HIPCHK(hipStreamSynchronize(sV));
    HIPCHK(hipStreamSynchronize(sR));
    HIPCHK(hipGetLastError());
    if (p->chain_used) {
        int stalled = 0;
        HIPCHK(hipMemcpy(&stalled, p->A.prog + p->sch.ngroups, sizeof(int), hipMemcpyDeviceToHost));
        if (stalled) return fail(SMASHX_E_HIP, "chained routing launch stalled");
    }
    *cost_d = jobs_d + p->opt.wjreg * jreg_d;                        // COMPUTE_COST_D (forward_db.f90:3248)
    p->last_jobs_d = jobs_d; p->last_jreg_d = jreg_d;
    if (qsim_d && p->ng > 0) {
        std::vector<float> qd((size_t)p->ng * p->nt);
        HIPCHK(hipMemcpy(qd.data(), p->d_qsim_d, qd.size() * 4, hipMemcpyDeviceToHost));
        for (int g = 0; g < p->ng; ++g)
            for (int t = 0; t < p->nt; ++t) qsim_d[g + (size_t)p->ng * t] = qd[(size_t)g * p->nt + t];
    }
    p->last_adjoint = 0;
    // primal outputs exactly like base_forward (parameters / states denormalised + round trip, states restored)
    return smashx_download(p, 0, params, states, qsim, costs, nullptr, nullptr, nullptr);
}

int smashx_tangent_terms(const smashx_plan* p, float* jobs_d, float* jreg_d) {
    if (!p || !jobs_d || !jreg_d) return fail(SMASHX_E_ARG, "null argument");
    *jobs_d = p->last_jobs_d; *jreg_d = p->last_jreg_d;
    return 0;
}

int smashx_forward_b(smashx_plan* p, smashx_parameters* params, const smashx_parameters* params_bgd, smashx_states* states,
                     const smashx_states* states_bgd, float cost_b, float* qsim, smashx_costs* costs,
                     smashx_parameters* params_b, smashx_states* states_b) {
    int rc;
    if ((rc = smashx_upload(p, params, params_bgd, states, states_bgd))) return rc;
    if ((rc = smashx_sweep(p, 1, cost_b))) return rc;
    return smashx_download(p, 1, params, states, qsim, costs, nullptr, params_b, states_b);
}

}  // extern "C"
