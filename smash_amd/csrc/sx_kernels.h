// sx_kernels.h -- HIP kernels of the smashx hot path (gfx950 / MI355X, wave64).
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   cells are renumbered k = 0..n-1 in routing-group order (sx_plan.h); npad = n rounded up to 256.
//   forcing      prcp/pet [nt][npad]     cell index fastest  -> lanes = consecutive cells, coalesced
//   params/state [npad] per field
//   qtT, hrT     [Tc/4][npad][4]         "T4": one float4 = 4 consecutive steps of one cell; cell index next.
//                hrT belongs to the routing kernels alone; with hr_skew its row index is shifted by the cell's stage in its group:
//                [Tc/4 + max stage][npad][4]
//                                        The marching vertical threads (lanes = consecutive cells) move
//                                        1 KiB contiguous per wave instruction; the time-skewed routing threads
//                                        of a group read rows tb = w - stage, and because the group's slots are
//                                        in breadth-first order neighbouring cells sit one stage apart, so a
//                                        128-B line is re-touched within a super-step or two (L1/L2 hits)
//   tape_*       [Tc][npad]              pre-step reservoir levels of the current time chunk
//   exchange     [Tc/4][nx][4]           discharge (forward) / adjoint (reverse) series between groups, T4
//   gauge series [ngauge][nt]
#pragma once

#include <hip/hip_runtime.h>

#include <climits>

#include "sx_ops.h"
#include "sx_tangent.h"
#include "sx_vic.h"

#define SX_BT 4           // time steps per routing super-step (one float4 per cell per super-step)
#define SX_VBLOCK 256     // threads (cells) per vertical workgroup

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (every global store and
// prefetch load in flight), which put a full HBM round trip into each routing super-step: 77 -> (see
// profiles/) ms per routing pass at 1024^2 x 8760.  The exchange between super-steps goes through LDS alone;
// global data written here is only read by later kernels.
__device__ __forceinline__ void sx_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// between two sub-levels of one wavefront (sx_plan.h "components"): writes of the lower sub-level before reads of the higher one
__device__ __forceinline__ void sx_lds_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Loads that must be global_load (vmcnt only).  When a pointer is selected between two buffers hipcc loses the
// address space and emits flat_load, which also counts on lgkmcnt -- the counter the LDS barrier waits on -- so
// every super-step barrier waited for the staged HBM loads (the routing tail rounds ran at ~2 us per super-step).
typedef float sx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 sx_gload4(const float4* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const sx_f4v v = *(const __attribute__((address_space(1))) sx_f4v*)p;
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
// streaming variants (SX_R_NT: 1 = loads, 2 = stores, 3 = both) for the cell-indexed arrays of the routing kernels (qt, hr tape,
// qt_b): read once / written once per pass.  Never for the exchange rows (xT), which neighbouring groups re-read through L2: stores
// whose destination may be such a row go through sx_gstore4 (always plain).
// Measured at 1024^2 x 8760 (sweep 172.8 ms with 0): loads 175.3, stores 181.1 (route_fwd 24.8 -> 33.6 ms), both 180.9 -> off.
#ifndef SX_R_NT
#define SX_R_NT 0
#endif
__device__ __forceinline__ float4 sx_gload4s(const float4* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (SX_R_NT & 1) {
        const sx_f4v v = __builtin_nontemporal_load((const __attribute__((address_space(1))) sx_f4v*)p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
#endif
    return sx_gload4(p);
}
__device__ __forceinline__ void sx_gstore4s(float4* p, const float4& q) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (SX_R_NT & 2) {
        sx_f4v v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
        __builtin_nontemporal_store(v, (__attribute__((address_space(1))) sx_f4v*)p);
        return;
    }
    {   // global_store, never flat_store: the pointer may have been selected between two buffers (see sx_gload4)
        sx_f4v v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
        *(__attribute__((address_space(1))) sx_f4v*)p = v;
        return;
    }
#endif
    *p = q;
}
// always a plain global store (never nontemporal, whatever SX_R_NT says): for destinations that may be exchange rows (xT), which
// other groups re-read through L2 behind the progress counters
__device__ __forceinline__ void sx_gstore4(float4* p, const float4& q) {
#if defined(__HIP_DEVICE_COMPILE__)
    sx_f4v v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    *(__attribute__((address_space(1))) sx_f4v*)p = v;
#else
    *p = q;
#endif
}
__device__ __forceinline__ float sx_gload1(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(1))) float*)p;
#else
    return *p;
#endif
}

// "The staged loads of the previous macro-step must have landed HERE": consuming the registers in an opaque asm
// makes hipcc place its vmcnt wait at this point -- before the next macro-step's loads are issued -- instead of at
// the first arithmetic use, where the in-order vmcnt(0) would also wait for the loads issued just before it.
__device__ __forceinline__ void sx_pin(float4& v) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : : "memory");
}

__device__ __forceinline__ void sx_pin1(float& v) { asm volatile("" : "+v"(v) : : "memory"); }
__device__ __forceinline__ void sx_pinu(unsigned& v) { asm volatile("" : "+v"(v) : : "memory"); }

// ---- rounds chained inside one launch (DESIGN.md "Chained rounds") --------------------------------------
// A group of round r only needs, for time block b, what the groups of earlier rounds published for block b,
// so consecutive rounds can run in ONE launch as a wavefront over (round, time): a producer group counts the
// blocks it has published in prog[g], a consumer polls that counter before it requests a block.
// Coherence (the 8 XCD L2s are not coherent with each other): the series themselves use ordinary loads and
// stores -- agent-coherent (sc1) accesses cost ~8 us each on MI355X and stalled every macro-step -- and the
// counter carries release / acquire at agent scope: every SX_PK macro-steps the producer, once all its waves
// have drained their stores (vmcnt(0), then the workgroup barrier), writes its L2 back (release fence) and
// bumps the counter; a consumer that sees the counter move invalidates its caches (acquire fence) before it
// requests the newly published blocks.
// Forward progress (round 3): a chained launch no longer relies on the order in which the hardware dispatches workgroups.  Every
// workgroup that becomes resident draws the next TICKET (one atomic add) and routes the group of that ticket -- tickets run through
// the chained groups in dependency order (forward: ascending group id; reverse: descending), so a group only ever waits for groups
// whose tickets were drawn earlier, i.e. by workgroups that are resident or done -- and loops until the tickets are exhausted (the
// grid may be smaller than the number of groups).
typedef __attribute__((address_space(1))) int sx_gint;
#define SX_PROG_STALL(A) ((A).prog + (A).ngroups)          // stall flag
#define SX_PROG_TICKET(A) ((A).prog + (A).ngroups + 1)     // next ticket of the running chained launch
#define SX_PROG_EXTRA 8
#ifndef SX_PK
#define SX_PK 16              // macro-steps between two publications (x SX_MU x SX_BT = 256 time steps); 2..16 measured, fences dominate
#endif
#define SX_SPIN_LIMIT (1 << 22)   // polls (~1 us each at least) before a stalled chain is reported instead of hanging (SxDeviceArrays::spin_limit)
// stalled[0] = flag; stalled[3..6] = diagnostics of the first waiter that gave up: tag (the waiting group), the
// counter it followed (address distance to the flag, in ints), the blocks it needed and the blocks it last saw
__device__ __forceinline__ void sx_wait_prog(const int* prog, int need, int& seen, int* stalled, int limit, int tag = -1) {
    if (seen >= need) return;
    int spins = 0;
    while (seen < need) {
        seen = __hip_atomic_load((const sx_gint*)prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen < need) {
            // a wait that runs out of polls voids the sweep (the host repeats it without chained launches); once one waiter has
            // given up every other one follows at its next look at the flag instead of serving its own full limit
            const bool out = ++spins > limit;
            if (out || (spins & 1023) == 0) {
                if (out) {
                    if (atomicCAS(stalled + 3, 0, tag + 2) == 0) { stalled[4] = (int)(prog - stalled); stalled[5] = need; stalled[6] = seen; }
                    __hip_atomic_store((sx_gint*)stalled, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (out || __hip_atomic_load((const sx_gint*)stalled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { seen = 0x7fffffff; break; }
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void sx_publish(int* prog, int blocks) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store((sx_gint*)prog, blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct SxDeviceArrays {
    // sizes
    int n, npad, nt, Tc;          // Tc = allocated chunk length (multiple of 16)
    int k0, k1;                   // cell range [k0, k1) of a vertical launch (whole domain: 0, n)
    int nx;                       // exchange series count (>= 1)
    float dt, dx;
    // forcing: fp32 rows [nt][npad], or -- prcp16 != null -- the lossless compact form (smashx_set_forcing_layout):
    //   prcp16 [nt][npad] u16: rain depth in raster units, prcp = real(k) * prcp_c exactly as the reference's reader forms it
    //          (_read_input_data.py:191-196); 65535 = the gap marker prcp_gap
    //   petd   [ndays][npad]: daily PET; pet(t) = petd(day(t)) * pet_ratio(hour(t)), the reader's product (:255-283); a negative
    //          daily value is a gap day and stands for every hour
    const float* prcp; const float* pet;
    const unsigned short* prcp16; const float* petd; const float* pet_ratio;   // pet_ratio: 24 floats in HBM, read by the scalar unit
    float prcp_c, prcp_gap; int hour0;
    // parameters (denormalised) and per-cell invariants
    float *ci, *cp, *cft, *cst, *exc, *lr;
    // vic-a reuses the slots above (b -> ci, cusl1 -> cp, cusl2 -> cft, clsl -> cst, ks -> exc; husl1 -> hi, husl2 -> hp,
    // hlsl -> hft, same for the tapes and gradients) and adds three: ds, dsm, ws
    float* px[3]; float* px_b[3];
    float *rt_a, *rt_f, *rt_denf, *rt_denb;   // exp(-dt/(60 lr)), real(flwacc-1), forward / adjoint denominators
    int* flwacc;
    // states (running) and adjoint state
    float *hi, *hp, *hft, *hst, *hlr;
    float *ci_b, *cp_b, *cft_b, *cst_b, *exc_b, *lr_b, *hi_b, *hp_b, *hft_b, *hst_b, *hlr_b;
    // chunk buffers
    float *qtT, *hrT;
    int hr_skew;                  // 1: row of hrT = time block + the slot's stage (what a routing group touches in one super-step is one row)
    // tangent sweep (base_forward_d): qt_d per cell (T4 like qtT), exchange series of q_d, q_d at the gauge cells.
    // The tangents of parameters and states live in the gradient arrays (ci_b .. hlr_b) during a tangent sweep.
    float *qtdT, *xdT, *qgd;
    float* qdT;                   // optional: discharge of every cell (setup%save_qsim_domain), T4 like qtT; null = off
    float *tape_hi, *tape_hp, *tape_hft, *tape_hst;
    float* ckpt_hi;               // gr-b / gr-c: hi at every SX_HIK-th step of the chunk, [Tc / SX_HIK][npad] (tape_hi: vic-a only)
    float* xT;                    // exchange series
    // staging rows of the chained groups ("Staging rows" below): [time block + stage][chained slot] float4, null = off.  cs0 = first slot
    // of the first chained group, ncs = slots of the chained groups
    float* qsk; int cs0, ncs;
    // gauges
    float *qg, *qgb;              // [ngc][nt] discharge at gauge cells / adjoint seeds
    int* cell_gauge;              // [npad] gauge-cell id or -1
    // schedule
    const int *g_slot_begin, *g_dmax;
    const int *s_cell, *s_stage, *s_sub, *s_wsub, *s_child, *s_ccount, *s_parent, *s_xout;   // sx_plan.h: components, sub-levels
    const int *x_prod, *x_cons;   // per exchange series: publishing group / group holding the inlet (-1: other tile)
    int* prog;                    // [ngroups + SX_PROG_EXTRA]: blocks published by each group in the running launch, then the stall flag and the
                                  // ticket counter (SX_PROG_STALL / _TICKET), then the stall diagnostics
    int ngroups;
    int spin_limit;               // polls before a waiting group gives up and raises the stall flag
    int mute_group;               // tests only (SMASHX_DEBUG_MUTE_GROUP): this group never publishes -> its consumers stall; -1 = none
    long long* gtime;             // diagnostics (SMASHX_TRACE_GROUPS=1): [2 passes][ngroups][start, end] wall_clock64 ticks, else null
};

// ------------------------------------------------------------------------------------------------
// Staging rows of the chained groups.  The groups of the rounds >= 1 are the river's main stems: a few workgroups whose super-step
// is a latency chain, and every memory instruction in it whose 64 lanes touch 64 different lines (slots of one wave sit at different
// stages, hence in different rows of the time-major arrays) costs that chain a full pass through the CU's memory path (DESIGN.md 12,
// anatomy of a super-step).  Round 2 cured the routing tape of that -- row = time block + stage, so a group's super-step touches one
// row.  qt / qt_b and the exchange series are shared with the vertical kernels and round 0, which want the plain rows, so the chained
// launches get a copy in that layout: qsk [time block + stage][chained slot].  The copy is a transposition, and it only pays if BOTH
// of its sides move whole rows (three earlier variants whose scattered side touched one 16-byte piece per row and lane ran at 0.85
// TB/s and cost more than the chained launches gained; a vertical kernel whose wavefronts store to 20 rows at once is four times
// slower: DESIGN.md 12).  Hence: one WAVEFRONT owns 64 consecutive slots of a group for all time blocks; per iteration it loads ONE
// plain row piece (the cells of consecutive slots are consecutive, and so are the series of consecutive inlets: sx_plan.cpp numbers
// the series in the order of the inlets that read them -- up to 1 KiB contiguous), pushes each lane's float4 into that
// lane's own FIFO in LDS -- as deep as the lane's stage is above the lowest stage of the 64 -- and stores ONE staging row piece from
// the FIFO heads (1 KiB contiguous).  A lane only ever touches its own FIFO: no barrier, no cross-lane traffic.  The reverse sweep
// runs the mirror image (GATHER = false): qt_b of the chained cells and the adjoint series of their inlets leave the staging rows
// for qtT / xT.  Series between two chained groups (produced and consumed inside the launch, behind the counters) keep their plain rows.
// Measured at 2048^2 (644 k chained slots x 548 time blocks, 11.3 GB per pass): 2.75 ms gather, 2.26 ms scatter (four wave-blocks per
// workgroup; 3.06 / 2.72 with one).  A plan takes the staging rows only when it has at least two chained
// groups per compute unit (smashx.hip): below that the chained launch is bound by the latency of its chain and gains less than the
// copies cost.
// Tables (host, smashx.hip): per wave-block the first slot, the slot count, the lowest stage and the stage spread; per chained slot
// the offset of its FIFO in the block's LDS for each direction.
// ------------------------------------------------------------------------------------------------
struct SxStageTables {
    const int* wb_slot0; const int* wb_n; const int* wb_smin; const int* wb_spread;   // per wave-block
    const int* fifo_g; const int* fifo_s;                                               // per chained slot: FIFO offset (float4) for gather / scatter
};
#ifndef SX_STG_PF
#define SX_STG_PF 16         // rows requested ahead of the one being pushed (the loop is a latency chain: its pace is the load latency / this)
#endif
// SX_STG_WAVES consecutive wave-blocks per workgroup: wavefronts that start together and advance at the same pace visit a row's
// neighbouring KiB together (DRAM pages, TLB entries: a row is 10 - 67 MB from the next)
#ifndef SX_STG_WAVES
#define SX_STG_WAVES 4
#endif
template <bool GATHER>
__global__ __launch_bounds__(64 * SX_STG_WAVES) void sx_k_chain_transpose(SxDeviceArrays A, SxStageTables S, int g0, int nb, int nblocks, int lds_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float4 sx_fifo_all[];
    const int wv = threadIdx.x >> 6;
    const int b = blockIdx.x * SX_STG_WAVES + wv, L = threadIdx.x & 63;
    if (b >= nblocks) return;
    float4* const sx_fifo = sx_fifo_all + (size_t)wv * (lds_per_wave / 16);
    const int nL = S.wb_n[b], smin = S.wb_smin[b], spread = S.wb_spread[b];
    const bool lane = L < nL;
    const int sl = S.wb_slot0[b] + (lane ? L : 0), cs = sl - A.cs0;
    const int c = lane ? A.s_cell[sl] : INT_MIN;
    // what this slot keeps in the plain arrays: its cell's row of qtT, or the row of a series that crosses the chain's boundary (handed up by
    // a round below the chain or by another rank; in the reverse sweep: handed back).  Series between chained groups and holes: nothing.
    float4* plain = nullptr; size_t pstride = 0;
    if (c >= 0) { plain = reinterpret_cast<float4*>(A.qtT) + c; pstride = (size_t)A.npad; }
    else if (c != INT_MIN && A.x_prod[-1 - c] < g0) { plain = reinterpret_cast<float4*>(A.xT) + (-1 - c); pstride = (size_t)A.nx; }
    const int d = lane ? A.s_stage[sl] - smin : 0;              // rows this slot's staging row runs ahead of the block's lowest
    const int delay = GATHER ? d : spread - d, depth = delay + 1;
    float4* fifo = sx_fifo + (lane ? (GATHER ? S.fifo_g[cs] : S.fifo_s[cs]) : 0);
    float4* stag = reinterpret_cast<float4*>(A.qsk) + cs;       // + row * ncs
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int niter = nb + spread;
    // GATHER: iteration i pushes plain row i and emits staging row i + smin (this lane's part of it is its block i - d);
    // else:   iteration i pushes staging row i + smin (this lane's block i - d) and emits plain row i - spread
    auto request = [&](int i) -> float4 {
        if (!lane) return zero4;
        if (GATHER) return (plain && i < nb) ? sx_gload4(plain + (size_t)i * pstride) : zero4;
        return (i - d >= 0 && i - d < nb) ? sx_gload4(stag + (size_t)(i + smin) * A.ncs) : zero4;
    };
    float4 pf[SX_STG_PF];
#pragma unroll
    for (int u = 0; u < SX_STG_PF; ++u) pf[u] = request(u);
    int wpos = 0, rpos = delay == 0 ? 0 : 1 % depth;            // the entry pushed `delay` iterations ago sits one past the write position
    for (int i0 = 0; i0 < niter; i0 += SX_STG_PF) {
#pragma unroll
        for (int u = 0; u < SX_STG_PF; ++u) {
            const int i = i0 + u;
            float4 v = pf[u];
            sx_pin(v);
            pf[u] = request(i + SX_STG_PF);
            if (i < niter && lane) {
                fifo[wpos] = v;
                const float4 o = fifo[rpos];                    // (delay 0: the entry just written)
                wpos = wpos + 1 == depth ? 0 : wpos + 1;
                rpos = rpos + 1 == depth ? 0 : rpos + 1;
                if (GATHER) {
                    const int tb = i - d;
                    sx_gstore4(stag + (size_t)(i + smin) * A.ncs, (tb >= 0 && tb < nb) ? o : zero4);
                } else {
                    const int tb = i - spread;
                    if (plain && tb >= 0) sx_gstore4(plain + (size_t)tb * pstride, o);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-cell invariants of the routing operators (md_routing_operator.f90:55-56,75; LINEAR_ROUTING_B
// forward_db.f90:6643-6648; UPSTREAM_DISCHARGE_B :6551)
// ------------------------------------------------------------------------------------------------
__global__ void sx_k_prep_routing(SxDeviceArrays A) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n) return;
    const float lr = A.lr[k];
    const float f = (float)(A.flwacc[k] - 1);
    A.rt_a[k] = sx_expf(-A.dt / (lr * 60.f));
    A.rt_f[k] = f;
    A.rt_denf[k] = 0.001f * A.dx * A.dx * f;
    A.rt_denb[k] = 0.001f * (A.dx * A.dx) * f;
}

// load / store of one float at (wave-uniform row pointer) + (per-lane byte offset < 2^31): buffer instruction with the
// row as its scalar resource, so the row advance is scalar-unit work and the vector unit sees no address arithmetic
// AUX = cache-policy bits (gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Measured at 1024^2 x 8760: nt on the forward kernel's
// stream (forcing in, tapes and qt out) 42.0 -> 41.1 ms; nt on the reverse kernel's loads 76.4 -> 78.7 ms, so only the
// forward kernel uses it.
#define SX_NT 2
// Still steps (prcp = pet = 0 on every lane of a wavefront: sx_ops.h) take the short form of the vertical step, forward and reverse.
// -DSX_STILL=0 compiles the general step only (A/B builds: tools/anatomy.sh).
#ifndef SX_STILL
#define SX_STILL 1
#endif
#define SX_HIK 8             // steps per block of the interception level's checkpoint / rebuild (chunk offsets are multiples of 16)
#ifndef SX_VADJ_NT
#define SX_VADJ_NT 0
#endif
template <int AUX = 0>
__device__ __forceinline__ float sx_row_load(const float* row, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void sx_row_store4(float* row, unsigned byte_off, float a, float b, float c, float d) {
    typedef int sx_v4i __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    sx_v4i v = {__builtin_bit_cast(int, a), __builtin_bit_cast(int, b), __builtin_bit_cast(int, c), __builtin_bit_cast(int, d)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void sx_row_store(float* row, unsigned byte_off, float v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, byte_off, 0, AUX);
}

template <int AUX = 0>
__device__ __forceinline__ unsigned sx_row_load_u16(const unsigned short* row, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, AUX);
}

// Forcing cursor of one marching thread.  request() issues the loads of a step and returns at once; hold() -- called one step
// later -- is where the wait lands (the words are pinned there, before the next request is issued), and prcp() / pet() decode the
// held words.  Layouts: COMPACT = false: fp32 rows; true: u16 rain count + the day's PET + the hour's ratio.  The hour of day is
// wave-uniform, so the day-change test is scalar work; the ratio is requested like a row, one step ahead (a load issued where it is
// used drags the prefetched rows of the next step into this step's wait -- measured: vert_fwd 42 -> 55 ms, vert_adj 76 -> 97 ms).
typedef const __attribute__((address_space(4))) float sx_cfloat;
template <bool COMPACT>
struct SxForcing {
    const SxDeviceArrays& A;
    const unsigned kb;          // byte offset of the cell in an fp32 row
    const size_t npad;
    int h;                      // hour index of the step the next request() is for
    const char* prow; const char* erow;      // rows of that step (wave-uniform running pointers: one scalar add per step and stream)
    unsigned p_n, p_w; float e_n, e_w, r_n, r_w;
    __device__ __forceinline__ SxForcing(const SxDeviceArrays& A_, unsigned kb_, int t_first)
        : A(A_), kb(kb_), npad((size_t)A_.npad), h(COMPACT ? (t_first + A_.hour0) % 24 : 0),
          prow(COMPACT ? (const char*)(A_.prcp16 + (size_t)t_first * A_.npad) : (const char*)(A_.prcp + (size_t)t_first * A_.npad)),
          erow(COMPACT ? nullptr : (const char*)(A_.pet + (size_t)t_first * A_.npad)),
          p_n(0u), p_w(0u), e_n(0.f), e_w(0.f), r_n(0.f), r_w(0.f) {}
    // t: absolute time step (consecutive calls move by one step, forwards or BACKwards); first: nothing is held yet
    template <int AUX, bool BACK> __device__ __forceinline__ void request(int t, bool first) {
        const long step = (long)npad * (COMPACT ? 2 : 4) * (BACK ? -1 : 1);
        if (!COMPACT) {
            p_n = __float_as_uint(sx_row_load<AUX>((const float*)prow, kb));
            e_n = sx_row_load<AUX>((const float*)erow, kb);
            prow += step; erow += step;
            return;
        }
        // the u16 count travels as the dword that holds it (lane pairs share one): nothing may touch the loaded word before hold(),
        // and a 16-bit load result is zero-extended by an ALU instruction right behind the load -- a wait in the same step
        // (measured: vert_adj 76 -> 89 ms).  The ratio comes the same way, one broadcast dword: a scalar load would be waited for
        // on the spot as well.
        p_n = __float_as_uint(sx_row_load<AUX>((const float*)prow, (kb >> 1) & ~3u));
        prow += step;
        if (first || h == (BACK ? 23 : 0)) e_n = sx_row_load<AUX>(A.petd + (size_t)((t + A.hour0) / 24) * npad, kb);   // a new day
        r_n = sx_row_load(A.pet_ratio + h, 0u);
        h = BACK ? (h == 0 ? 23 : h - 1) : (h == 23 ? 0 : h + 1);
    }
    __device__ __forceinline__ void hold() {
        p_w = p_n; e_w = e_n; r_w = r_n;
        sx_pinu(p_w); sx_pin1(e_w);
        if (COMPACT) sx_pin1(r_w);
    }
    __device__ __forceinline__ float prcp() const {
        if (!COMPACT) return __uint_as_float(p_w);
        const unsigned k = (p_w >> ((kb & 4u) << 2)) & 0xffffu;     // this lane's half of the dword (kb = 4 x cell)
        const float v = (float)k * A.prcp_c;
        return k == 65535u ? A.prcp_gap : v;
    }
    __device__ __forceinline__ float pet() const {
        if (!COMPACT) return e_w;
        return e_w < 0.f ? e_w : e_w * r_w;
    }
};

// both values of step t at once, layout decided at run time (kernels off the headline path: no prefetch)
__device__ __forceinline__ void sx_forcing_at(const SxDeviceArrays& A, int t, unsigned kb, float& p, float& e) {
    const size_t npad = (size_t)A.npad;
    if (A.prcp16 == nullptr) { p = sx_row_load(A.prcp + (size_t)t * npad, kb); e = sx_row_load(A.pet + (size_t)t * npad, kb); return; }
    const unsigned raw = sx_row_load_u16(A.prcp16 + (size_t)t * npad, kb >> 1);
    const float v = (float)raw * A.prcp_c;
    p = raw == 65535u ? A.prcp_gap : v;
    const int q = t + A.hour0;
    const float D = sx_row_load(A.petd + (size_t)(q / 24) * npad, kb);
    e = D < 0.f ? D : D * ((sx_cfloat*)A.pet_ratio)[q % 24];
}

// ------------------------------------------------------------------------------------------------
// vertical forward: one thread per cell marches the time chunk [t0, t0+T)
// ------------------------------------------------------------------------------------------------
// gr-b with the tape on and the compact forcing (the headline's forward kernel) comes out at 68 registers = 7 waves per SIMD; held to
// 64 (two values spilled, outside the time loop's critical path) it runs 8: 41.2 -> 39.7 ms at 1024^2 x 8760, 239.7 -> 232.2 ms per
// sweep at 2048^2.  The other instantiations are at or below 64 anyway, or (gr-c) would spill 6-10 registers: left alone.
#ifndef SX_VFWD_WAVES
#define SX_VFWD_WAVES 8
#endif
template <int ST, bool TAPE, bool CF>
#ifndef SX_VFWD_WAVES_GRC
#define SX_VFWD_WAVES_GRC 7       // gr-c, taped, compact: 74 registers = 6 waves as compiled, 72 without a spill when asked for 7
#endif
__global__ __launch_bounds__(SX_VBLOCK, (ST == 2 && CF) ? SX_VFWD_WAVES : (ST == 3 && TAPE && CF) ? SX_VFWD_WAVES_GRC : 1)
void sx_k_vert_fwd(SxDeviceArrays A, int t0, int T) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = A.k0 + blockIdx.x * SX_VBLOCK + threadIdx.x;   // the vertical kernels can be launched on a cell range
    if (k >= A.k1) return;
    const size_t npad = (size_t)A.npad;

    SxCellParams P;
    float hi = 0.f, hp = 0.f, hft = 0.f, hst = 0.f;
    P.ci = (ST == 2 || ST == 3) ? A.ci[k] : 1.f;
    P.cp = A.cp[k];
    P.cft = A.cft[k];
    P.cst = (ST == 3) ? A.cst[k] : 1.f;
    P.exc = (ST != 4) ? A.exc[k] : 0.f;
    sx_cell_params_init(P);
    P.cft_m4 = sx_pow_m4(P.cft);
    P.cst_m4 = (ST == 3) ? sx_pow_m4(P.cst) : 1.f;
    if (ST == 2 || ST == 3) hi = A.hi[k];
    hp = A.hp[k];
    hft = A.hft[k];
    if (ST == 3) hst = A.hst[k];

    // rows are wave-uniform (sx_row_load / sx_row_store): the vector unit does no address arithmetic in the time loop
    const unsigned kb = (unsigned)k * 4u;
    SxForcing<CF> F(A, kb, t0);
    if (T > 0) F.template request<SX_NT, false>(t0, true);
    for (int tq = 0; tq * 4 < T; ++tq) {
        float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = tq * 4 + i;
            if (tt < T) {
                F.hold();                            // the wait for this step's forcing goes here, before the next loads
                if (tt + 1 < T) F.template request<SX_NT, false>(t0 + tt + 1, false);
                const float prcp = F.prcp(), pet = F.pet();
                if (TAPE) {
                    const size_t o = (size_t)tt * npad;
                    if (ST == 2 || ST == 3) {     // the interception level: a full tape when it fits, else one checkpoint per SX_HIK steps
                        if (A.tape_hi) sx_row_store<SX_NT>(A.tape_hi + o, kb, hi);
                        else if ((tt % SX_HIK) == 0) sx_row_store<SX_NT>(A.ckpt_hi + (size_t)(tt / SX_HIK) * npad, kb, hi);
                    }
                    sx_row_store<SX_NT>(A.tape_hp + o, kb, hp);
                    sx_row_store<SX_NT>(A.tape_hft + o, kb, hft);
                    if (ST == 3) sx_row_store<SX_NT>(A.tape_hst + o, kb, hst);
                }
                const bool still = SX_STILL && sx_wave_all(sx_is_still<ST>(prcp, pet, hi, hp));      // wave-uniform
                q[i] = sx_vertical_step<ST>(P, prcp, pet, hi, hp, hft, hst, still);
            }
        }
        sx_row_store4<SX_NT>(A.qtT + (size_t)tq * npad * 4, kb * 4u, q[0], q[1], q[2], q[3]);
    }
    if (ST == 2 || ST == 3) A.hi[k] = hi;
    A.hp[k] = hp;
    A.hft[k] = hft;
    if (ST == 3) A.hst[k] = hst;
}

// ------------------------------------------------------------------------------------------------
// vic-a: vertical forward and adjoint (vic_a_forward md_forward_structure.f90:804-829, VIC_A_FORWARD_B
// forward_db.f90:10271-10316).  Same thread-per-cell time march, T4 qt, three tapes (husl1, husl2, hlsl).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ SxVicParams sx_vic_load(const SxDeviceArrays& A, int k) {
    SxVicParams P;
    P.b = A.ci[k]; P.cusl1 = A.cp[k]; P.cusl2 = A.cft[k]; P.clsl = A.cst[k]; P.ks = A.exc[k];
    P.ds = A.px[0][k]; P.dsm = A.px[1][k]; P.ws = A.px[2][k];
    sx_vic_derive(P);
    return P;
}
// Same shape as the GR kernels: wave-uniform rows through buffer descriptors, forcing one step ahead (SxForcing), streaming stores
// of the tapes, qt as one float4 per four steps; the reverse kernel prefetches the three taped levels and qt_b a step ahead.
#ifndef SX_VFWD_WAVES_VIC
#define SX_VFWD_WAVES_VIC 5       // taped (98-100 registers as compiled = 4 waves): asked for 5
#endif
template <bool TAPE, bool CF>
__global__ __launch_bounds__(SX_VBLOCK, TAPE ? SX_VFWD_WAVES_VIC : 1) void sx_k_vert_fwd_vic(SxDeviceArrays A, int t0, int T) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = A.k0 + blockIdx.x * SX_VBLOCK + threadIdx.x;   // the vertical kernels can be launched on a cell range
    if (k >= A.k1) return;
    const size_t npad = (size_t)A.npad;
    const SxVicParams P = sx_vic_load(A, k);
    const float cusl2_m4 = sx_pow_m4(P.cusl2);
    float husl1 = A.hi[k], husl2 = A.hp[k], hlsl = A.hft[k];
    const unsigned kb = (unsigned)k * 4u;
    SxForcing<CF> F(A, kb, t0);
    if (T > 0) F.template request<SX_NT, false>(t0, true);
    for (int tq = 0; tq * 4 < T; ++tq) {
        float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = tq * 4 + i;
            if (tt < T) {
                F.hold();
                if (tt + 1 < T) F.template request<SX_NT, false>(t0 + tt + 1, false);
                if (TAPE) {
                    const size_t o = (size_t)tt * npad;
                    sx_row_store<SX_NT>(A.tape_hi + o, kb, husl1);
                    sx_row_store<SX_NT>(A.tape_hp + o, kb, husl2);
                    sx_row_store<SX_NT>(A.tape_hft + o, kb, hlsl);
                }
                q[i] = sx_vic_step(P, cusl2_m4, F.prcp(), F.pet(), husl1, husl2, hlsl);
            }
        }
        sx_row_store4<SX_NT>(A.qtT + (size_t)tq * npad * 4, kb * 4u, q[0], q[1], q[2], q[3]);
    }
    A.hi[k] = husl1; A.hp[k] = husl2; A.hft[k] = hlsl;
}

// compiled for three waves per SIMD (168 registers, a handful spilled): 160 -> 140 ms against the natural 177 registers / two waves;
// four waves (128 registers) spill the fp64 polynomial constants of log2 / exp2 and take 236 ms
#ifndef SX_VADJ_WAVES_VIC
#define SX_VADJ_WAVES_VIC 3
#endif
template <bool CF>
__global__ __launch_bounds__(SX_VBLOCK, SX_VADJ_WAVES_VIC) void sx_k_vert_adj_vic(SxDeviceArrays A, int t0, int T) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = A.k0 + blockIdx.x * SX_VBLOCK + threadIdx.x;   // the vertical kernels can be launched on a cell range
    if (k >= A.k1) return;
    const size_t npad = (size_t)A.npad;
    const SxVicParams P = sx_vic_load(A, k);
    float cusl2_m4, cusl2_m5;
    sx_pow_m4_m5(P.cusl2, &cusl2_m4, &cusl2_m5);
    SxVicGrads G;
    G.b_b = A.ci_b[k]; G.cusl1_b = A.cp_b[k]; G.cusl2_b = A.cft_b[k]; G.clsl_b = A.cst_b[k]; G.ks_b = A.exc_b[k];
    G.ds_b = A.px_b[0][k]; G.dsm_b = A.px_b[1][k]; G.ws_b = A.px_b[2][k];
    G.husl1_b = A.hi_b[k]; G.husl2_b = A.hp_b[k]; G.hlsl_b = A.hft_b[k];
    const unsigned kb = (unsigned)k * 4u;
    SxForcing<CF> F(A, kb, t0 + T - 1);
    float n_h1 = 0.f, n_h2 = 0.f, n_hl = 0.f, n_q = 0.f;
    auto fetch = [&](int tt, bool first) {
        const size_t o = (size_t)tt * npad;
        F.template request<SX_VADJ_NT, true>(t0 + tt, first);
        n_h1 = sx_row_load<SX_VADJ_NT>(A.tape_hi + o, kb); n_h2 = sx_row_load<SX_VADJ_NT>(A.tape_hp + o, kb);
        n_hl = sx_row_load<SX_VADJ_NT>(A.tape_hft + o, kb);
        n_q = sx_row_load(A.qtT + (size_t)(tt >> 2) * npad * 4 + (tt & 3), kb * 4u);
    };
    if (T > 0) fetch(T - 1, true);
    for (int tt = T - 1; tt >= 0; --tt) {
        float h1 = n_h1, h2 = n_h2, hl = n_hl, q_b = n_q;
        F.hold();
        sx_pin1(h1); sx_pin1(h2); sx_pin1(hl); sx_pin1(q_b);
        if (tt > 0) fetch(tt - 1, false);
        sx_vic_step_b(P, cusl2_m4, cusl2_m5, F.prcp(), F.pet(), h1, h2, hl, q_b, G);
    }
    A.ci_b[k] = G.b_b; A.cp_b[k] = G.cusl1_b; A.cft_b[k] = G.cusl2_b; A.cst_b[k] = G.clsl_b; A.exc_b[k] = G.ks_b;
    A.px_b[0][k] = G.ds_b; A.px_b[1][k] = G.dsm_b; A.px_b[2][k] = G.ws_b;
    A.hi_b[k] = G.husl1_b; A.hp_b[k] = G.husl2_b; A.hft_b[k] = G.hlsl_b;
}

// vic-a with tangents (VIC_A_FORWARD_D): tangents of the parameters / levels live in the gradient arrays
__global__ __launch_bounds__(SX_VBLOCK) void sx_k_vert_fwd_vic_d(SxDeviceArrays A, int t0, int T) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = A.k0 + blockIdx.x * SX_VBLOCK + threadIdx.x;
    if (k >= A.k1) return;
    const size_t npad = (size_t)A.npad;
    const SxVicParams P = sx_vic_load(A, k);
    SxVicTan D;
    D.b_d = A.ci_b[k]; D.cusl1_d = A.cp_b[k]; D.cusl2_d = A.cft_b[k]; D.clsl_d = A.cst_b[k]; D.ks_d = A.exc_b[k];
    D.ds_d = A.px_b[0][k]; D.dsm_d = A.px_b[1][k]; D.ws_d = A.px_b[2][k];
    float cusl2_m4, cusl2_m5;
    sx_pow_m4_m5(P.cusl2, &cusl2_m4, &cusl2_m5);
    SxVD husl1 = sx_vd(A.hi[k], A.hi_b[k]), husl2 = sx_vd(A.hp[k], A.hp_b[k]), hlsl = sx_vd(A.hft[k], A.hft_b[k]);
    float* qt = A.qtT + (size_t)k * 4;
    float* qd = A.qtdT + (size_t)k * 4;
    for (int tt = 0; tt < T; ++tt) {
        float prcp, pet;
        sx_forcing_at(A, t0 + tt, (unsigned)k * 4u, prcp, pet);
        const SxVD r = sx_vic_step_d(P, D, cusl2_m4, cusl2_m5, prcp, pet, husl1, husl2, hlsl);
        const size_t o = (size_t)(tt >> 2) * npad * 4 + (tt & 3);
        qt[o] = r.v; qd[o] = r.d;
    }
    A.hi[k] = husl1.v; A.hi_b[k] = husl1.d; A.hp[k] = husl2.v; A.hp_b[k] = husl2.d; A.hft[k] = hlsl.v; A.hft_b[k] = hlsl.d;
}

// ------------------------------------------------------------------------------------------------
// vertical forward with tangents (inner body of GR_x_FORWARD_D, forward_db.f90:7748-9602): thread per cell; writes
// qt to qtT and qt_d to qtdT; the tangents of the states march along in the *_b arrays
// ------------------------------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(SX_VBLOCK) void sx_k_vert_fwd_d(SxDeviceArrays A, int t0, int T) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = A.k0 + blockIdx.x * SX_VBLOCK + threadIdx.x;   // the vertical kernels can be launched on a cell range
    if (k >= A.k1) return;
    const size_t npad = (size_t)A.npad;
    SxCellParams P;
    P.ci = (ST == 2 || ST == 3) ? A.ci[k] : 1.f;
    P.cp = A.cp[k];
    P.cft = A.cft[k];
    P.cst = (ST == 3) ? A.cst[k] : 1.f;
    P.exc = (ST != 4) ? A.exc[k] : 0.f;
    sx_cell_params_init(P);
    SxAdjParams Q;
    Q.cst_m5 = 1.f;
    sx_pow_m4_m5(P.cft, &P.cft_m4, &Q.cft_m5);
    P.cst_m4 = 1.f;
    if (ST == 3) sx_pow_m4_m5(P.cst, &P.cst_m4, &Q.cst_m5);
    Q.dcft2 = sx_mkdiv(P.cft * P.cft);
    Q.dcst2 = sx_mkdiv(P.cst * P.cst);
    Q.dcp2 = sx_mkdiv(P.cp * P.cp);
    SxTanParams D;
    D.ci_d = (ST == 2 || ST == 3) ? A.ci_b[k] : 0.f;
    D.cp_d = A.cp_b[k];
    D.cft_d = A.cft_b[k];
    D.cst_d = (ST == 3) ? A.cst_b[k] : 0.f;
    D.exc_d = (ST != 4) ? A.exc_b[k] : 0.f;
    SxDual hi = sx_mk(0.f, 0.f), hp, hft, hst = sx_mk(0.f, 0.f);
    if (ST == 2 || ST == 3) hi = sx_mk(A.hi[k], A.hi_b[k]);
    hp = sx_mk(A.hp[k], A.hp_b[k]);
    hft = sx_mk(A.hft[k], A.hft_b[k]);
    if (ST == 3) hst = sx_mk(A.hst[k], A.hst_b[k]);
    float4* qt4 = reinterpret_cast<float4*>(A.qtT) + k;
    float4* qd4 = reinterpret_cast<float4*>(A.qtdT) + k;
    for (int tq = 0; tq * 4 < T; ++tq) {
        float q[4] = {0.f, 0.f, 0.f, 0.f}, qd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = tq * 4 + i;
            if (tt < T) {
                float prcp, pet;
                sx_forcing_at(A, t0 + tt, (unsigned)k * 4u, prcp, pet);
                const SxDual r = sx_vertical_step_d<ST>(P, Q, D, prcp, pet, hi, hp, hft, hst);
                q[i] = r.v; qd[i] = r.d;
            }
        }
        qt4[(size_t)tq * npad] = make_float4(q[0], q[1], q[2], q[3]);
        qd4[(size_t)tq * npad] = make_float4(qd[0], qd[1], qd[2], qd[3]);
    }
    if (ST == 2 || ST == 3) { A.hi[k] = hi.v; A.hi_b[k] = hi.d; }
    A.hp[k] = hp.v; A.hp_b[k] = hp.d;
    A.hft[k] = hft.v; A.hft_b[k] = hft.d;
    if (ST == 3) { A.hst[k] = hst.v; A.hst_b[k] = hst.d; }
}

// ------------------------------------------------------------------------------------------------
// routing forward: one workgroup per routing group, time-skewed wavefront through LDS.
// Slot j at stage s handles time block (w - s) in super-step w; its children (stage s-1) published that
// block in super-step w-1.  upstream_discharge + linear_routing + the q update of
// md_forward_structure.f90:150-156, in the reference's operation order.
//
// Global memory is kept off the critical path: super-steps are grouped by SX_MU; the inputs of macro-step
// m+1 are requested at the start of macro-step m and the results of macro-step m are stored at the start
// of macro-step m+1 (before the next requests), so the one vmcnt wait per macro-step only ever meets
// operations that have been in flight for SX_MU super-steps.  The exchange between super-steps is LDS only.
// ------------------------------------------------------------------------------------------------
#ifndef SX_MU
#define SX_MU 4
#endif
#ifndef SX_MU_F
#define SX_MU_F SX_MU     // forward kernel
#endif
#ifndef SX_MU_A
#define SX_MU_A SX_MU     // reverse kernel
#endif
#ifndef SX_MAXGROUP
#define SX_MAXGROUP 512   // largest routing workgroup (group_size): 8 waves = 2 per SIMD
#endif
// Occupancy of the routing kernels: 132 (forward) / 154 (adjoint) registers = ONE resident group per CU.  Measured at
// 1024^2 x 8760: compiling them for two groups per CU (128 registers) makes round 0 slower (route_fwd 26.0 -> 27.7 ms; the
// adjoint spills, 28 -> 98 ms) and leaves the chained rounds unchanged (their time is (blocks + longest cell path) x the
// latency of one super-step, whoever is resident), so one group per CU stays.
// Issue priority of the routing waves (s_setprio 0..3).  Beside the vertical kernels -- four to six waves per SIMD that always have an
// instruction ready -- a routing wave at the default priority gets one issue slot in five or seven, and its super-step is a latency
// chain (LDS read, ~140 instructions, barrier): raised, it issues as if it were alone and costs the vertical waves only the slots it uses.
#ifndef SX_R_PRIO
#define SX_R_PRIO 0
#endif
#ifndef SX_RLB_F
#define SX_RLB_F 1        // waves per SIMD the chained forward kernel is compiled for (a resident group = 2 per SIMD)
#endif
#ifndef SX_RLB_A
#define SX_RLB_A 1        // adjoint
#endif

// TMODE 2: the same wavefront on the tangents (UPSTREAM_DISCHARGE_D, LINEAR_ROUTING_D and the q update of
// GR_x_FORWARD_D): reads qt_d (qtdT) and the hr_imd tape of the value pass, carries hlr_d (hlr_b), publishes q_d series
// in xdT, q_d at the gauge cells in qgd; TAPE must be false.
// TMODE 1: the value pass of a tangent sweep.  The Tapenade tangent code re-associates two primal expressions
// (qup = dt*(qup/temp), forward_db.f90:6464; q = temp*((qt + f*qrout)/dt), :8445-8448): forward_d's discharge differs
// from forward's in the last bit, and the criteria derivatives amplify that to ~5e-6 of cost_d, so the tangent sweep
// evaluates the primal the way forward_d does.
// draws the next ticket for the whole workgroup (chained launches); the barrier also separates two groups' use of the LDS rows
__device__ __forceinline__ int sx_next_ticket(const SxDeviceArrays& A) {
    __shared__ int s_ticket;
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add((sx_gint*)SX_PROG_TICKET(A), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return s_ticket;
}

template <bool TAPE, bool CHAIN, int TMODE>
__device__ __forceinline__ void sx_route_fwd_group(const SxDeviceArrays& A, const int g, const int g0, const int gend, const int t0, const int T) {
    constexpr bool TAN = (TMODE == 2);
    constexpr int MU = SX_MU_F, PK = SX_PK * 4 / MU;     // super-steps per macro-step; macro-steps between two publications
    constexpr bool DFORM = (TMODE == 1);
    extern __shared__ __attribute__((aligned(16))) float4 sx_lds[];   // [2][blockDim.x]
    const int sb = A.g_slot_begin[g], m = A.g_slot_begin[g + 1] - sb, dmax = A.g_dmax[g];
    const int j = threadIdx.x, M = blockDim.x;
    bool valid = j < m;
    const int nb = (T + SX_BT - 1) / SX_BT;
    if (A.gtime && j == 0) A.gtime[2 * g] = wall_clock64();

    int cell = -1, stage = 0, sub = 0, ccount = 0, xout = -1, xin = -1, gid = -1;
    unsigned ch01 = ~0u, ch23 = ~0u, ch45 = ~0u, ch67 = ~0u;      // children: 16 bits each, slot | 0x8000 = same component (sx_plan.h)
    float a = 0.f, f = 0.f, den = 1.f, hlr = 0.f, ad = 0.f;
    bool hasup = false;
    // sub-levels of this wavefront's components (wave-uniform): the number of passes through a super-step's body
    const int nsubw = __builtin_amdgcn_readfirstlane(m > 0 ? A.s_wsub[sb + min(j, m - 1)] : 1);
    if (valid && A.s_cell[sb + j] == INT_MIN) valid = false;      // a hole of a partly filled wavefront
    if (valid) {
        const int c = A.s_cell[sb + j];
        stage = A.s_stage[sb + j];
        sub = A.s_sub[sb + j];
        if (c >= 0) {
            cell = c;
            const int4 cw = reinterpret_cast<const int4*>(A.s_child)[sb + j];
            ch01 = (unsigned)cw.x; ch23 = (unsigned)cw.y; ch45 = (unsigned)cw.z; ch67 = (unsigned)cw.w;
            ccount = A.s_ccount[sb + j]; xout = A.s_xout[sb + j];
            a = A.rt_a[c]; f = A.rt_f[c]; den = (TAN || DFORM) ? A.rt_denb[c] : A.rt_denf[c]; hlr = TAN ? A.hlr_b[c] : A.hlr[c];
            if (TAN) { const float lrv = A.lr[c]; ad = a * ((A.dt / (60.f * lrv)) * A.lr_b[c] / lrv); }
            hasup = A.flwacc[c] > 1;
            gid = A.cell_gauge[c];
        } else {
            xin = -1 - c;
        }
    }
    const float dt = A.dt, dx = A.dx;
    const SxDiv dden = sx_mkdiv(den), ddt = sx_mkdiv(dt);
    // chained rounds: an inlet whose series is published inside this launch follows its producer's counter
    const int* wprog = nullptr;
    int seen = 0;
    if (CHAIN && xin >= 0) { const int pg = A.x_prod[xin]; if (pg >= g0 && pg < gend) wprog = A.prog + pg; }
    // chained rounds: everything else a slot reads waits in the staging rows ("Staging rows" above), row = time block + stage
    const bool staged = CHAIN && !TAN && A.qsk != nullptr && valid && !wprog;
    // T4 addressing: element (tb, id) of an array with `stride` float4 per time block
    const float4* src = staged ? reinterpret_cast<const float4*>(A.qsk) + (size_t)stage * A.ncs + (sb + j - A.cs0)
                      : (cell >= 0) ? reinterpret_cast<const float4*>(TAN ? A.qtdT : A.qtT) + cell
                                    : reinterpret_cast<const float4*>(TAN ? A.xdT : A.xT) + (xin >= 0 ? xin : 0);
    const size_t sstride = staged ? (size_t)A.ncs : (cell >= 0) ? (size_t)A.npad : (size_t)A.nx;
    float4* x4 = reinterpret_cast<float4*>(TAN ? A.xdT : A.xT);
    float4* hr4 = reinterpret_cast<float4*>(A.hrT);
    // hr_imd tape, private to the routing kernels: row = time block + stage, so the slots of a group -- which work on time block
    // w - stage in super-step w -- all write row w, cell next to cell (measured: 64 -> 8 line requests per wave store)
    const int hs = A.hr_skew ? stage : 0;
    float* gauge_out = TAN ? A.qgd : A.qg;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int tb) -> float4 { return cell >= 0 ? sx_gload4s(src + (size_t)tb * sstride) : sx_gload4(src + (size_t)tb * sstride); };

    float4 nxt[MU], outq[MU], outh[MU], nhr[MU];
    if (CHAIN && wprog) sx_wait_prog(wprog, min(MU - stage, nb), seen, SX_PROG_STALL(A), A.spin_limit, g);
#pragma unroll
    for (int u = 0; u < MU; ++u) {
        const int tb = u - stage;
        nxt[u] = (valid && tb >= 0 && tb < nb) ? fetch(tb) : zero4;
        nhr[u] = (TAN && valid && cell >= 0 && tb >= 0 && tb < nb) ? sx_gload4(hr4 + (size_t)(tb + hs) * A.npad + cell) : zero4;
        outq[u] = zero4; outh[u] = zero4;
    }
    const int nsuper = nb + dmax;
    const int nmacro = (nsuper + MU - 1) / MU;
    for (int mw = 0; mw <= nmacro; ++mw) {
        float4 cur[MU], chr[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) { cur[u] = nxt[u]; sx_pin(cur[u]); if (TAN) { chr[u] = nhr[u]; sx_pin(chr[u]); } }
        // chained: the stores of the previous macro-step (four super-steps old) have completed past this point
        if (CHAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // results of the previous macro-step leave now
        if (mw > 0 && cell >= 0) {
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int tb = MU * (mw - 1) + u - stage;
                if (tb >= 0 && tb < nb) {
                    if (xout >= 0) x4[(size_t)tb * A.nx + xout] = outq[u];
                    if (TAPE) sx_gstore4s(hr4 + (size_t)(tb + hs) * A.npad + cell, outh[u]);
                    if (A.qdT) reinterpret_cast<float4*>(A.qdT)[(size_t)tb * A.npad + cell] = outq[u];
                    if (gid >= 0) {
                        const float qv[4] = {outq[u].x, outq[u].y, outq[u].z, outq[u].w};
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i)
                            if (tb * SX_BT + i < T) gauge_out[(size_t)gid * A.nt + t0 + tb * SX_BT + i] = qv[i];
                    }
                }
            }
        }
        if (mw == nmacro) break;
        // inputs of the next macro-step are requested now
        if (CHAIN && wprog) sx_wait_prog(wprog, min(MU * (mw + 2) - stage, nb), seen, SX_PROG_STALL(A), A.spin_limit, g);
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int tb = MU * (mw + 1) + u - stage;
            nxt[u] = (valid && tb >= 0 && tb < nb) ? fetch(tb) : zero4;
            if (TAN) nhr[u] = (valid && cell >= 0 && tb >= 0 && tb < nb) ? sx_gload4(hr4 + (size_t)(tb + hs) * A.npad + cell) : zero4;
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int w = MU * mw + u;
            const int tb = w - stage;
            const bool act = valid && tb >= 0 && tb < nb;
            float4* pub = sx_lds + (size_t)(w & 1) * M;
            // a child's block: published in THIS super-step by a lower sub-level of the same component (same wavefront: row w & 1),
            // or in the previous super-step by the root of a component one stage below (row (w + 1) & 1)
            auto child_val = [&](unsigned e) -> float4 {
                const float4* row = sx_lds + (size_t)((w + 1 + (int)((e >> 15) & 1u)) & 1) * M;
                return row[min((int)(e & 0x7fffu), M - 1)];
            };
          auto sub_pass = [&](const int sl) {
            if (act && sub == sl) {
                const int tl = tb * SX_BT;           // first step of the block, chunk-local
                if (cell >= 0) {
                    // upstream sum in D8 order; the first two children (almost every cell has <= 2) are fetched
                    // together so their LDS latencies overlap; "+ 0" for an absent child is exact
                    float s[SX_BT];
                    {
                        const float4 v0 = child_val(ch01 & 0xffffu), v1 = child_val(ch01 >> 16);
                        const bool h0 = ccount > 0, h1 = ccount > 1;
                        s[0] = (h0 ? v0.x : 0.f) + (h1 ? v1.x : 0.f); s[1] = (h0 ? v0.y : 0.f) + (h1 ? v1.y : 0.f);
                        s[2] = (h0 ? v0.z : 0.f) + (h1 ? v1.z : 0.f); s[3] = (h0 ? v0.w : 0.f) + (h1 ? v1.w : 0.f);
                    }
                    for (int c = 2; c < ccount; ++c) {
                        const unsigned wd = c < 4 ? ch23 : c < 6 ? ch45 : ch67;
                        const float4 v = child_val((wd >> ((c & 1) * 16)) & 0xffffu);
                        s[0] = s[0] + v.x; s[1] = s[1] + v.y; s[2] = s[2] + v.z; s[3] = s[3] + v.w;
                    }
                    const float qt[SX_BT] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
                    float q[SX_BT], hr[SX_BT], qup[SX_BT], qro[SX_BT];
                    // three phases so that only the two-operation store recurrence is serial: the divisions of the
                    // four steps are independent and overlap (a lone wave pays the full latency of every dependent op)
                    {
                        float nq[SX_BT], dq[SX_BT];
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) nq[i] = DFORM ? s[i] : s[i] * dt;
                        sx_div4(dq, nq, dden);
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) qup[i] = hasup ? (DFORM ? dt * dq[i] : dq[i]) : 0.f;
                    }
                    if (!TAN) {
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) {
                            // steps beyond T (only in the last block of a chunk) are computed and discarded
                            const bool live = tl + i < T;
                            const float hr_imd = hlr + qup[i];
                            const float hnew = hr_imd * a;
                            qro[i] = hr_imd - hnew;
                            hr[i] = live ? hr_imd : 0.f;
                            hlr = live ? hnew : hlr;
                        }
                        float nq[SX_BT], dq[SX_BT];
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) nq[i] = DFORM ? qt[i] + f * qro[i] : (qt[i] + qro[i] * f) * dx * dx * 0.001f;
                        sx_div4(dq, nq, ddt);
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) {
                            const float v = DFORM ? (0.001f * (dx * dx)) * dq[i] : dq[i];
                            q[i] = (tl + i < T) ? v : 0.f;
                        }
                    } else {
                        // hlr carries hr_d; hrv = hr_imd of the value pass:  hr_d = a hr_imd_d + hr_imd (a arg1_d)
                        const float hrv[SX_BT] = {chr[u].x, chr[u].y, chr[u].z, chr[u].w};
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) {
                            const bool live = tl + i < T;
                            const float hr_imd_d = hlr + qup[i];
                            const float hnew_d = a * hr_imd_d + hrv[i] * ad;
                            qro[i] = hr_imd_d - hnew_d;
                            hr[i] = 0.f;
                            hlr = live ? hnew_d : hlr;
                        }
#pragma unroll
                        for (int i = 0; i < SX_BT; ++i) {
                            const float v = sx_div((0.001f * (dx * dx)) * (qt[i] + f * qro[i]), ddt);
                            q[i] = (tl + i < T) ? v : 0.f;
                        }
                    }
                    const float4 q4 = make_float4(q[0], q[1], q[2], q[3]);
                    pub[j] = q4;
                    outq[u] = q4;
                    outh[u] = make_float4(hr[0], hr[1], hr[2], hr[3]);
                } else {
                    pub[j] = cur[u];
                }
            }
          };
            // one level per component (the default schedule): the body once, no loop (route_adj 27.6 -> 26.5 ms at 1024^2 x 8760).  Else
            // the next sub-level of this wavefront reads what this one has just published: LDS operations of one wave complete in
            // order, the fence keeps the compiler from moving the reads up
            if (nsubw == 1) sub_pass(0);
            else for (int sl = 0; sl < nsubw; ++sl) { sub_pass(sl); if (sl + 1 < nsubw) sx_lds_wave_fence(); }
            sx_lds_barrier();
            // every wave has passed this macro-step's vmcnt(0): what the roots (stage dmax) stored one macro-step
            // ago -- blocks below MU (mw-1) - dmax -- has reached L2 and can be released
            if (CHAIN && u == 0 && j == 0 && mw % PK == 0) { const int done = min(MU * (mw - 1) - dmax, nb); if (done > 0 && g != A.mute_group) sx_publish(A.prog + g, done); }
        }
    }
    if (valid && cell >= 0) { if (TAN) A.hlr_b[cell] = hlr; else A.hlr[cell] = hlr; }
    if (A.gtime && j == 0) A.gtime[2 * g + 1] = wall_clock64();
    if (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j == 0 && g != A.mute_group) sx_publish(A.prog + g, nb);
    }
}

template <bool TAPE, bool CHAIN, int TMODE = 0>
__global__ __launch_bounds__(SX_MAXGROUP, SX_RLB_F) void sx_k_route_fwd(SxDeviceArrays A, int g0, int gend, int t0, int T) {
    if (SX_R_PRIO) __builtin_amdgcn_s_setprio(SX_R_PRIO);
    if (!CHAIN) { sx_route_fwd_group<TAPE, false, TMODE>(A, g0 + (int)blockIdx.x, g0, gend, t0, T); return; }
    for (;;) {
        const int ticket = sx_next_ticket(A);
        if (ticket >= gend - g0) break;
        sx_route_fwd_group<TAPE, CHAIN, TMODE>(A, g0 + ticket, g0, gend, t0, T);
    }
}

// ------------------------------------------------------------------------------------------------
// routing adjoint: same groups, roots first, time descending.  Reverse of the q update, LINEAR_ROUTING_B
// (forward_db.f90:6628-6652) and UPSTREAM_DISCHARGE_B (:6520-6564), as in GR_x_FORWARD_B :8649-8672.
// Reads hrT (hr_imd of the recomputed forward) and the gauge seeds; writes qt_b over qtT.
// Same macro-step staging of global memory as the forward kernel.
// ------------------------------------------------------------------------------------------------
template <bool CHAIN>
__device__ __forceinline__ void sx_route_adj_group(const SxDeviceArrays& A, const int g, const int g0, const int gend, const int t0, const int T) {
    constexpr int MU = SX_MU_A, PK = SX_PK * 4 / MU;
    extern __shared__ __attribute__((aligned(16))) float4 sx_lds[];
    const int sb = A.g_slot_begin[g], m = A.g_slot_begin[g + 1] - sb, dmax = A.g_dmax[g];
    const int j = threadIdx.x, M = blockDim.x;
    bool valid = j < m;
    const int nb = (T + SX_BT - 1) / SX_BT;
    if (A.gtime && j == 0) A.gtime[2 * (A.ngroups + g)] = wall_clock64();

    int cell = -1, rstage = 0, sub = 0, par = -1, xout = -1, xin = -1, gid = -1;
    bool psame = false;                 // the parent sits in the same component (same wavefront, next sub-level up)
    float a = 0.f, f = 0.f, den = 1.f, lr = 1.f, hr_b = 0.f, lr_b = 0.f;
    bool hasup = false;
    const int nsubw = __builtin_amdgcn_readfirstlane(m > 0 ? A.s_wsub[sb + min(j, m - 1)] : 1);
    if (valid && A.s_cell[sb + j] == INT_MIN) valid = false;      // a hole of a partly filled wavefront
    if (valid) {
        const int c = A.s_cell[sb + j];
        rstage = dmax - A.s_stage[sb + j];
        sub = A.s_sub[sb + j];
        par = A.s_parent[sb + j];
        if (par >= 0) { psame = (par & 0x40000000) != 0; par &= 0xffff; }
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
    const SxDiv dden = sx_mkdiv(den), ddt = sx_mkdiv(dt), dlr = sx_mkdiv((lr * lr) * 60.f);
    float4* x4 = reinterpret_cast<float4*>(A.xT);
    const float4* hr4p = reinterpret_cast<const float4*>(A.hrT);
    const int hs = A.hr_skew ? dmax - rstage : 0;     // the forward kernel's row shift: in reverse super-step w a group reads row nb - 1 - w + dmax
    float4* qt4 = reinterpret_cast<float4*>(A.qtT);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool root_in = (cell >= 0 && par < 0 && xout >= 0);   // subtree root fed by an exchange series
    // where a slot's result goes: qt_b of its cell, or -- inlet -- the adjoint series of the subtree upstream.  ONE store instruction
    // for both kinds (per-lane base and row stride): in the chained rounds, where a third of the slots are inlets, the second store
    // instruction of a super-step cost as much as the first whatever its lane count (anatomy of the reverse launch)
    // chained rounds: qt_b of the cells and the adjoint series that leave the chain go to the staging rows (row = time block + stage: one
    // row per reverse super-step of the group; sx_k_chain_transpose<false> puts them in place) -- not the adjoint series of an inlet whose
    // subtree upstream is routed inside this launch
    const bool staged = CHAIN && A.qsk != nullptr && valid && (cell >= 0 || (xin >= 0 && !(A.x_prod[xin] >= g0 && A.x_prod[xin] < gend)));
    float4* const dst = staged ? reinterpret_cast<float4*>(A.qsk) + (size_t)(dmax - rstage) * A.ncs + (sb + j - A.cs0)
                               : (cell >= 0) ? qt4 + cell : x4 + (xin >= 0 ? xin : 0);
    const size_t dstride = staged ? (size_t)A.ncs : (cell >= 0) ? (size_t)A.npad : (size_t)A.nx;

    // gauge cells also fetch their adjoint seeds (qsim_b summed per cell) with the staged loads, so the
    // super-step loop itself contains no global memory operation and no vmcnt wait
    const float* seedp = (gid >= 0) ? A.qgb + (size_t)gid * A.nt + t0 : nullptr;
    auto load_seed = [&](int tb) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gid >= 0) {
            const int tl = tb * SX_BT;
            if (tl + 0 < T) v.x = sx_gload1(seedp + tl + 0);
            if (tl + 1 < T) v.y = sx_gload1(seedp + tl + 1);
            if (tl + 2 < T) v.z = sx_gload1(seedp + tl + 2);
            if (tl + 3 < T) v.w = sx_gload1(seedp + tl + 3);
        }
        return v;
    };
    // chained rounds: a subtree root follows the counter of the group that holds its inlet (a later round)
    const int* wprog = nullptr;
    int seen = 0;
    if (CHAIN && root_in) { const int cg = A.x_cons[xout]; if (cg >= g0 && cg < gend) wprog = A.prog + cg; }
    auto fetch_in = [&](int tb) -> float4 { return sx_gload4(x4 + (size_t)tb * A.nx + xout); };
    float4 nhr[MU], nin[MU], nsd[MU], outq[MU];
    if (CHAIN && wprog) sx_wait_prog(wprog, min(MU - rstage, nb), seen, SX_PROG_STALL(A), A.spin_limit, g);
#pragma unroll
    for (int u = 0; u < MU; ++u) {
        const int tbr = u - rstage;
        const int tb = nb - 1 - tbr;
        const bool ok = valid && tbr >= 0 && tbr < nb;
        nhr[u] = (ok && cell >= 0) ? sx_gload4s(hr4p + (size_t)(tb + hs) * A.npad + cell) : zero4;
        nin[u] = (ok && root_in) ? fetch_in(tb) : zero4;
        nsd[u] = ok ? load_seed(tb) : zero4;
        outq[u] = zero4;
    }
    const int nsuper = nb + dmax;
    const int nmacro = (nsuper + MU - 1) / MU;
    for (int mw = 0; mw <= nmacro; ++mw) {
        float4 chr[MU], cin[MU], csd[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            chr[u] = nhr[u]; cin[u] = nin[u]; csd[u] = nsd[u];
            sx_pin(chr[u]); sx_pin(cin[u]); sx_pin(csd[u]);
        }
        if (CHAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mw > 0 && valid) {
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int tbr = MU * (mw - 1) + u - rstage;
                if (tbr >= 0 && tbr < nb) {
                    const int tb = nb - 1 - tbr;
                    // one store instruction for both kinds of slot; the inlets' destination is an exchange row: never nontemporal
                    sx_gstore4(dst + (size_t)tb * dstride, outq[u]);
                }
            }
        }
        if (mw == nmacro) break;
        if (CHAIN && wprog) sx_wait_prog(wprog, min(MU * (mw + 2) - rstage, nb), seen, SX_PROG_STALL(A), A.spin_limit, g);
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int tbr = MU * (mw + 1) + u - rstage;
            const int tb = nb - 1 - tbr;
            const bool ok = valid && tbr >= 0 && tbr < nb;
            nhr[u] = (ok && cell >= 0) ? sx_gload4s(hr4p + (size_t)(tb + hs) * A.npad + cell) : zero4;
            nin[u] = (ok && root_in) ? fetch_in(tb) : zero4;
            nsd[u] = ok ? load_seed(tb) : zero4;
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int w = MU * mw + u;
            const int tbr = w - rstage;
            const bool act = valid && tbr >= 0 && tbr < nb;
            float4* pub = sx_lds + (size_t)(w & 1) * M;
            const float4* prev = sx_lds + (size_t)((w + 1) & 1) * M;
          auto sub_pass = [&](const int sl) {
            if (act && sub == sl) {
                const int tb = nb - 1 - tbr;
                const int tl = tb * SX_BT;
                float4 in4 = cin[u];                 // contribution of the downstream cell
                if (par >= 0) in4 = psame ? pub[par] : prev[par];
                if (cell >= 0) {
                    const float hrv[SX_BT] = {chr[u].x, chr[u].y, chr[u].z, chr[u].w};
                    const float inv[SX_BT] = {in4.x, in4.y, in4.z, in4.w};
                    const float sdv[SX_BT] = {csd[u].x, csd[u].y, csd[u].z, csd[u].w};
                    float pb[SX_BT], qtb[SX_BT];
                    float tmpb[SX_BT], qrb[SX_BT], himb[SX_BT], a1b[SX_BT];
                    float nq[SX_BT];
#pragma unroll
                    for (int i = SX_BT - 1; i >= 0; --i) {          // independent of the carried adjoint state
                        float q_b = 0.f;
                        if (gid >= 0) q_b = q_b + sdv[i];
                        q_b = q_b + inv[i];
                        nq[i] = (dx * dx) * 0.001f * q_b;
                    }
                    sx_div4(tmpb, nq, ddt);
#pragma unroll
                    for (int i = SX_BT - 1; i >= 0; --i) qrb[i] = f * tmpb[i];
#pragma unroll
                    for (int i = SX_BT - 1; i >= 0; --i) {          // the serial part: hr_b recurrence (3 operations per step)
                        const bool live = tl + i < T;
                        const float hb1 = hr_b - qrb[i];
                        himb[i] = qrb[i] + a * hb1;
                        a1b[i] = a * hrv[i] * hb1;
                        hr_b = live ? himb[i] : hr_b;
                    }
                    float na[SX_BT], nh[SX_BT], da[SX_BT], dh[SX_BT];
#pragma unroll
                    for (int i = 0; i < SX_BT; ++i) { na[i] = dt * a1b[i]; nh[i] = dt * himb[i]; }
                    sx_div4(da, na, dlr);
                    sx_div4(dh, nh, dden);
#pragma unroll
                    for (int i = SX_BT - 1; i >= 0; --i) {          // lr_b keeps its reverse-time summation order
                        const bool live = tl + i < T;
                        const float lnew = lr_b + da[i];
                        lr_b = live ? lnew : lr_b;
                        pb[i] = (live && hasup) ? dh[i] : 0.f;
                        qtb[i] = live ? tmpb[i] : 0.f;
                    }
                    pub[j] = make_float4(pb[0], pb[1], pb[2], pb[3]);
                    outq[u] = make_float4(qtb[0], qtb[1], qtb[2], qtb[3]);
                } else {
                    // inlet pseudo-cell: hand the receiver's contribution to the subtree rooted upstream
                    outq[u] = in4;
                }
            }
          };
            if (nsubw == 1) sub_pass(0);
            else for (int sl = nsubw - 1; sl >= 0; --sl) { sub_pass(sl); if (sl > 0) sx_lds_wave_fence(); }     // the reverse of the forward order: a component's top first
            sx_lds_barrier();
            // inlet slots sit at reverse stage <= dmax: reverse blocks below MU (mw-1) - dmax are complete
            if (CHAIN && u == 0 && j == 0 && mw % PK == 0) { const int done = min(MU * (mw - 1) - dmax, nb); if (done > 0 && g != A.mute_group) sx_publish(A.prog + g, done); }
        }
    }
    if (valid && cell >= 0) { A.hlr_b[cell] = hr_b; A.lr_b[cell] = lr_b; }
    if (A.gtime && j == 0) A.gtime[2 * (A.ngroups + g) + 1] = wall_clock64();
    if (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j == 0 && g != A.mute_group) sx_publish(A.prog + g, nb);
    }
}

template <bool CHAIN>
__global__ __launch_bounds__(SX_MAXGROUP, SX_RLB_A) void sx_k_route_adj(SxDeviceArrays A, int g0, int gend, int t0, int T) {
    if (SX_R_PRIO) __builtin_amdgcn_s_setprio(SX_R_PRIO);
    if (!CHAIN) { sx_route_adj_group<false>(A, g0 + (int)blockIdx.x, g0, gend, t0, T); return; }
    // chained rounds run roots-of-the-basin first: tickets walk the groups downwards
    for (;;) {
        const int ticket = sx_next_ticket(A);
        if (ticket >= gend - g0) break;
        sx_route_adj_group<CHAIN>(A, gend - 1 - ticket, g0, gend, t0, T);
    }
}

// ------------------------------------------------------------------------------------------------
// vertical adjoint: one thread per cell marches the chunk backwards, consuming qt_b (in qtT) and the
// taped pre-step levels; parameter gradients accumulate per cell in reverse time order like
// parameters_b%x(row,col) does in the reference (forward_db.f90:8699-8702).
// ------------------------------------------------------------------------------------------------
// gr-b on the compact forcing (the headline's reverse kernel): 112 registers = 4 waves per SIMD; held to 96 = 5 waves it spills 12
// values (52 B of scratch per lane) and still wins: 68.3 -> 66.6 ms at 1024^2 x 8760, 258.2 -> 249.8 ms per sweep at 2048^2 (round 2
// had measured 69.0 -> 67.7 and declined; with the larger grid's 64 waves per SIMD the fifth wave is worth 3 %).  Bit-identical.
#ifndef SX_VADJ_WAVES
#define SX_VADJ_WAVES 5
#endif
template <int ST, bool CF>
#ifndef SX_VADJ_WAVES_GRC
#if SX_EXACT_LIBM
#define SX_VADJ_WAVES_GRC 1       // (the exact-libm build keeps what the compiler chooses)
#else
#define SX_VADJ_WAVES_GRC 5       // gr-c, compact: 122 registers = 4 waves as compiled; held to 96 = 5 waves (about 20 values spilled):
#endif                            // vert_adj 80.2 -> 79.1 ms at 1024^2 x 8760, twice on one box (round 3, last session; bit-identical)
#endif
__global__ __launch_bounds__(SX_VBLOCK, (ST == 2 && CF) ? SX_VADJ_WAVES : (ST == 3 && CF) ? SX_VADJ_WAVES_GRC : 1)
void sx_k_vert_adj(SxDeviceArrays A, int t0, int T) {
    SX_LIBM_INIT();      // exact-libm build: the tables of expf / logf / powf into LDS (sx_libm.h); nothing otherwise
    const int k = A.k0 + blockIdx.x * SX_VBLOCK + threadIdx.x;   // the vertical kernels can be launched on a cell range
    if (k >= A.k1) return;
    const size_t npad = (size_t)A.npad;
    SxCellParams P;
    P.ci = (ST == 2 || ST == 3) ? A.ci[k] : 1.f;
    P.cp = A.cp[k];
    P.cft = A.cft[k];
    P.cst = (ST == 3) ? A.cst[k] : 1.f;
    P.exc = (ST != 4) ? A.exc[k] : 0.f;
    sx_cell_params_init(P);
    SxAdjParams Q;
    Q.cst_m5 = 1.f;
    sx_pow_m4_m5(P.cft, &P.cft_m4, &Q.cft_m5);
    P.cst_m4 = 1.f;
    if (ST == 3) sx_pow_m4_m5(P.cst, &P.cst_m4, &Q.cst_m5);
    Q.dcft2 = sx_mkdiv(P.cft * P.cft);
    Q.dcst2 = sx_mkdiv(P.cst * P.cst);
    Q.dcp2 = sx_mkdiv(P.cp * P.cp);
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
    // one-step-ahead software prefetch of everything a step reads (forcing, taped levels, qt_b)
    // addresses = wave-uniform row (buffer descriptor in scalar registers, advanced by the scalar unit) + the cell's 32-bit
    // byte offset: no vector address arithmetic in the time loop
    const unsigned kb = (unsigned)k * 4u;
    SxForcing<CF> F(A, kb, t0 + T - 1);
    float n_hi = 0.f, n_hp = 0.f, n_hft = 0.f, n_hst = 0.f, n_q = 0.f;
    const bool hi_taped = (ST == 2 || ST == 3) && A.tape_hi != nullptr;       // wave-uniform
    // tape rows of the step the next fetch is for: running pointers, stepped back one row per fetch
    const size_t olast = (size_t)(T > 0 ? T - 1 : 0) * npad;
    const float* r_hi = hi_taped ? A.tape_hi + olast : nullptr;
    const float* r_hp = A.tape_hp + olast;
    const float* r_hft = A.tape_hft + olast;
    const float* r_hst = (ST == 3) ? A.tape_hst + olast : nullptr;
    auto fetch = [&](int tt, bool first) {
        F.template request<SX_VADJ_NT, true>(t0 + tt, first);
        if (hi_taped) { n_hi = sx_row_load<SX_VADJ_NT>(r_hi, kb); r_hi -= npad; }
        n_hp = sx_row_load<SX_VADJ_NT>(r_hp, kb); n_hft = sx_row_load<SX_VADJ_NT>(r_hft, kb);
        r_hp -= npad; r_hft -= npad;
        if (ST == 3) { n_hst = sx_row_load<SX_VADJ_NT>(r_hst, kb); r_hst -= npad; }
        n_q = sx_row_load(A.qtT + (size_t)(tt >> 2) * npad * 4 + (tt & 3), kb * 4u);
    };
    // When the interception level is not taped (it depends on the forcing and ci only; the plan drops its tape when that is what
    // lets the whole period fit): each block of SX_HIK steps is marched forward once more from its checkpoint -- the same
    // sx_interception on the same operands, hence the same bits -- and the pre-step levels wait in this thread's LDS
    // column for the reverse steps of the block.
    __shared__ float s_hi[(ST == 2 || ST == 3) ? SX_HIK : 1][SX_VBLOCK];
    if (T > 0) fetch(T - 1, true);
    for (int tb = (T - 1) / SX_HIK; tb >= 0; --tb) {
        const int tt0 = tb * SX_HIK, len = min(SX_HIK, T - tt0);
        if ((ST == 2 || ST == 3) && !hi_taped) {
            float fp[SX_HIK], fe[SX_HIK];
#pragma unroll
            for (int j = 0; j < SX_HIK; ++j) sx_forcing_at(A, t0 + tt0 + (j < len ? j : 0), kb, fp[j], fe[j]);
            float h = sx_row_load(A.ckpt_hi + (size_t)tb * npad, kb);
#pragma unroll
            for (int j = 0; j < SX_HIK; ++j) {
                s_hi[j][threadIdx.x] = h;
                if (j < len && fp[j] >= 0.f && fe[j] >= 0.f) { float pn, ei; sx_interception(fp[j], fe[j], P.ci, P.dci, h, pn, ei); }
            }
        }
        for (int tt = tt0 + len - 1; tt >= tt0; --tt) {
            float hit = n_hi, hp = n_hp, hft = n_hft, hst = n_hst, q = n_q;
            F.hold();
            sx_pin1(hp); sx_pin1(hft); sx_pin1(q);
            if (ST == 2 || ST == 3) sx_pin1(hit);
            if (ST == 3) sx_pin1(hst);
            if (tt > 0) fetch(tt - 1, false);
            const float prcp = F.prcp(), pet = F.pet();
            const float hi = (ST == 2 || ST == 3) ? (hi_taped ? hit : s_hi[tt - tt0][threadIdx.x]) : 0.f;
            const bool still = SX_STILL && sx_wave_all(sx_is_still<ST>(prcp, pet, hi, hp));      // wave-uniform
            sx_vertical_step_b<ST>(P, Q, prcp, pet, hi, hp, hft, hst, q, G, still);
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
