/* smashx.h -- C ABI of libsmashx: the MI355X-native forward + adjoint solver that stands in for the
 * reference's differentiated core.
 *
 * What it replaces.  The reference's wrapped boundary is
 *     mw_forward::forward    smash/solver/forward/mw_forward.f90:18-39
 *     mw_forward::forward_b  smash/solver/forward/mw_forward.f90:41-68
 * which forward, through an implicit interface, to the external procedures
 *     base_forward    smash/solver/forward/forward.f90:1-80
 *     base_forward_b  smash/solver/forward/forward_db.f90:10648-10936
 * A drop-in links objects defining base_forward_/base_forward_b_ that marshal the derived types'
 * allocatable components into the calls below (fortran/smashx_dropin.f90).
 *
 * Conventions.  Plain pointers and sizes only.  Host arrays are column-major exactly as the
 * reference holds them (row index fastest); gauge_pos and path are 0-based here (the Fortran shim
 * subtracts 1), optimize_start_step stays 1-based like setup%optimize%optimize_start_step.  The
 * caller owns every array; the plan owns all device memory.  Every function returns 0 on success
 * or a negative SMASHX_E_* code (the reference has no error channel: the Fortran shim prints
 * smashx_last_error() and stops with `error stop`, the Python mirror raises SmashxError).  There is
 * no CPU fallback: without a usable HIP device plan creation fails.
 */
#ifndef SMASHX_H
#define SMASHX_H

#ifdef __cplusplus
extern "C" {
#endif

#define SMASHX_GNP 16 /* md_constant.f90:32  ci cp beta cft cst alpha exc b cusl1 cusl2 clsl ks ds dsm ws lr */
#define SMASHX_GNS 8  /* md_constant.f90:33  hi hp hft hst husl1 husl2 hlsl hlr */

enum { SMASHX_GR_A = 1, SMASHX_GR_B = 2, SMASHX_GR_C = 3, SMASHX_GR_D = 4, SMASHX_VIC_A = 5 };   /* setup%structure, forward.f90:43-65 */
enum { SMASHX_NSE = 1, SMASHX_KGE = 2, SMASHX_KGE2 = 3, SMASHX_SE = 4, SMASHX_RMSE = 5, SMASHX_LOGARITHMIC = 6 }; /* mwd_cost.f90:98-126 */
enum { SMASHX_PRIOR = 1, SMASHX_SMOOTHING = 2, SMASHX_HARD_SMOOTHING = 3 };   /* mwd_cost.f90:199-224 */
enum { SMASHX_P_CI = 0, SMASHX_P_CP = 1, SMASHX_P_BETA = 2, SMASHX_P_CFT = 3, SMASHX_P_CST = 4, SMASHX_P_ALPHA = 5,
       SMASHX_P_EXC = 6, SMASHX_P_B = 7, SMASHX_P_CUSL1 = 8, SMASHX_P_CUSL2 = 9, SMASHX_P_CLSL = 10, SMASHX_P_KS = 11,
       SMASHX_P_DS = 12, SMASHX_P_DSM = 13, SMASHX_P_WS = 14, SMASHX_P_LR = 15 };
enum { SMASHX_S_HI = 0, SMASHX_S_HP = 1, SMASHX_S_HFT = 2, SMASHX_S_HST = 3, SMASHX_S_HUSL1 = 4, SMASHX_S_HUSL2 = 5,
       SMASHX_S_HLSL = 6, SMASHX_S_HLR = 7 };

enum {
    SMASHX_OK = 0,
    SMASHX_E_ARG = -1,          /* bad argument / inconsistent sizes */
    SMASHX_E_UNSUPPORTED = -2,  /* option outside the hot path built so far */
    SMASHX_E_HIP = -3,          /* a HIP runtime call failed (smashx_last_error() has the text) */
    SMASHX_E_NODEVICE = -4,     /* no usable gfx950 device: there is no CPU fallback */
    SMASHX_E_MESH = -5,         /* flow directions do not form a forest over the active cells */
    SMASHX_E_STATE = -6         /* call order (forcing / options not set) */
};

/* SetupDT + MeshDT scalars the path reads (mwd_setup.f90:108-157, mwd_mesh.f90:45-72) */
typedef struct {
    int structure;     /* SMASHX_GR_* / SMASHX_VIC_A */
    int nrow, ncol;
    int nt;            /* setup%ntime_step */
    int ng;
    float dt;          /* s */
    float dx;          /* m */
    int chunk_steps;   /* storage chunk: steps kept on the tape at once (recompute unit of the checkpointed adjoint); 0 = from free HBM */
    int pipe_steps;    /* pipeline sub-chunk: vertical and routing kernels overlap chunk by chunk on two streams; 0 = no sub-chunking */
    int group_size;    /* routing workgroup size (cells + inlets per group); 0 = default */
    int device;        /* HIP device ordinal; -1 = current device */
    int tile[4];       /* multi-GPU: this rank owns rows [tile[0],tile[1]) x cols [tile[2],tile[3]) of the grid;
                          all zero = the whole grid.  Mesh and parameter arrays stay global-sized. */
} smashx_config;

typedef struct {
    const int* flwdir;       /* (nrow,ncol) D8 codes 1..8, <=0 nodata            mwd_mesh.f90:57 */
    const int* flwacc;       /* (nrow,ncol) cells draining through, self included mwd_mesh.f90:58 */
    const int* active_cell;  /* (nrow,ncol) 1 = active                            mwd_mesh.f90:60 */
    const int* path;         /* (2,nrow*ncol) 0-based visiting order; only needed for the sparse forcing layout; may be NULL */
    const int* gauge_pos;    /* (ng,2) 0-based (row, col)                         mwd_mesh.f90:63 */
    const float* area;       /* (ng) m^2                                          mwd_mesh.f90:65 */
    const int* owner_mask;   /* multi-GPU, optional: (nrow,ncol) 1 = this rank owns the cell.  Overrides smashx_config.tile:
                              * any partition whose rank graph is acyclic works, e.g. sub-catchments cut at confluences
                              * (smash_amd.tiles.partition_subcatchments).  NULL: the tile rectangle / the whole grid. */
} smashx_mesh;

/* Optimize_SetupDT fields the path reads (mwd_setup.f90:57-105) */
typedef struct {
    int denormalize_forward;
    int optimize_start_step;        /* 1-based */
    int njf;
    int jobs_fun[8];                /* SMASHX_NSE ... */
    float wjobs_fun[8];
    int njr;
    int jreg_fun[4];                /* SMASHX_PRIOR ... */
    float wjreg_fun[4];
    float wjreg;
    int optim_parameters[SMASHX_GNP];
    int optim_states[SMASHX_GNS];
    float lb_parameters[SMASHX_GNP], ub_parameters[SMASHX_GNP];
    float lb_states[SMASHX_GNS], ub_states[SMASHX_GNS];
    const float* wgauge;            /* (ng) */
} smashx_options;

/* ParametersDT / StatesDT as 16 / 8 pointers to (nrow,ncol) host arrays in md_constant order;
 * fields the structure does not use may be NULL (smash/core/_constant.py:15-29). */
typedef struct { float* f[SMASHX_GNP]; } smashx_parameters;
typedef struct { float* f[SMASHX_GNS]; } smashx_states;

typedef struct { float cost, cost_jobs, cost_jreg; } smashx_costs;   /* output%cost*, mwd_cost.f90:300-304 */

/* device-side time of the last sweep, measured with HIP events on the plan's stream */
typedef struct {
    float sweep_ms;          /* whole device sweep (forward, or forward + cost + adjoint) */
    float vert_fwd_ms;       /* vertical (production/transfer) forward kernels            */
    float route_fwd_ms;      /* routing forward kernels                                   */
    float cost_ms;
    float route_adj_ms;
    float vert_adj_ms;
    int   vert_fwd_launches, route_fwd_launches, route_adj_launches, vert_adj_launches;
    int   n_chunks, chunk_steps, pipe_steps, n_rounds, n_groups;
    double device_bytes;     /* HBM held by the plan */
    double cellsteps[4];     /* cell-steps the launches of the last sweep processed: vert_fwd, route_fwd, route_adj, vert_adj
                                (a checkpointed adjoint runs the forward kernels over most of the period twice) */
    float route_fwd_chained_ms, route_adj_chained_ms;   /* the part of route_fwd_ms / route_adj_ms spent in the chained launches (rounds >= 1) */
    int   route_fwd_chained_launches, route_adj_chained_launches;
    int   max_stage;         /* stages of the deepest routing group (the fill of a routing launch, in super-steps) */
    int   n_chained_groups;  /* groups of the chained rounds */
    int   chain_staged;      /* 1: the chained launches read and write staging rows through the copy passes (their time is part of
                                route_*_chained_ms), 0: the plain rows */
} smashx_timing;

typedef struct smashx_plan smashx_plan;

const char* smashx_last_error(void);
int smashx_device_count(void);
/* ABI guard for bindings that mirror the structs by hand (the Fortran shim, ctypes): sizes in bytes of
 * {smashx_config, smashx_mesh, smashx_options, smashx_parameters, smashx_states, smashx_costs, smashx_timing};
 * returns SMASHX_ABI_VERSION. */
#define SMASHX_ABI_VERSION 8
int smashx_abi_sizes(int sizes[7]);

/* builds the routing schedule from the mesh and allocates device storage */
int smashx_plan_create(const smashx_config* cfg, const smashx_mesh* mesh, smashx_plan** out);
int smashx_plan_destroy(smashx_plan* plan);

/* number of active cells and the device cell order: k -> (row, col), 0-based */
int smashx_plan_ncells(const smashx_plan* plan);
int smashx_plan_cell_order(const smashx_plan* plan, int* rows, int* cols);

/* forcing: Input_DataDT%prcp/pet (nrow,ncol,nt) [sparse = 0] or %sparse_prcp/pet (nac,nt) numbered along
 * path over active cells [sparse = 1] (mwd_input_data.f90:32-50, mw_sparse_storage.f90:12-49); nac counts the WHOLE grid's
 * active cells, also on a tiled plan (which reads the rows of its own cells at their whole-grid positions).
 * Stays resident in HBM across sweeps. */
int smashx_set_forcing(smashx_plan* plan, const float* prcp, const float* pet, int sparse);
/* device-resident block: d_prcp/d_pet are DEVICE pointers to (t1-t0, ncells) arrays, cell index in
 * plan order (smashx_plan_cell_order) fastest.  Used by bench.py to build the forcing in HBM.
 * The forcing counts as set once blocks have covered every step of [0, nt).  fp32 rows: blocks may arrive in any order and a
 * block may be refreshed later (coverage is only forgotten when the layout is reset).  Compact layout: a block with t0 = 0 opens
 * a NEW upload cycle (the daily PET field and the gap value of the previous data set are dropped, and so is the coverage), the
 * other blocks of the cycle may follow in any order. */
int smashx_set_forcing_device_block(smashx_plan* plan, int t0, int t1, const float* d_prcp, const float* d_pet);
/* Lossless compact residency of the forcing.  The reference's reader forms every value it stores from far fewer bits:
 *   prcp(cell, t) = real(k) * prcp_conversion_factor, k the raster's integer depth (0.1 mm units in its datasets), or -99
 *                   for a missing file            smash/core/_read_input_data.py:176-196, mwd_setup.f90:128
 *   pet(cell, t)  = daily(cell, day) * ratio(hour) with the 24-entry RATIO_PET_HOURLY table when
 *                   setup%daily_interannual_pet   _read_input_data.py:223-283, smash/core/_constant.py:47-75
 * With compact = 1 the plan stores k as uint16 (65535 = the gap marker) and the daily PET field, and the kernels form the fp32
 * values with the reader's own single fp32 multiply: 2.17 B per cell-step resident instead of 8 (a year of hourly forcing
 * for 2048^2 cells: 80 GB instead of 294 GB).  Every value handed in is checked BIT FOR BIT against what the kernels will
 * reconstruct; data that is not of this form is never approximated: smashx_set_forcing falls back to fp32 rows by itself,
 * smashx_set_forcing_device_block returns SMASHX_E_UNSUPPORTED for the offending block (the caller resets the layout and
 * sends the blocks again).  Requires dt = 3600 s for the PET form.  Call before the forcing is set; smashx_forcing_info
 * reports what the plan holds (compact 0/1, resident bytes per cell-step). */
typedef struct {
    int compact;           /* 0 = fp32 rows (default), 1 = compact, verified */
    float prcp_factor;     /* setup%prcp_conversion_factor */
    float pet_ratio[24];   /* hourly share of the daily PET; hour index = (time step + pet_hour0) mod 24 */
    int pet_hour0;         /* ratio index of time step 0 (the reference's first step is start_time + dt: 1 for a run starting at midnight) */
} smashx_forcing_layout;
int smashx_set_forcing_layout(smashx_plan* plan, const smashx_forcing_layout* layout);
int smashx_forcing_info(const smashx_plan* plan, int* compact, double* bytes_per_cellstep);
int smashx_set_qobs(smashx_plan* plan, const float* qobs /* (ng,nt) */);
int smashx_set_options(smashx_plan* plan, const smashx_options* opt);

/* base_forward (forward.f90:1-80): qsim (ng,nt); fstates = output%fstates; params/states are inout
 * like the reference (denormalised on return when denormalize_forward; states restored). */
int smashx_forward(smashx_plan* plan, smashx_parameters* params, const smashx_parameters* params_bgd,
                   smashx_states* states, const smashx_states* states_bgd, float* qsim, smashx_costs* costs,
                   smashx_states* fstates /* nullable */);

/* base_forward_b (forward_db.f90:10648-10936): params_b/states_b are fully overwritten with
 * d cost / d (normalised when denormalize_forward) parameters and initial states, times cost_b. */
int smashx_forward_b(smashx_plan* plan, smashx_parameters* params, const smashx_parameters* params_bgd,
                     smashx_states* states, const smashx_states* states_bgd, float cost_b, float* qsim,
                     smashx_costs* costs, smashx_parameters* params_b, smashx_states* states_b);

/* split-phase form of the two calls above, for measurement with inputs resident in HBM and for calibration loops:
 * upload -> (sweep)* -> download.  adjoint = 0: forward sweep; 1: forward + cost + adjoint sweep.
 * After a first complete upload, a NULL field in params / states means "unchanged": only the fields the optimiser moved
 * travel (var_to_control / control_to_var of mw_optimize.f90:679-777 touch only the optimised fields).  download skips NULL
 * fields likewise; downloading parameters / states under denormalize_forward hands the caller denormalised fields and a
 * complete upload is required again. */
int smashx_upload(smashx_plan* plan, const smashx_parameters* params, const smashx_parameters* params_bgd,
                  const smashx_states* states, const smashx_states* states_bgd);
int smashx_sweep(smashx_plan* plan, int adjoint, float cost_b);
int smashx_download(smashx_plan* plan, int adjoint, smashx_parameters* params, smashx_states* states, float* qsim,
                    smashx_costs* costs, smashx_states* fstates, smashx_parameters* params_b, smashx_states* states_b);
int smashx_get_timing(const smashx_plan* plan, smashx_timing* out);

/* Control vector of the variational calibration on the device (mw_optimize.f90:679-777: var_to_control_lbfgsb /
 * control_to_var_lbfgsb).  The control vector holds the fields flagged in optim_parameters / optim_states
 * (smashx_set_options), parameters first, md_constant order, each over the active cells with the column index outer and the row
 * index inner, in fp64 -- exactly what the reference hands to lbfgsb.f.  Fields are in the optimiser's space: normalised when
 * denormalize_forward is set (the calibration's mode of operation).
 *   smashx_control_size      length n of the control vector (0 when nothing is flagged)
 *   smashx_control_set       control_to_var: x -> the flagged device fields (denormalised on the device like smashx_upload does);
 *                            every other field keeps the device copy of the last smashx_upload, which must have happened once
 *   smashx_control_get       var_to_control of the fields the device holds
 *   smashx_control_gradient  after an adjoint sweep: d cost / d x, i.e. parameters_b / states_b of the flagged fields packed
 *                            the same way (what mw_optimize.f90:606 builds from forward_b's output)
 * x / g are host arrays of n doubles; packing, casts and (de)normalisation run on the device, one contiguous copy crosses PCIe. */
int smashx_control_size(smashx_plan* plan);
int smashx_control_set(smashx_plan* plan, const double* x);
int smashx_control_get(smashx_plan* plan, double* x);
int smashx_control_gradient(smashx_plan* plan, double* g);
/* Optional whole-domain stores of the forward run, OutputDT%qsim_domain / net_prcp_domain (mwd_output.f90:43-47,
 * written at md_forward_structure.f90:158-194 when setup%save_qsim_domain / save_net_prcp_domain): caller-owned host
 * arrays (nrow, ncol, nt) column-major -- inactive cells are set to -99 like OutputDT_initialise does -- or, with
 * sparse != 0, the (nac, nt) sparse_ forms.  They are filled by every following smashx_forward / forward sweep
 * until reset with NULL; adjoint sweeps do not touch them.  A tiled plan fills the cells of its own part and leaves -99 everywhere
 * else (the caller overlays the parts), in either form: nac of the sparse form is the whole grid's. */
int smashx_set_domain_outputs(smashx_plan* plan, float* qsim_domain, float* net_prcp_domain, int sparse);

/* base_forward_d (forward_db.f90:10517-10601), the tangent-linear model behind mw_forward::forward_d
 * (mw_forward.f90:70-97): directional derivative of the discharge and of the cost along (params_d, states_d), which
 * are given in the same space as params / states (normalised when denormalize_forward).  qsim_d (ng, nt) may be NULL.
 * params / states / qsim / costs come back as from smashx_forward.  parameters_bgd_d / states_bgd_d of the reference
 * are passive and have no counterpart.  Deviation: the reference's forward_d ends with an unconditional
 * denormalisation (forward_db.f90:3246-3247) that corrupts parameters and states when denormalize_forward is off;
 * here they are left as smashx_forward leaves them, and params_d / states_d are not modified. */
int smashx_forward_d(smashx_plan* plan, smashx_parameters* params, const smashx_parameters* params_d,
                     const smashx_parameters* params_bgd, smashx_states* states, const smashx_states* states_d,
                     const smashx_states* states_bgd, float* qsim, float* qsim_d, smashx_costs* costs, float* cost_d);
/* The two terms of the last smashx_forward_d's cost_d = jobs_d + wjreg x jreg_d (COMPUTE_COST_D, forward_db.f90:3248).  On a tiled
 * plan smashx_forward_d is collective like the sweeps (the boundary series of the value pass, then of the tangent pass, travel through
 * the plan's exchange, a message per pipeline sub-chunk); jobs_d then covers the gauges of this part and jreg_d the whole grid, so the
 * decomposition's cost_d is the sum of the parts' jobs_d + wjreg x jreg_d of any one part. */
int smashx_tangent_terms(const smashx_plan* plan, float* jobs_d, float* jreg_d);

/* ---- multi-GPU tiles: discharge series that cross the tile boundary -------------------
 * A cell whose D8 receiver lies in another tile publishes its discharge series ("out" edge); a cell of another
 * tile draining into this one is an "in" edge.  Both lists are sorted by the flat (row + col*nrow) index of
 * the SOURCE cell, so the two sides of a tile border enumerate a shared edge set in the same order.
 * The plan packs / unpacks the series of one pipeline sub-chunk into the caller's device buffers and calls
 * `fn` at the four points where data must move between ranks; the host (torch.distributed over RCCL in
 * bench.py) does the send / recv.  Buffers: out_buf holds n_out * pipe_steps floats, in_buf n_in * pipe_steps.
 *   phase 0  FWD_RECV  fill in_buf  (series of the in edges for this sub-chunk)   before routing forward
 *   phase 1  FWD_SEND  out_buf is ready (series of the out edges)                 after  routing forward
 *   phase 2  ADJ_RECV  fill out_buf (adjoint contributions for the out edges)     before routing adjoint
 *   phase 3  ADJ_SEND  in_buf is ready (adjoint series of the in edges)           after  routing adjoint
 * Layout of a buffer: [edge][ceil(nsteps/4)] float4 (4 consecutive steps each).
 * A checkpointed adjoint sweep (several storage chunks) runs the forward of every chunk but the last twice; the second run -- the
 * recomputation inside the reverse sweep -- moves NOTHING between ranks (round 4): every plan keeps the series it received for the
 * chunk in the first pass (n_in x chunk_steps floats per chunk) and sends none.  Phases 0 and 1 are therefore called once per
 * sub-chunk and sweep, phases 2 and 3 once; the same holds for the native exchange below. */
typedef int (*smashx_halo_fn)(void* user, int phase, int t0, int nsteps);
/* host-only (no GPU needed): builds the routing schedule of one tile and reports
 * info = {cells, rounds, groups, slots, exchange series, deepest stage, n_out, n_in}; edge arrays may be NULL or
 * must hold `cap` entries (flat row + col*nrow indices, sorted by source cell). */
int smashx_tile_probe(const smashx_config* cfg, const smashx_mesh* mesh, int* info, int* out_src, int* out_dst,
                      int* in_src, int* in_dst, int cap);
int smashx_halo_counts(const smashx_plan* plan, int* n_out, int* n_in);
int smashx_halo_edges(const smashx_plan* plan, int* out_src, int* out_dst, int* in_src, int* in_dst);
int smashx_plan_chunking(smashx_plan* plan, int* chunk_steps, int* pipe_steps);   /* fixes and returns the chunk lengths */
/* HBM accounting of a plan: out = {free bytes on the device at the moment the storage-chunk length was chosen (-1 before then),
 * total bytes of the device, bytes the plan holds now}.  With chunk_steps = 0 the chunk length -- and with it the number of forward
 * passes of a checkpointed adjoint -- follows from the first number: a card with less free memory (another process, a larger RCCL
 * buffer pool) plans more, shorter chunks; bench.py prints it beside hbm_plan_gb. */
int smashx_plan_hbm(const smashx_plan* plan, double out[3]);
int smashx_set_halo(smashx_plan* plan, float* d_out_buf, float* d_in_buf, smashx_halo_fn fn, void* user);

/* Cost terms that span the tiles of a decomposition (round 3; reference smash/solver/optimize/mwd_cost.f90:139-154, 159-245).
 *  - Regularisation (compute_jreg) needs nothing new: every plan holds whole (nrow, ncol) parameter / state planes, so every rank
 *    evaluates the reference's ordered sums over the WHOLE grid -- bit-identical to the single domain -- and takes the gradient of
 *    its own cells.  smashx_costs.cost_jreg is therefore the same on every rank and smashx_costs.cost contains wjreg * cost_jreg on
 *    every rank: the cost of the decomposition is  sum over ranks of cost_jobs  +  wjreg * cost_jreg (once).
 *  - The median over the negative-weight gauges (wgauge < 0, mwd_cost.f90:139-154 + quantile1d_r :675-723) needs the gauge_jobs of
 *    gauges on other ranks.  slot_of_gauge[g] = position of local gauge g among ALL negative-weight gauges of the decomposition in
 *    global gauge order (-1 for the others), nslots = their number (the same on every rank, > 0 even on a rank that owns none).
 *    Between the two phases of the cost kernel the nslots values are summed over the ranks: by ncclAllReduce on the routing stream
 *    when smashx_set_exchange is active, else by reduce_fn (host: sums n floats in place over the tiles; called once per sweep by
 *    every tile).  Each rank's cost_jobs then carries its own gauges' share of the median (the interpolation weights of its slots),
 *    so the sum over ranks is the reference's jobs.  nslots = 0 switches it off.  Call before smashx_set_options. */
typedef int (*smashx_reduce_fn)(void* user, float* values, int n);
int smashx_set_median_slots(smashx_plan* plan, int nslots, const int* slot_of_gauge, smashx_reduce_fn reduce_fn, void* user);

/* ---- native exchange: grouped ncclSend / ncclRecv on the plan's routing stream ----------------------
 * The reference has no counterpart (its only parallel code is the OpenMP replica loop, mw_multiple_run.f90:96-117).
 * One RCCL communicator per process (= per GPU); rank 0 draws the id, every rank calls smashx_comm_create with it (the
 * launcher moves the 128 bytes: bench.py broadcasts them through torch.distributed).  RCCL is resolved at run time
 * (dlopen of librccl.so.1: the copy the host process already holds, e.g. PyTorch's, else ROCm's), so single-GPU users
 * never load it.  With an exchange set, a sweep moves the boundary series of every pipeline sub-chunk itself:
 *   forward   recv(in edges, grouped by upstream peer) -> unpack -> routing -> pack -> send(out edges, by downstream peer)
 *   reverse   recv(out edges)                          -> unpack -> routing adjoint -> pack -> send(in edges)
 * all stream-ordered on the routing stream: no host synchronisation, no callback.  out_peer[n_out] / in_peer[n_in] give
 * the rank that owns the other end of each boundary edge (edge order of smashx_halo_edges).  Every rank of the
 * decomposition must cut time identically: smashx_set_exchange checks (all-reduce of chunk_steps / pipe_steps / nt) and
 * fails with SMASHX_E_ARG on disagreement; a plan with boundary series refuses chunk_steps = 0 (sized from each rank's
 * own free HBM) with SMASHX_E_ARG.  smashx_comm_allreduce_sum: sum over ranks of n doubles (the global cost =
 * sum of the per-tile partial costs), host in / host out. */
#define SMASHX_COMM_ID_BYTES 128
int smashx_comm_unique_id(unsigned char id[SMASHX_COMM_ID_BYTES]);
int smashx_comm_create(const unsigned char id[SMASHX_COMM_ID_BYTES], int rank, int nranks, int device, void** comm);
int smashx_comm_destroy(void* comm);
int smashx_comm_allreduce_sum(void* comm, double* values, int n);
/* diagnostics: the ranks the communicator spans (ncclCommCount) and RCCL's version code (ncclGetVersion); either pointer may be NULL */
int smashx_comm_info(void* comm, int* nranks, int* version);
int smashx_set_exchange(smashx_plan* plan, void* comm, const int* out_peer, const int* in_peer);

/* ---- hyper mappings (host side; round 4) -----------------------------------------------------------------------------------
 * mw_forward::hyper_forward / hyper_forward_b / hyper_forward_d (mw_forward.f90:99-181 -> base_hyper_forward forward.f90:82-157,
 * BASE_HYPER_FORWARD_B / _D forward_db.f90:11231-11560, 11079-11162) are forward / forward_b / forward_d with one step in front of the
 * time loop -- every parameter (state) field is a sigmoid of a linear or polynomial form of nd catchment descriptors
 * (hyper_parameters_to_parameters mwd_parameters_manipulation.f90:304-362, hyper_states_to_states mwd_states_manipulation.f90:270-329)
 * -- and that step's tangent / adjoint (forward_db.f90:1313-1537, 2179-2369) around the sweep.  The map is host code in the reference
 * and host code here (sx_hyper.cpp: no GPU, the caller's planes, fp32 in the reference's operation order); a host composes
 *     smashx_hyper_map_forward (parameters, states)  ->  smashx_forward                      (base_hyper_forward)
 *     ... -> smashx_forward_b -> smashx_hyper_map_b (parameters_b -> hyper_parameters_b, ...)  (base_hyper_forward_b)
 *     smashx_hyper_map_d -> smashx_forward_d                                                 (base_hyper_forward_d)
 * with denormalize_forward off and no regulariser (hyper_compute_cost knows neither), states left at their final values
 * (forward.f90:150).  smash_amd.hyper_forward / _b / _d do exactly that; under the Fortran shim the reference's own routines do. */
#define SMASHX_HYPER_LINEAR 1
#define SMASHX_HYPER_POLYNOMIAL 2
typedef struct {
    int mapping;               /* SMASHX_HYPER_LINEAR / _POLYNOMIAL (setup%optimize%mapping) */
    int nrow, ncol, nd;        /* grid, number of descriptors (setup%nd) */
    int nfields;               /* 16 parameter fields or 8 state fields, md_constant order */
    const float* descriptor;   /* input_data%descriptor (nrow, ncol, nd), column-major */
    const float* lb;           /* setup%optimize%lb_parameters / lb_states (nfields) */
    const float* ub;
} smashx_hyper_map;
/* rows of a hyper matrix: 1 + nd (linear), 1 + 2 nd (polynomial) = setup%optimize%nhyper; hyper matrices are (nhyper, nfields)
 * column-major: column i = Hyper_ParametersDT field i (nhyper, 1) */
int smashx_hyper_nhyper(const smashx_hyper_map* map);
int smashx_hyper_map_forward(const smashx_hyper_map* map, const float* hyper, float* const* planes /* nfields x (nrow,ncol), NULL skipped */);
int smashx_hyper_map_d(const smashx_hyper_map* map, const float* hyper, const float* hyper_d, float* const* planes, float* const* planes_d);
int smashx_hyper_map_b(const smashx_hyper_map* map, const float* hyper, float* const* planes_b /* NULL = zero */, float* hyper_b /* overwritten */);

/* ---- diagnostics ---------------------------------------------------------------------------------------
 * Start / end ticks (100 MHz device wall clock) of every routing group in the last forward (pass 0) and adjoint
 * (pass 1) routing launches: out[2][groups][2]; round_of_group[groups] may be NULL.  Only recorded when the plan
 * was created with SMASHX_TRACE_GROUPS=1 in the environment (otherwise SMASHX_E_STATE).  The environment variable
 * SMASHX_CHAIN_ROUNDS=0 restores one routing launch per round (default: all rounds chained in one launch). */
int smashx_debug_group_times(smashx_plan* plan, long long* out, int* round_of_group);
/* Device self-test of the 6-instruction division the kernels use for per-step denominators (sx_math.h sx_fdiv)
 * against the IEEE quotient on n pseudo-random pairs, b in [blo, bhi): out[0] = mismatching calls, out[1] = of
 * those, off by more than one ulp. */
int smashx_selftest_math(int device, long long n, unsigned seed, float blo, float bhi, long long* out);

/* Device self-test of the per-wavefront straight-line paths of sx_math.h (sx_tanhf for small arguments, the base-2 logarithm and the
 * power for ordinary operands) against the branchy forms they shortcut, on n pseudo-random arguments, a wavefront at a time uniform
 * (the fast path runs) or mixed with special operands (it must not): out[0] = tanh results that differ, out[1] = powers that differ.
 * Both must be 0: the paths execute the same operations. */
int smashx_selftest_paths(int device, long long n, unsigned seed, long long* out);

/* ---- L-BFGS-B, the optimiser of the variational calibration (reference: lbfgsb.f driven by optimize_lbfgsb,
 * smash/solver/optimize/mw_optimize.f90:484-676: m = 10, factr, pgtol, bounds of the normalised control, reverse communication).
 * A from-the-paper implementation (Byrd-Lu-Nocedal-Zhu 1995, Morales-Nocedal 2011, More'-Thuente line search; smash_amd/csrc/
 * sx_lbfgsb.cpp), host C++ with threaded n-vector work, no GPU and no third-party library needed: same method, parameters and stopping
 * tests as lbfgsb.f, iterates equal to rounding of the inner products (not bit for bit).  lower / upper: NULL or n values (+-inf = none).
 * Protocol: task = SMASHX_LBFGSB_START; loop { step(...); FG: evaluate f and g at x and call again; NEW_X: an iteration is complete (x is
 * the new iterate), call again to continue; CONVERGED / ABNORMAL: done }. */
#define SMASHX_LBFGSB_START 0
#define SMASHX_LBFGSB_FG 1
#define SMASHX_LBFGSB_NEW_X 2
#define SMASHX_LBFGSB_CONVERGED 3
#define SMASHX_LBFGSB_ABNORMAL 4
typedef struct smashx_lbfgsb smashx_lbfgsb;
int smashx_lbfgsb_create(long n, int m, const double* lower, const double* upper, double factr, double pgtol, smashx_lbfgsb** out);
int smashx_lbfgsb_step(smashx_lbfgsb* opt, double* x, double f, const double* g, int* task);
long smashx_lbfgsb_iterations(const smashx_lbfgsb* opt);
long smashx_lbfgsb_evaluations(const smashx_lbfgsb* opt);              /* f / g evaluations so far (isave(34) of lbfgsb.f) */
double smashx_lbfgsb_projected_gradient(const smashx_lbfgsb* opt);     /* infinity norm of the projected gradient at the last iterate (dsave(13)) */
const char* smashx_lbfgsb_message(const smashx_lbfgsb* opt);
int smashx_lbfgsb_destroy(smashx_lbfgsb* opt);

#ifdef __cplusplus
}
#endif
#endif /* SMASHX_H */
