/* smash_oracle.h -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (smash_amd/, include/smashx.h) never does.
 *
 * Parity status: PINNED -- checked bit-for-bit / to rounding against the unmodified reference
 * Fortran (flang -O2 -ffp-contract=off build, oracle/ref/build_ref.sh) on the golden cases under
 * tests/golden/ (generator: tests/golden/make_golden.py).
 */
#ifndef SMASH_ORACLE_H
#define SMASH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_GNP 16
#define ORC_GNS 8

enum { ORC_GR_A = 1, ORC_GR_B = 2, ORC_GR_C = 3, ORC_GR_D = 4, ORC_VIC_A = 5 };
enum { ORC_NSE = 1, ORC_KGE = 2, ORC_KGE2 = 3, ORC_SE = 4, ORC_RMSE = 5, ORC_LOGARITHMIC = 6 };
enum { ORC_PRIOR = 1, ORC_SMOOTHING = 2, ORC_HARD_SMOOTHING = 3 };

typedef struct {
    int structure;               /* ORC_GR_*  (setup%structure, forward.f90:43-65) */
    int nrow, ncol, nt, ng;
    int denormalize_forward;     /* setup%optimize%denormalize_forward (forward.f90:33) */
    int optimize_start_step;     /* 1-based (mwd_cost.f90:80) */
    int njf;
    int jobs_fun[8];
    float wjobs_fun[8];
    int njr;
    int jreg_fun[4];
    float wjreg_fun[4];
    float wjreg;
    float dt, dx;
    int optim_parameters[ORC_GNP];
    int optim_states[ORC_GNS];
    float lb_parameters[ORC_GNP], ub_parameters[ORC_GNP];
    float lb_states[ORC_GNS], ub_states[ORC_GNS];
} orc_config;

/* All 2-D/3-D arrays are column-major exactly as the reference holds them (row index fastest).
 * path is (2, nrow*ncol) and gauge_pos (ng, 2), both 0-based here.
 * params / states are the (nrow,ncol,16) / (nrow,ncol,8) packings of get_parameters / get_states
 * (mwd_parameters_manipulation.f90:59, mwd_states_manipulation.f90:58); they come back
 * denormalised when denormalize_forward is set, states restored to entry values (forward.f90:41,72).
 */
/* optional whole-domain stores of the NEXT orc_forward calls (setup%save_qsim_domain / save_net_prcp_domain,
 * md_forward_structure.f90:158-194): (nrow, ncol, nt) column-major, only active cells are written; NULL = off */
void orc_set_domain_outputs(float* qsim_domain, float* net_prcp_domain);

int orc_forward(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                const int* active_cell, const int* gauge_pos, const float* area, const float* prcp,
                const float* pet, const float* qobs, const float* wgauge, float* params,
                const float* params_bgd, float* states, const float* states_bgd, float* qsim,
                float* costs /* cost, jobs, jreg */, float* fstates);

int orc_forward_b(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                  const int* active_cell, const int* gauge_pos, const float* area, const float* prcp,
                  const float* pet, const float* qobs, const float* wgauge, float* params,
                  const float* params_bgd, float* states, const float* states_bgd, float cost_b,
                  float* qsim, float* costs, float* params_b, float* states_b);

/* Tangent model, base_forward_d (forward_db.f90:10517-10601), smash_oracle_d.c.  params_d / states_d: the direction
 * (in normalised space when denormalize_forward; they come back denormalised like the reference leaves them);
 * qsim_d (ng, nt) and cost_d out; costs[1] = jobs.  The background tangents are passive in the reference. */
int orc_forward_d(const orc_config* cfg, const int* flwdir, const int* flwacc, const int* path,
                  const int* active_cell, const int* gauge_pos, const float* area, const float* prcp,
                  const float* pet, const float* qobs, const float* wgauge, float* params, float* params_d,
                  const float* params_bgd, float* states, float* states_d, const float* states_bgd,
                  float* qsim, float* qsim_d, float* costs, float* cost_d);

#ifdef __cplusplus
}
#endif
#endif
