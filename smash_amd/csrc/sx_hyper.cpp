// sx_hyper.cpp -- the hyper-linear / hyper-polynomial mappings of the reference on the host (include/smashx.h, "hyper mappings").
//
// mw_forward::hyper_forward(_b, _d) differ from forward(_b, _d) by one step in front of the time loop and its adjoint / tangent behind
// it: every parameter (state) field is a sigmoid of a linear or polynomial form of nd catchment descriptors,
//     field_i = lb_i + (ub_i - lb_i) / (1 + exp(-(h(1,i) + sum_j a_j d_j ** b_j)))
// (hyper_parameters_to_parameters, smash/solver/routine/mwd_parameters_manipulation.f90:304-362; hyper_states_to_states,
// mwd_states_manipulation.f90:270-329; their Tapenade twins HYPER_*_TO_*_D / _B, forward_db.f90:1313-1403, 1434-1537, 2179-2256,
// 2272-2369).  It is host code in the reference -- nhyper x 24 coefficients against whole-grid planes, once per call -- and stays host
// code here: plain C++ on the caller's planes, fp32 in the reference's operation order, powf / expf / logf from the C library the
// reference's own build calls, whole-grid sums in column-major order.  Nothing here touches the GPU; the time loop between the map and
// its adjoint is smashx_forward / smashx_forward_b / smashx_forward_d.
#include <cmath>
#include <cstddef>
#include <vector>

#include "../../include/smashx.h"

namespace {

int nhyper_of(const smashx_hyper_map* m) { return m->mapping == SMASHX_HYPER_POLYNOMIAL ? 1 + 2 * m->nd : 1 + m->nd; }

bool bad(const smashx_hyper_map* m) {
    return !m || (m->mapping != SMASHX_HYPER_LINEAR && m->mapping != SMASHX_HYPER_POLYNOMIAL) || m->nrow < 1 || m->ncol < 1 || m->nd < 0 ||
           m->nfields < 1 || (m->nd > 0 && !m->descriptor) || !m->lb || !m->ub;
}

// coefficient a and exponent b of descriptor j (0-based) in column i of the hyper matrix (nhyper x nfields, column-major)
inline void coef(const smashx_hyper_map* m, const float* h, int nh, int i, int j, float& a, float& b) {
    if (m->mapping == SMASHX_HYPER_LINEAR) { a = h[(size_t)i * nh + j + 1]; b = 1.f; }
    else { a = h[(size_t)i * nh + 2 * j + 1]; b = h[(size_t)i * nh + 2 * j + 2]; }
}
// d ** b the way the reference's compiler evaluates it: a call of powf (an exponent of exactly 1 returns the base)
inline float powb(float d, float b) { return b == 1.f ? d : powf(d, b); }

}  // namespace

extern "C" {

int smashx_hyper_nhyper(const smashx_hyper_map* m) { return bad(m) ? SMASHX_E_ARG : nhyper_of(m); }

// planes[i] (nrow, ncol) <- the mapped field i; NULL planes are skipped
int smashx_hyper_map_forward(const smashx_hyper_map* m, const float* hyper, float* const* planes) {
    if (bad(m) || !hyper || !planes) return SMASHX_E_ARG;
    const size_t n2 = (size_t)m->nrow * m->ncol;
    const int nh = nhyper_of(m);
    for (int i = 0; i < m->nfields; ++i) {
        float* p = planes[i];
        if (!p) continue;
        for (size_t c = 0; c < n2; ++c) p[c] = hyper[(size_t)i * nh];
        for (int j = 0; j < m->nd; ++j) {
            const float* d = m->descriptor + (size_t)j * n2;
            float a, b;
            coef(m, hyper, nh, i, j, a, b);
            for (size_t c = 0; c < n2; ++c) p[c] = p[c] + a * powb(d[c], b);
        }
        const float w = m->ub[i] - m->lb[i];      // sigmoid transformation, lambda = 1
        for (size_t c = 0; c < n2; ++c) p[c] = w * (1.f / (1.f + expf(-p[c]))) + m->lb[i];
    }
    return 0;
}

// tangent: planes / planes_d <- value and directional derivative along hyper_d (the value in the form the tangent code evaluates it)
int smashx_hyper_map_d(const smashx_hyper_map* m, const float* hyper, const float* hyper_d, float* const* planes, float* const* planes_d) {
    if (bad(m) || !hyper || !hyper_d || !planes || !planes_d) return SMASHX_E_ARG;
    const size_t n2 = (size_t)m->nrow * m->ncol;
    const int nh = nhyper_of(m);
    for (int i = 0; i < m->nfields; ++i) {
        float *p = planes[i], *pd = planes_d[i];
        if (!p || !pd) continue;
        for (size_t c = 0; c < n2; ++c) { pd[c] = hyper_d[(size_t)i * nh]; p[c] = hyper[(size_t)i * nh]; }
        for (int j = 0; j < m->nd; ++j) {
            const float* d = m->descriptor + (size_t)j * n2;
            float a, b, a_d, b_d;
            coef(m, hyper, nh, i, j, a, b);
            coef(m, hyper_d, nh, i, j, a_d, b_d);
            if (m->mapping == SMASHX_HYPER_LINEAR) b_d = 0.f;
            for (size_t c = 0; c < n2; ++c) {
                const float t = powb(d[c], b);
                const float dpb_d = d[c] <= 0.f ? 0.f : t * logf(d[c]) * b_d;
                pd[c] = pd[c] + (t * a_d + a * dpb_d);
                p[c] = p[c] + a * t;
            }
        }
        const float w = m->ub[i] - m->lb[i];
        for (size_t c = 0; c < n2; ++c) {
            const float e = expf(-p[c]);
            const float t = w / (e + 1.f);
            pd[c] = t * e * pd[c] / (e + 1.f);
            p[c] = m->lb[i] + t;
        }
    }
    return 0;
}

// adjoint: hyper_b (nhyper x nfields) <- the gradient w.r.t. the coefficients, given the gradient planes_b of the mapped fields
// (NULL = a field the structure does not use: zero).  hyper_b is overwritten.
int smashx_hyper_map_b(const smashx_hyper_map* m, const float* hyper, float* const* planes_b, float* hyper_b) {
    if (bad(m) || !hyper || !planes_b || !hyper_b) return SMASHX_E_ARG;
    const size_t n2 = (size_t)m->nrow * m->ncol;
    const int nh = nhyper_of(m);
    std::vector<float> lin(n2), g(n2);
    for (size_t k = 0; k < (size_t)nh * m->nfields; ++k) hyper_b[k] = 0.f;
    for (int i = m->nfields - 1; i >= 0; --i) {
        const float* pb = planes_b[i];
        if (!pb) continue;                                   // (a zero plane adds zeros to every sum)
        for (size_t c = 0; c < n2; ++c) lin[c] = hyper[(size_t)i * nh];       // the linear form, recomputed
        for (int j = 0; j < m->nd; ++j) {
            const float* d = m->descriptor + (size_t)j * n2;
            float a, b;
            coef(m, hyper, nh, i, j, a, b);
            for (size_t c = 0; c < n2; ++c) lin[c] = lin[c] + a * powb(d[c], b);
        }
        const float w = m->ub[i] - m->lb[i];
        for (size_t c = 0; c < n2; ++c) {
            const float e = expf(-lin[c]);
            const float t = e + 1.f;
            g[c] = e * w * pb[c] / (t * t);
        }
        for (int j = m->nd - 1; j >= 0; --j) {
            const float* d = m->descriptor + (size_t)j * n2;
            float a, b;
            coef(m, hyper, nh, i, j, a, b);
            float a_b = 0.f, b_b = 0.f;
            for (size_t c = 0; c < n2; ++c) a_b = a_b + powb(d[c], b) * g[c];
            if (m->mapping == SMASHX_HYPER_POLYNOMIAL) {
                for (size_t c = 0; c < n2; ++c)
                    if (!(d[c] <= 0.f)) b_b = b_b + powb(d[c], b) * logf(d[c]) * (a * g[c]);
                hyper_b[(size_t)i * nh + 2 * j + 2] = hyper_b[(size_t)i * nh + 2 * j + 2] + b_b;
                hyper_b[(size_t)i * nh + 2 * j + 1] = hyper_b[(size_t)i * nh + 2 * j + 1] + a_b;
            } else {
                hyper_b[(size_t)i * nh + j + 1] = hyper_b[(size_t)i * nh + j + 1] + a_b;
            }
        }
        float s = 0.f;
        for (size_t c = 0; c < n2; ++c) s = s + g[c];
        hyper_b[(size_t)i * nh] = hyper_b[(size_t)i * nh] + s;
    }
    return 0;
}

}  // extern "C"
