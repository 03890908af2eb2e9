/* Host check of the library's L-BFGS-B (smash_amd/csrc/sx_lbfgsb.cpp) through its C entry points (include/smashx.h), built with
   -fsanitize=address,undefined -- and, a second time, -fsanitize=thread -- by tests/test_lbfgsb_sanitized.py.  Runs bounded, badly
   scaled problems of sizes below and above the threading threshold and checks what must hold whatever the rounding: the iterates stay
   in the box, the accepted function values never increase, the run ends by a convergence test with a small projected gradient.
   usage: sx_lbfgsb_check n m seed maxiter */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "smashx.h"

static unsigned long long rng_state;
static double uni() { rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(rng_state >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const long n = atol(argv[1]);
    const int m = atoi(argv[2]);
    rng_state = (unsigned long long)atol(argv[3]) * 2654435761ULL + 12345;
    const int maxiter = atoi(argv[4]);
    std::vector<double> tgt(n), w(n), lo(n, 0.0), up(n, 1.0), x(n, 0.5), g(n, 0.0);
    for (long i = 0; i < n; ++i) { tgt[i] = -0.3 + 1.6 * uni(); w[i] = std::pow(10.0, -2.0 + 4.0 * uni()); }
    for (long i = 0; i < n; i += 7) x[i] = 0.0;          // some variables start on a bound
    for (long i = 3; i < n; i += 11) x[i] = 1.0;
    auto fun = [&](const std::vector<double>& p, std::vector<double>& gr) {
        double f = 0.0;
        for (long i = 0; i < n; ++i) {
            const double d = p[i] - tgt[i];
            f += w[i] * d * d + 0.1 * std::sin(5.0 * p[i]);
            gr[i] = 2.0 * w[i] * d + 0.5 * std::cos(5.0 * p[i]);
        }
        return f;
    };
    smashx_lbfgsb* h = nullptr;
    if (smashx_lbfgsb_create(n, m, lo.data(), up.data(), 10.0, 1e-9, &h) != 0) { printf("VIOLATION create\n"); return 1; }
    int task = SMASHX_LBFGSB_START, nit = 0, nfev = 0;
    double f = 0.0, flast = INFINITY;
    for (;;) {
        if (smashx_lbfgsb_step(h, x.data(), f, g.data(), &task) != 0) { printf("VIOLATION step\n"); return 1; }
        if (task == SMASHX_LBFGSB_FG) {
            for (long i = 0; i < n; ++i) if (!(x[i] >= 0.0 && x[i] <= 1.0)) { printf("VIOLATION outside the box: x[%ld] = %.17g\n", i, x[i]); return 1; }
            f = fun(x, g); ++nfev;
        } else if (task == SMASHX_LBFGSB_NEW_X) {
            if (!(f <= flast)) { printf("VIOLATION increase %.17g -> %.17g at iteration %d\n", flast, f, nit); return 1; }
            flast = f;
            if (++nit >= maxiter) break;
        } else break;
    }
    double pg = 0.0;                                      // projected gradient at the end
    for (long i = 0; i < n; ++i) {
        const double gi = g[i], step = gi < 0.0 ? std::fmax(x[i] - 1.0, gi) : std::fmin(x[i] - 0.0, gi);
        pg = std::fmax(pg, std::fabs(step));
    }
    if (smashx_lbfgsb_iterations(h) != nit && task != SMASHX_LBFGSB_CONVERGED) { printf("VIOLATION iteration count\n"); return 1; }
    printf("ok n %ld it %d evals %d f %.12g projg %.3g task %d (%s)\n", n, nit, nfev, f, pg, task, smashx_lbfgsb_message(h));
    if (nit < maxiter && task != SMASHX_LBFGSB_CONVERGED) { printf("VIOLATION ended without convergence\n"); return 1; }
    if (nit < maxiter && pg > 1e-4) { printf("VIOLATION projected gradient %.3g at convergence\n", pg); return 1; }
    smashx_lbfgsb_destroy(h);
    return 0;
}
