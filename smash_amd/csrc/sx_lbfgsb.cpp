// sx_lbfgsb.cpp -- limited-memory BFGS with bounds, written from the published algorithm:
//   R. H. Byrd, P. Lu, J. Nocedal, C. Zhu, "A limited memory algorithm for bound constrained optimization", SIAM J. Sci. Comput. 16
//   (1995): generalized Cauchy point (Algorithm CP), direct primal subspace minimisation (sec. 5.1), compact representation
//   B = theta I - W M W^T;   J. L. Morales, J. Nocedal, "Remark on Algorithm 778" (2011): projection of the subspace point;
//   J. J. More', D. J. Thuente, "Line search algorithms with guaranteed sufficient decrease", ACM TOMS 20 (1994): the line search.
// It stands where the reference calls lbfgsb.f (smash/solver/optimize/mw_optimize.f90:541-633: m = 10, factr, pgtol, reverse
// communication) -- same method, same parameters, same stopping tests, NOT the same code: iterates agree with lbfgsb.f to rounding of the
// inner products, not bit for bit.  Host-only C++ (no GPU): the n-vector work is spread over threads, so the 1.7e7 control variables of
// a 2048 x 2048 calibration cost a fraction of a second per iteration instead of the ~0.7 s of the single-threaded library routine.
#include "../../include/smashx.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

const double EPS = std::numeric_limits<double>::epsilon();
const double INF = std::numeric_limits<double>::infinity();

// ---- n-vector helpers, threaded above a size where it pays --------------------------------------------------------------------
int n_threads(long n) {
    if (n < (1L << 18)) return 1;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::min<long>(std::min<unsigned>(hw, 16u), n >> 16);
}
template <class F> void pfor(long n, F f) {          // f(thread, begin, end)
    const int T = n_threads(n);
    if (T == 1) { f(0, 0L, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=]() { f(t, n * t / T, n * (t + 1) / T); });
    for (auto& x : th) x.join();
}
template <class F> double psum(long n, F term) {      // sum of term(i), fixed partition => the same sum for the same n on every call
    const int T = n_threads(n);
    std::vector<double> part(T, 0.0);
    pfor(n, [&](int t, long a, long b) { double s = 0.0; for (long i = a; i < b; ++i) s += term(i); part[t] = s; });
    double s = 0.0;
    for (double v : part) s += v;
    return s;
}

// K running sums in ONE pass over 0..n-1: body(i, acc) adds element i's terms to acc[0..K)
template <class F> void pacc(long n, int K, double* out, F body) {
    const int T = n_threads(n);
    std::vector<double> part((size_t)T * K, 0.0);
    pfor(n, [&](int t, long a, long b) { double* acc = part.data() + (size_t)t * K; for (long i = a; i < b; ++i) body(i, acc); });
    for (int k = 0; k < K; ++k) { double s = 0.0; for (int t = 0; t < T; ++t) s += part[(size_t)t * K + k]; out[k] = s; }
}

// ---- small dense linear algebra (at most 2m x 2m = 20 x 20) ---------------------------------------------------------------------
struct Dense {
    int n = 0;
    std::vector<double> a;      // row-major
    std::vector<int> piv;
    bool ok = true;
    void resize(int k) { n = k; a.assign((size_t)k * k, 0.0); piv.assign(k, 0); ok = true; }
    double& at(int i, int j) { return a[(size_t)i * n + j]; }
    double at(int i, int j) const { return a[(size_t)i * n + j]; }
    void factor() {             // LU with partial pivoting, in place
        ok = true;
        for (int k = 0; k < n; ++k) {
            int p = k; double mx = std::fabs(at(k, k));
            for (int i = k + 1; i < n; ++i) if (std::fabs(at(i, k)) > mx) { mx = std::fabs(at(i, k)); p = i; }
            piv[k] = p;
            if (!(mx > 0.0)) { ok = false; return; }
            if (p != k) for (int j = 0; j < n; ++j) std::swap(at(k, j), at(p, j));
            for (int i = k + 1; i < n; ++i) {
                const double f = at(i, k) / at(k, k);
                at(i, k) = f;
                for (int j = k + 1; j < n; ++j) at(i, j) -= f * at(k, j);
            }
        }
    }
    void solve(double* b) const {   // after factor()
        for (int k = 0; k < n; ++k) if (piv[k] != k) std::swap(b[k], b[piv[k]]);     // whole rows were exchanged: all interchanges first
        for (int k = 0; k < n; ++k) for (int i = k + 1; i < n; ++i) b[i] -= at(i, k) * b[k];
        for (int k = n - 1; k >= 0; --k) { for (int j = k + 1; j < n; ++j) b[k] -= at(k, j) * b[j]; b[k] /= at(k, k); }
    }
};

// ---- More'-Thuente line search (MINPACK-2 dcsrch / dcstep, restated) -------------------------------------------------------------
struct LineSearch {
    bool brackt = false; int stage = 1;
    double finit = 0, ginit = 0, gtest = 0, width = 0, width1 = 0, stx = 0, fx = 0, gx = 0, sty = 0, fy = 0, gy = 0, stmin = 0, stmax = 0;
    double ftol = 1e-3, gtol = 0.9, xtol = 0.1, stpmin = 0.0, stpmax = 1e10;
    enum { FG = 0, CONVERGED = 1, WARNING = 2, ERROR = 3 };

    static void step(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp, double fp, double dp,
                     bool& brackt, double stpmin, double stpmax) {
        const double sgnd = dp * (dx / std::fabs(dx));
        double stpf, stpc, stpq;
        if (fp > fx) {                                   // higher value: the minimum is bracketed
            const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp, s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
            double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
            if (stp < stx) gamma = -gamma;
            const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
            stpc = stx + r * (stp - stx);
            stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
            stpf = std::fabs(stpc - stx) < std::fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
            brackt = true;
        } else if (sgnd < 0.0) {                         // lower value, derivatives of opposite sign: bracketed
            const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp, s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
            double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
            if (stp > stx) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
            stpc = stp + r * (stx - stp);
            stpq = stp + (dp / (dp - dx)) * (stx - stp);
            stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
            brackt = true;
        } else if (std::fabs(dp) < std::fabs(dx)) {      // lower value, same sign, derivative decreases
            const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp, s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
            double gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
            if (stp > stx) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
            if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
            else stpc = stp > stx ? stpmax : stpmin;
            stpq = stp + (dp / (dp - dx)) * (stx - stp);
            if (brackt) {
                stpf = std::fabs(stpc - stp) < std::fabs(stpq - stp) ? stpc : stpq;
                stpf = stp > stx ? std::min(stp + 0.66 * (sty - stp), stpf) : std::max(stp + 0.66 * (sty - stp), stpf);
            } else {
                stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
                stpf = std::max(stpmin, std::min(stpmax, stpf));
            }
        } else {                                         // lower value, same sign, derivative does not decrease
            if (brackt) {
                const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp, s = std::max({std::fabs(theta), std::fabs(dy), std::fabs(dp)});
                double gamma = s * std::sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
                if (stp > sty) gamma = -gamma;
                const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
                stpf = stp + r * (sty - stp);
            } else stpf = stp > stx ? stpmax : stpmin;
        }
        if (fp > fx) { sty = stp; fy = fp; dy = dp; }
        else { if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; } stx = stp; fx = fp; dx = dp; }
        stp = stpf;
    }
    void start(double f, double g, double stp) {
        brackt = false; stage = 1; finit = f; ginit = g; gtest = ftol * ginit; width = stpmax - stpmin; width1 = width / 0.5;
        stx = 0; fx = finit; gx = ginit; sty = 0; fy = finit; gy = ginit; stmin = 0; stmax = stp + 4.0 * stp;
    }
    int next(double& stp, double f, double g) {           // f, g at stp; returns FG with the next trial step, or a final state
        const double ftest = finit + stp * gtest;
        if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
        int task = FG;
        if (brackt && (stp <= stmin || stp >= stmax)) task = WARNING;
        if (brackt && stmax - stmin <= xtol * stmax) task = WARNING;
        if (stp == stpmax && f <= ftest && g <= gtest) task = WARNING;
        if (stp == stpmin && (f > ftest || g >= gtest)) task = WARNING;
        if (f <= ftest && std::fabs(g) <= gtol * (-ginit)) task = CONVERGED;
        if (task != FG) return task;
        if (stage == 1 && f <= fx && f > ftest) {
            double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest, gm = g - gtest, gxm = gx - gtest, gym = gy - gtest;
            step(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
            fx = fxm + stx * gtest; fy = fym + sty * gtest; gx = gxm + gtest; gy = gym + gtest;
        } else step(stx, fx, gx, sty, fy, gy, stp, f, g, brackt, stmin, stmax);
        if (brackt) {
            if (std::fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
            width1 = width; width = std::fabs(sty - stx);
        }
        if (brackt) { stmin = std::min(stx, sty); stmax = std::max(stx, sty); }
        else { stmin = stp + 1.1 * (stp - stx); stmax = stp + 4.0 * (stp - stx); }
        stp = std::max(stpmin, std::min(stpmax, stp));
        if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) stp = stx;
        return FG;
    }
};

}  // namespace

struct smashx_lbfgsb {
    long n = 0; int m = 10;
    double factr = 1e7, pgtol = 1e-5;
    std::vector<double> lo, up;                 // +-inf where absent
    bool boxed = true, constrained = false;
    // history: S, Y as m columns of n (column j at [j * n]); order[] lists the stored columns oldest first
    std::vector<std::unique_ptr<double[]>> S, Y;      // a column is allocated when its slot is first used (2 x m x n doubles otherwise
    std::vector<int> order;                            // have to be touched before the first iteration: 10 s at n = 1.7e7)
    std::vector<double> ss, sy, yy;             // m x m inner products by storage slot: s_i.s_j, s_i.y_j, y_i.y_j
    double theta = 1.0;
    // iteration state
    std::vector<double> x0, g0, xc, d, r, z;    // iterate / gradient at the start of the line search, Cauchy point, direction
    std::vector<double> xbar;                   // the subspace minimiser: the unit-step trial point, taken as it is (x0 + (xbar - x0) may differ in the last bit)
    std::vector<char> free_var;                 // at the Cauchy point
    int state = 0;                              // 0 not started, 1 waits for f(x0), 2 inside a line search
    long iter = 0; int nfev_ls = 0, total_fev = 0;
    double f0 = 0, gd0 = 0, stp = 0, dnorm = 0, sbgnrm = 0, stpmx = 0;
    LineSearch ls;
    int restarts = 0;
    std::string msg;
    const char* pending = nullptr;              // convergence verdict reached with the last accepted step, delivered after its NEW_X

    int col() const { return (int)order.size(); }
    const double* Scol(int k) const { return S[order[k]].get(); }
    const double* Ycol(int k) const { return Y[order[k]].get(); }

    // middle matrix M = [ -D  L^T ; L  theta S^T S ]^-1 over the stored pairs (oldest first), as an LU factorisation
    Dense Minv;
    void form_middle() {
        const int c = col();
        Minv.resize(2 * c);
        for (int i = 0; i < c; ++i)
            for (int j = 0; j < c; ++j) {
                const int a = order[i], b = order[j];
                if (i == j) Minv.at(i, j) = -sy[(size_t)a * m + b];                    // -D
                if (i > j) { Minv.at(c + i, j) = sy[(size_t)a * m + b]; Minv.at(j, c + i) = sy[(size_t)a * m + b]; }   // L and L^T
                Minv.at(c + i, c + j) = theta * ss[(size_t)a * m + b];
            }
        Minv.factor();
    }
    void applyM(double* v) const { if (col() > 0) Minv.solve(v); }
    // W^T v = (Y^T v ; theta S^T v), optionally over a subset
    void WtV(const double* v, const char* mask, char want, double* out) const {
        const int c = col();
        if (c == 0) return;
        std::vector<const double*> yc(c), sc(c);
        for (int k = 0; k < c; ++k) { yc[k] = Ycol(k); sc[k] = Scol(k); }
        pacc(n, 2 * c, out, [&](long i, double* acc) {
            if (mask && mask[i] != want) return;
            const double vi = v[i];
            if (vi == 0.0) return;
            for (int k = 0; k < c; ++k) { acc[k] += yc[k][i] * vi; acc[c + k] += sc[k][i] * vi; }
        });
        for (int k = 0; k < c; ++k) out[c + k] *= theta;
    }
    void Wrow(long i, double* w) const { const int c = col(); for (int k = 0; k < c; ++k) { w[k] = Ycol(k)[i]; w[c + k] = theta * Scol(k)[i]; } }

    double proj_grad_norm(const double* x, const double* g) const {
        const int T = n_threads(n);
        std::vector<double> part(T, 0.0);
        pfor(n, [&](int t, long a, long b) {
            double mx = 0.0;
            for (long i = a; i < b; ++i) {
                double gi = g[i];
                if (gi < 0.0) { if (up[i] < INF) gi = std::max(x[i] - up[i], gi); }
                else if (lo[i] > -INF) gi = std::min(x[i] - lo[i], gi);
                mx = std::max(mx, std::fabs(gi));
            }
            part[t] = mx;
        });
        return *std::max_element(part.begin(), part.end());
    }

    // ---- generalized Cauchy point (Algorithm CP).  Leaves xc, free_var and c = W^T (xc - x) in cvec
    std::vector<double> cvec;
    void cauchy(const double* x, const double* g) {
        const int c2 = 2 * col();
        std::vector<double> p(c2, 0.0), w(c2), v(c2);
        cvec.assign(c2, 0.0);
        std::vector<double> t(n);
        // breakpoints and the steepest descent direction
        pfor(n, [&](int, long a, long b) {
            for (long i = a; i < b; ++i) {
                double ti = INF;
                if (g[i] < 0.0 && up[i] < INF) ti = (x[i] - up[i]) / g[i];
                else if (g[i] > 0.0 && lo[i] > -INF) ti = (x[i] - lo[i]) / g[i];
                else if (g[i] == 0.0 && (x[i] <= lo[i] || x[i] >= up[i])) ti = 0.0;     // on a bound with zero gradient: stays there
                t[i] = ti;
                d[i] = ti == 0.0 ? 0.0 : -g[i];
                xc[i] = x[i];
                free_var[i] = ti == 0.0 ? 0 : 1;
            }
        });
        if (c2 == 0) {
            // no curvature pairs yet: B = theta I, the model along the projected path has slope (theta tau - 1) * sum of g_i^2 over the
            // variables still free, whatever that set is -- the minimiser is tau = 1 / theta: the Cauchy point is the projection of the
            // steepest-descent step, no breakpoint needs sorting (Algorithm CP gives the same point one crossing at a time)
            const double tau = 1.0 / theta;
            pfor(n, [&](int, long a, long b) {
                for (long i = a; i < b; ++i) {
                    if (t[i] == 0.0) continue;
                    if (t[i] <= tau) { xc[i] = d[i] > 0.0 ? up[i] : lo[i]; free_var[i] = 0; }
                    else xc[i] = x[i] + tau * d[i];
                }
            });
            return;
        }
        WtV(d.data(), nullptr, 0, p.data());
        double fp = -psum(n, [&](long i) { return d[i] * d[i]; });
        const double fpp0 = -theta * fp;
        double fpp = fpp0;
        if (c2 > 0) { v = p; applyM(v.data()); double q = 0.0; for (int k = 0; k < c2; ++k) q += p[k] * v[k]; fpp -= q; }
        if (!(fp < 0.0)) { return; }                       // projected gradient is zero: x is the Cauchy point
        double dtm = -fp / fpp, told = 0.0;
        // Breakpoints in increasing order, only as many as the search crosses.  Most iterations stop before the first one (no list is
        // built then); otherwise the breakpoints are collected window by window -- (0, H], (H, 4H], ... with H a little beyond where the
        // minimiser is expected -- and each window is a heap: building one is O(window), each crossing O(log window).  Sorting all
        // 1.7e7 breakpoints of a boxed problem every iteration would cost more than everything else in it.
        typedef std::pair<double, long> Bp;
        auto later = [](const Bp& a, const Bp& b) { return a.first > b.first || (a.first == b.first && a.second > b.second); };
        double tmin = INF;
        {
            const int T = n_threads(n);
            std::vector<double> part(T, INF);
            pfor(n, [&](int th, long a, long b) { double mn = INF; for (long i = a; i < b; ++i) if (t[i] > 0.0 && t[i] < mn) mn = t[i]; part[th] = mn; });
            for (double v : part) tmin = std::min(tmin, v);
        }
        bool done = !(dtm >= tmin);
        double lowT = 0.0, H = std::max(tmin, 2.0 * dtm);
        while (!done) {
            std::vector<Bp> heap;
            long beyond = 0;
            {
                const int T = n_threads(n);
                std::vector<std::vector<Bp>> part(T);
                std::vector<long> more(T, 0);
                pfor(n, [&](int th, long a, long b) {
                    auto& v = part[th]; long c = 0;
                    for (long i = a; i < b; ++i) { if (t[i] > lowT && t[i] <= H) v.emplace_back(t[i], i); else if (t[i] > H && t[i] < INF) ++c; }
                    more[th] = c;
                });
                size_t tot = 0;
                for (auto& v : part) tot += v.size();
                heap.reserve(tot);
                for (auto& v : part) heap.insert(heap.end(), v.begin(), v.end());
                for (long c : more) beyond += c;
            }
            std::make_heap(heap.begin(), heap.end(), later);
            while (!heap.empty()) {
                const long b = heap.front().second;
                const double dt = t[b] - told;
                if (dtm < dt) { done = true; break; }
                std::pop_heap(heap.begin(), heap.end(), later);
                heap.pop_back();
                // variable b reaches its bound
                const double gb = g[b];
                xc[b] = d[b] > 0.0 ? up[b] : lo[b];
                const double zb = xc[b] - x[b];
                free_var[b] = 0;
                told = t[b];
                for (int q = 0; q < c2; ++q) cvec[q] += dt * p[q];
                fp += dt * fpp + gb * gb + theta * gb * zb;
                fpp -= theta * gb * gb;
                Wrow(b, w.data());
                v = cvec; applyM(v.data());
                double wmc = 0.0; for (int q = 0; q < c2; ++q) wmc += w[q] * v[q];
                v = p; applyM(v.data());
                double wmp = 0.0; for (int q = 0; q < c2; ++q) wmp += w[q] * v[q];
                v = w; applyM(v.data());
                double wmw = 0.0; for (int q = 0; q < c2; ++q) wmw += w[q] * v[q];
                fp -= gb * wmc;
                fpp -= 2.0 * gb * wmp + gb * gb * wmw;
                for (int q = 0; q < c2; ++q) p[q] += gb * w[q];
                fpp = std::max(EPS * fpp0, fpp);
                d[b] = 0.0;
                dtm = -fp / fpp;
                if (!(fp < 0.0)) { dtm = 0.0; done = true; break; }
            }
            if (done) break;
            if (beyond == 0 || told + dtm <= H) break;       // no breakpoint left, or the minimiser lies before the next one (> H)
            lowT = H;
            H = std::max(4.0 * H, told + 2.0 * dtm);
        }
        (void)done;
        dtm = std::max(dtm, 0.0);
        told += dtm;
        pfor(n, [&](int, long a, long b) { for (long i = a; i < b; ++i) if (free_var[i] && d[i] != 0.0) xc[i] = x[i] + told * d[i]; });
        if (c2 > 0) for (int q = 0; q < c2; ++q) cvec[q] += dtm * p[q];
    }

    // ---- subspace minimisation over the free variables (direct primal method), then projection (Morales-Nocedal)
    void subspace(const double* x, const double* g, std::vector<double>& xbar) {
        xbar = xc;
        const int c = col(), c2 = 2 * c;
        long nfree = 0;
        for (long i = 0; i < n; ++i) nfree += free_var[i];
        if (nfree == 0 || c == 0) return;
        // r = -(g + theta (xc - x) - W M c) on the free variables
        std::vector<double> mc(cvec);
        applyM(mc.data());
        pfor(n, [&](int, long a, long b) {
            for (long i = a; i < b; ++i) {
                if (!free_var[i]) { r[i] = 0.0; continue; }
                double wmc = 0.0;
                for (int k = 0; k < c; ++k) wmc += Ycol(k)[i] * mc[k] + theta * Scol(k)[i] * mc[c + k];
                r[i] = -(g[i] + theta * (xc[i] - x[i]) - wmc);
            }
        });
        // v = M W_F^T r;  N = I - (1/theta) M (W_F^T W_F);  solve N u = v;  d = (1/theta) (r + (1/theta) W_F u)
        std::vector<double> v(c2);
        WtV(r.data(), free_var.data(), 1, v.data());
        applyM(v.data());
        // W_F^T W_F from whichever side is smaller: the free set, or the whole (kept incrementally) minus the active set
        std::vector<double> wtw((size_t)c2 * c2, 0.0);
        const bool use_free = nfree * 2 <= n;
        const char want = use_free ? 1 : 0;
        {   // Gram matrix of the columns [y_1..y_c, s_1..s_c] over the chosen set, one pass, upper triangle
            std::vector<const double*> colp(c2);
            for (int k = 0; k < c; ++k) { colp[k] = Ycol(k); colp[c + k] = Scol(k); }
            const int K = c2 * (c2 + 1) / 2;
            std::vector<double> gram(K, 0.0);
            pacc(n, K, gram.data(), [&](long i, double* acc) {
                if (free_var[i] != want) return;
                double vals[128];
                for (int k = 0; k < c2; ++k) vals[k] = colp[k][i];
                int q = 0;
                for (int a = 0; a < c2; ++a) { const double va = vals[a]; for (int b = a; b < c2; ++b) acc[q++] += va * vals[b]; }
            });
            int q = 0;
            for (int a = 0; a < c2; ++a)
                for (int b = a; b < c2; ++b) {
                    double gv = gram[q++];
                    if (!use_free) {     // whole-vector inner products are kept incrementally: subtract the active set's share
                        const int ia = order[a % c], ib = order[b % c];
                        const double full = (a < c && b < c) ? yy[(size_t)ia * m + ib] : (a >= c && b >= c) ? ss[(size_t)ia * m + ib]
                                          : sy[(size_t)ib * m + ia];        // a < c <= b: y_a . s_b = sy[s index][y index]
                        gv = full - gv;
                    }
                    const double scale = (a >= c ? theta : 1.0) * (b >= c ? theta : 1.0);
                    wtw[(size_t)a * c2 + b] = wtw[(size_t)b * c2 + a] = scale * gv;
                }
        }
        Dense N; N.resize(c2);
        std::vector<double> colv(c2);
        for (int j = 0; j < c2; ++j) {
            for (int i = 0; i < c2; ++i) colv[i] = wtw[(size_t)i * c2 + j];
            applyM(colv.data());
            for (int i = 0; i < c2; ++i) N.at(i, j) = (i == j ? 1.0 : 0.0) - colv[i] / theta;
        }
        N.factor();
        if (!N.ok) return;
        N.solve(v.data());
        pfor(n, [&](int, long a, long b) {
            for (long i = a; i < b; ++i) {
                if (!free_var[i]) continue;
                double wu = 0.0;
                for (int k = 0; k < c; ++k) wu += Ycol(k)[i] * v[k] + theta * Scol(k)[i] * v[c + k];
                z[i] = (r[i] + wu / theta) / theta;                    // the unconstrained subspace step
                xbar[i] = std::max(lo[i], std::min(up[i], xc[i] + z[i]));   // projected onto the box
            }
        });
        // the projected point must give a descent direction from x; otherwise fall back to the largest feasible step along z
        const double dd = psum(n, [&](long i) { return (xbar[i] - x[i]) * g[i]; });
        if (dd > 0.0) {
            double alpha = 1.0;
            long blocking = -1;                                        // the first variable that sets the step
            for (long i = 0; i < n; ++i) {
                if (!free_var[i]) continue;
                double a1 = alpha;
                if (z[i] > 0.0 && up[i] < INF) a1 = std::max(0.0, (up[i] - xc[i]) / z[i]);
                else if (z[i] < 0.0 && lo[i] > -INF) a1 = std::max(0.0, (lo[i] - xc[i]) / z[i]);
                if (a1 < alpha) { alpha = a1; blocking = i; }
            }
            // (the blocking variable lands exactly on its bound; the clamp only catches a last-bit excursion of the others)
            pfor(n, [&](int, long a, long b) {
                for (long i = a; i < b; ++i) if (free_var[i]) xbar[i] = std::max(lo[i], std::min(up[i], xc[i] + alpha * z[i]));
            });
            if (blocking >= 0) xbar[blocking] = z[blocking] > 0.0 ? up[blocking] : lo[blocking];
        }
    }

    void reset_memory() { order.clear(); theta = 1.0; }

    // the line search's trial point x0 + stp d, inside the box to the last bit
    void trial_point(double* x) const {
        if (stp == 1.0) { pfor(n, [&](int, long a, long b) { std::copy(xbar.begin() + a, xbar.begin() + b, x + a); }); return; }
        pfor(n, [&](int, long a, long b) { for (long i = a; i < b; ++i) x[i] = std::max(lo[i], std::min(up[i], x0[i] + stp * d[i])); });
    }

    // store the pair (s, y) = (stp d, g_new - g0); returns false when it is skipped (curvature too small)
    bool update(const double* gnew) {
        // y into r, s into z
        pfor(n, [&](int, long a, long b) { for (long i = a; i < b; ++i) { r[i] = gnew[i] - g0[i]; z[i] = stp * d[i]; } });
        const double rr = psum(n, [&](long i) { return r[i] * r[i]; });
        const double dr = psum(n, [&](long i) { return r[i] * z[i]; });
        const double ddum = -gd0 * stp;
        if (dr <= EPS * ddum) return false;
        int slot;
        if (col() < m) { std::vector<char> used(m, 0); for (int k : order) used[k] = 1; slot = 0; while (used[slot]) ++slot; }
        else { slot = order.front(); order.erase(order.begin()); }
        if (!S[slot]) { S[slot].reset(new double[n]); Y[slot].reset(new double[n]); }
        double *sdst = S[slot].get(), *ydst = Y[slot].get();
        pfor(n, [&](int, long a, long b) { std::memcpy(sdst + a, z.data() + a, (size_t)(b - a) * sizeof(double)); std::memcpy(ydst + a, r.data() + a, (size_t)(b - a) * sizeof(double)); });
        order.push_back(slot);
        {   // inner products of the new pair with every stored one, one pass
            const int c = col();
            std::vector<const double*> sc(c), yc(c);
            for (int k = 0; k < c; ++k) { sc[k] = Scol(k); yc[k] = Ycol(k); }
            const double *sn = S[slot].get(), *yn = Y[slot].get();
            std::vector<double> acc(4 * c, 0.0);
            pacc(n, 4 * c, acc.data(), [&](long i, double* a) {
                const double si = sn[i], yi = yn[i];
                for (int k = 0; k < c; ++k) { a[4 * k] += sc[k][i] * si; a[4 * k + 1] += sc[k][i] * yi; a[4 * k + 2] += yc[k][i] * si; a[4 * k + 3] += yc[k][i] * yi; }
            });
            for (int q = 0; q < c; ++q) {
                const int k = order[q];
                ss[(size_t)k * m + slot] = ss[(size_t)slot * m + k] = acc[4 * q];
                sy[(size_t)k * m + slot] = acc[4 * q + 1];
                sy[(size_t)slot * m + k] = acc[4 * q + 2];
                yy[(size_t)k * m + slot] = yy[(size_t)slot * m + k] = acc[4 * q + 3];
            }
        }
        theta = rr / dr;
        return true;
    }
};

extern "C" {

int smashx_lbfgsb_create(long n, int m, const double* lower, const double* upper, double factr, double pgtol, smashx_lbfgsb** out) {
    if (!out || n <= 0 || m <= 0 || m > 64) return SMASHX_E_ARG;
    smashx_lbfgsb* o = new smashx_lbfgsb();
    o->n = n; o->m = m; o->factr = factr; o->pgtol = pgtol;
    o->lo.assign(n, -INF); o->up.assign(n, INF);
    if (lower) for (long i = 0; i < n; ++i) o->lo[i] = lower[i];
    if (upper) for (long i = 0; i < n; ++i) o->up[i] = upper[i];
    for (long i = 0; i < n; ++i) {
        if (o->lo[i] > o->up[i]) { delete o; return SMASHX_E_ARG; }
        if (!(o->lo[i] > -INF && o->up[i] < INF)) o->boxed = false;
        if (o->lo[i] > -INF || o->up[i] < INF) o->constrained = true;
    }
    o->S.resize(m); o->Y.resize(m);
    o->ss.assign((size_t)m * m, 0.0); o->sy.assign((size_t)m * m, 0.0); o->yy.assign((size_t)m * m, 0.0);
    o->x0.assign(n, 0.0); o->g0.assign(n, 0.0); o->xc.assign(n, 0.0); o->d.assign(n, 0.0); o->r.assign(n, 0.0); o->z.assign(n, 0.0);
    o->free_var.assign(n, 1);
    *out = o;
    return 0;
}

int smashx_lbfgsb_destroy(smashx_lbfgsb* o) { delete o; return 0; }

const char* smashx_lbfgsb_message(const smashx_lbfgsb* o) { return o ? o->msg.c_str() : ""; }

// One step of the reverse communication.  *task in: SMASHX_LBFGSB_START on the first call, else what the previous call returned;
// f, g: function value and gradient at x whenever the previous call returned SMASHX_LBFGSB_FG.  x is read and rewritten.
// Returns in *task: FG (evaluate at x and call again), NEW_X (an iteration is complete, x is the new iterate: call again to go on),
// CONVERGED (projected gradient or relative reduction test), ABNORMAL (line search failed twice in a row), ERROR.
int smashx_lbfgsb_step(smashx_lbfgsb* o, double* x, double f, const double* g, int* task) {
    if (!o || !x || !task) return SMASHX_E_ARG;
    const long n = o->n;
    auto begin_iteration = [&](const double* gx) -> bool {       // from (x0, f0, g0): direction and first trial point; false = converged
        for (;;) {
            o->form_middle();
            if (o->col() > 0 && !o->Minv.ok) { o->reset_memory(); continue; }
            if (o->constrained) o->cauchy(o->x0.data(), gx);
            else { std::copy(o->x0.begin(), o->x0.end(), o->xc.begin()); std::fill(o->free_var.begin(), o->free_var.end(), 1); o->cvec.assign(2 * o->col(), 0.0); }
            std::vector<double>& xbar = o->xbar;
            if (!o->constrained && o->col() == 0) {
                xbar.resize(n);
                for (long i = 0; i < n; ++i) xbar[i] = o->x0[i] - gx[i];
            } else if (!o->constrained) {
                // unconstrained: the Cauchy step is skipped, the subspace is the whole space (r = -g)
                o->subspace(o->x0.data(), gx, xbar);
            } else o->subspace(o->x0.data(), gx, xbar);
            pfor(n, [&](int, long a, long b) { for (long i = a; i < b; ++i) o->d[i] = xbar[i] - o->x0[i]; });
            o->gd0 = psum(n, [&](long i) { return gx[i] * o->d[i]; });
            o->dnorm = std::sqrt(psum(n, [&](long i) { return o->d[i] * o->d[i]; }));
            if (o->gd0 >= 0.0) {
                if (o->col() == 0) { o->msg = "ABNORMAL_TERMINATION_IN_LNSRCH"; return false; }
                o->reset_memory();                      // ascent direction in projection: refresh the memory and restart the iteration
                continue;
            }
            break;
        }
        o->stpmx = 1e10;
        if (o->constrained) {
            if (o->iter == 0) o->stpmx = 1.0;
            else {
                double s = 1e10;
                for (long i = 0; i < n; ++i) {
                    const double di = o->d[i];
                    if (di < 0.0 && o->lo[i] > -INF) { const double a = o->lo[i] - o->x0[i]; if (a >= 0.0) s = 0.0; else if (di * s < a) s = a / di; }
                    else if (di > 0.0 && o->up[i] < INF) { const double a = o->up[i] - o->x0[i]; if (a <= 0.0) s = 0.0; else if (di * s > a) s = a / di; }
                }
                o->stpmx = s;
            }
        }
        o->stp = (o->iter == 0 && !o->boxed) ? std::min(1.0 / o->dnorm, o->stpmx) : 1.0;
        o->ls = LineSearch();
        o->ls.stpmax = o->stpmx;
        o->ls.start(o->f0, o->gd0, o->stp);
        o->nfev_ls = 0;
        o->trial_point(x);
        return true;
    };

    if (*task == SMASHX_LBFGSB_START) {
        for (long i = 0; i < n; ++i) x[i] = std::max(o->lo[i], std::min(o->up[i], x[i]));      // project the start into the box
        o->reset_memory(); o->iter = 0; o->total_fev = 0; o->restarts = 0;
        o->state = 1;
        *task = SMASHX_LBFGSB_FG;
        return 0;
    }
    if (!g) return SMASHX_E_ARG;
    if (o->state == 1) {                                   // f, g at the projected starting point
        o->total_fev++;
        std::copy(x, x + n, o->x0.begin()); std::copy(g, g + n, o->g0.begin()); o->f0 = f;
        o->sbgnrm = o->proj_grad_norm(x, g);
        if (o->sbgnrm <= o->pgtol) { o->msg = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"; *task = SMASHX_LBFGSB_CONVERGED; return 0; }
        if (!begin_iteration(o->g0.data())) { *task = SMASHX_LBFGSB_ABNORMAL; return 0; }
        o->state = 2;
        *task = SMASHX_LBFGSB_FG;
        return 0;
    }
    if (*task == SMASHX_LBFGSB_NEW_X) {                    // the caller accepted the iterate: next iteration from it
        // lbfgsb.f hands every accepted iterate to its caller as NEW_X first and tests for convergence on re-entry (mainlb, label 777):
        // the caller sees the last iterate like every other (its per-iteration print, its own stopping tests, the iteration count)
        if (o->pending) { o->msg = o->pending; o->pending = nullptr; *task = SMASHX_LBFGSB_CONVERGED; return 0; }
        if (!begin_iteration(o->g0.data())) { *task = SMASHX_LBFGSB_ABNORMAL; return 0; }
        o->state = 2;
        *task = SMASHX_LBFGSB_FG;
        return 0;
    }
    if (o->state != 2) return SMASHX_E_STATE;
    // inside the line search: f, g at x = x0 + stp d
    o->total_fev++; o->nfev_ls++;
    const double gd = psum(n, [&](long i) { return g[i] * o->d[i]; });
    int ls = o->ls.next(o->stp, f, gd);
    if (ls == LineSearch::FG && o->nfev_ls >= 20) ls = LineSearch::ERROR;        // maxls
    if (ls == LineSearch::FG) {
        o->trial_point(x);
        *task = SMASHX_LBFGSB_FG;
        return 0;
    }
    if (ls == LineSearch::ERROR || (ls == LineSearch::WARNING && !(f < o->f0))) {
        // line search failed: back to the start of the iteration; with memory, drop it and try the steepest descent from there
        std::copy(o->x0.begin(), o->x0.end(), x);
        if (o->col() == 0 || o->restarts >= 1) { o->msg = "ABNORMAL_TERMINATION_IN_LNSRCH"; *task = SMASHX_LBFGSB_ABNORMAL; return 0; }
        o->restarts++;
        o->reset_memory();
        if (!begin_iteration(o->g0.data())) { *task = SMASHX_LBFGSB_ABNORMAL; return 0; }
        *task = SMASHX_LBFGSB_FG;
        return 0;
    }
    // the step is accepted: new iterate
    o->restarts = 0;
    o->iter++;
    const double fold = o->f0;
    o->sbgnrm = o->proj_grad_norm(x, g);
    o->update(g);
    std::copy(x, x + n, o->x0.begin()); std::copy(g, g + n, o->g0.begin()); o->f0 = f;
    o->pending = nullptr;
    const double ddum = std::max({std::fabs(fold), std::fabs(f), 1.0});
    if (o->sbgnrm <= o->pgtol) o->pending = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL";
    else if (fold - f <= EPS * o->factr * ddum) o->pending = "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH";
    o->state = 3;                                          // waits for the caller's NEW_X acknowledgement
    *task = SMASHX_LBFGSB_NEW_X;
    return 0;
}

long smashx_lbfgsb_iterations(const smashx_lbfgsb* o) { return o ? o->iter : 0; }
long smashx_lbfgsb_evaluations(const smashx_lbfgsb* o) { return o ? o->total_fev : 0; }
double smashx_lbfgsb_projected_gradient(const smashx_lbfgsb* o) { return o ? o->sbgnrm : 0.0; }

}  // extern "C"
