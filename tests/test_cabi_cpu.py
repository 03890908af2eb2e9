"""CPU: the C-ABI library loads and exports every symbol include/smashx.h declares; host-side logic
that needs no GPU (argument checks, the routing schedule builder via plan creation failing cleanly)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from smash_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "smashx.h")).read()
    declared = set(re.findall(r"\b(smashx_[a-z_]+)\s*\(", hdr))
    L = _lib.lib()
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the product must refuse to run (no CPU path)."""
    import smash_amd
    from smash_amd import _lib, synth
    if _lib.lib().smashx_device_count() > 0:
        pytest.skip("a HIP device is present")
    m = synth.make_mesh(8, 8, ng=1)
    setup = smash_amd.SetupDT(0, 1, structure="gr-a", ntime_step=4)
    mesh = smash_amd.MeshDT.from_synth(setup, m)
    with pytest.raises(smash_amd.SmashxError) as e:
        smash_amd.Solver(setup, mesh)
    assert e.value.code == _lib.E_NODEVICE


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under smash_amd/ may reference it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "smash_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def _bounded_problem(n, seed=3):
    rng = np.random.default_rng(seed)
    tgt = rng.uniform(-0.3, 1.3, n)                    # some optima outside the box: active bounds
    w = 10.0 ** rng.uniform(-2, 2, n)

    def make():
        log = []

        def fg(x):
            d = x - tgt
            f = float(np.sum(w * d * d) + 0.1 * np.sum(np.sin(5 * x)))
            g = 2 * w * d + 0.5 * np.cos(5 * x)
            log.append((f, x.copy()))
            return f, g
        return fg, log
    return make


def test_scipy_lbfgsb_driver_reproduces_scipy_bit_for_bit():
    """smash_amd.optimize._lbfgsb_scipy (SMASHX_LBFGSB=scipy: the host loop around scipy's setulb with vectorised bound arrays)
    against scipy's own fmin_l_bfgs_b on a bounded, badly scaled problem: same iterates, same function values, same number of
    evaluations."""
    from scipy.optimize import fmin_l_bfgs_b
    from smash_amd.optimize import _lbfgsb_scipy
    n = 400
    make = _bounded_problem(n)
    x0 = np.full(n, 0.5)
    for maxiter in (1, 4, 25):
        fa, la = make()
        fb, lb = make()
        xa, va, ia = fmin_l_bfgs_b(fa, x0, m=10, factr=10.0, pgtol=1e-12, bounds=[(0.0, 1.0)] * n, maxiter=maxiter, maxfun=10 * maxiter + 20)
        xb, vb, ib = _lbfgsb_scipy(fb, x0, 10, 10.0, 1e-12, maxiter, 10 * maxiter + 20, None)
        assert len(la) == len(lb) and ia["nit"] == ib["nit"]
        assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(la, lb))
        assert np.array_equal(xa, xb) and va == vb


def test_native_lbfgsb_follows_the_reference_code_iterate_by_iterate():
    """The library's own L-BFGS-B (smashx_lbfgsb_*, the default of smash_amd.optimize) against scipy's build of the lbfgsb.f the
    reference carries, on a bounded, badly scaled problem with active bounds: the same number of iterations and evaluations, every
    trial point and function value equal to rounding of the inner products (the two codes sum in different orders; threads)."""
    from smash_amd.optimize import _lbfgsb_native, _lbfgsb_scipy
    n = 400
    make = _bounded_problem(n)
    x0 = np.full(n, 0.5)
    for maxiter in (1, 4, 12):
        fa, la = make()
        fb, lb = make()
        xa, va, ia = _lbfgsb_scipy(fa, x0, 10, 10.0, 1e-12, maxiter, 10 * maxiter + 20, None)
        xb, vb, ib = _lbfgsb_native(fb, x0, 10, 10.0, 1e-12, maxiter, 10 * maxiter + 20, None)
        assert len(la) == len(lb) and ia["nit"] == ib["nit"] == maxiter and ia["funcalls"] == ib["funcalls"]
        for (f1, x1), (f2, x2) in zip(la, lb):
            assert abs(f1 - f2) <= 1e-11 * abs(f1) and np.max(np.abs(x1 - x2)) <= 1e-10
        assert np.array_equal((xa == 0) | (xa == 1), (xb == 0) | (xb == 1))          # the same active set at the end


@pytest.mark.parametrize("n,m", [(2, 5), (10, 10), (50, 10), (1000, 7)])
def test_native_lbfgsb_converges_like_the_reference_code(n, m):
    """Rosenbrock in a box that cuts the valley ([0, 0.8]^n scaled to the unit box the driver works on) and unconstrained-in-effect
    ([0,1]^n, optimum at the corner 1): both codes stop by the same test within a few evaluations of each other at the same
    minimum."""
    from smash_amd.optimize import _lbfgsb_native, _lbfgsb_scipy
    for scale in (0.8, 1.0):
        def fg(u):
            x = scale * u
            f = float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
            g = np.zeros_like(x)
            g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
            g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
            return f, scale * g
        x0 = np.full(n, 0.3)
        xa, va, ia = _lbfgsb_scipy(fg, x0, m, 1e7, 1e-8, 5000, 8000, None)
        xb, vb, ib = _lbfgsb_native(fg, x0, m, 1e7, 1e-8, 5000, 8000, None)
        assert ia["task"].startswith("CONVERGENCE") and ib["task"].startswith("CONVERGENCE"), (ia["task"], ib["task"])
        assert abs(ia["nit"] - ib["nit"]) <= max(3, ia["nit"] // 50), (ia["nit"], ib["nit"])
        assert abs(va - vb) <= 1e-6 * max(1.0, abs(va)) and np.max(np.abs(xa - xb)) <= 1e-3


def test_native_lbfgsb_hands_over_the_last_iterate_before_it_reports_convergence():
    """lbfgsb.f returns NEW_X for every accepted iterate and tests for convergence on re-entry (mainlb, label 777); the library's
    optimiser does the same: the callback sees the final iterate, the iteration count and the trajectory length equal those of
    scipy's build of the reference code on a run to convergence (they were one short)."""
    from smash_amd.optimize import _lbfgsb_native, _lbfgsb_scipy

    def fg(u):                                            # bounded Rosenbrock, n = 2 and n = 10
        x = 0.8 * u
        f = float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        return f, 0.8 * g
    for n in (2, 10):
        ta, tb = [], []
        xa, va, ia = _lbfgsb_scipy(fg, np.full(n, 0.3), 5, 1e7, 1e-8, 5000, 8000, lambda xk: ta.append(np.copy(xk)))
        xb, vb, ib = _lbfgsb_native(fg, np.full(n, 0.3), 5, 1e7, 1e-8, 5000, 8000, lambda xk: tb.append(np.copy(xk)))
        assert ia["task"].startswith("CONVERGENCE") and ib["task"].startswith("CONVERGENCE")
        assert ia["nit"] == ib["nit"] == len(ta) == len(tb), (ia["nit"], ib["nit"], len(ta), len(tb))
        assert np.array_equal(tb[-1], xb)                 # the callback saw the final iterate
        assert np.max(np.abs(ta[-1] - tb[-1])) <= 1e-6


def test_native_lbfgsb_argument_and_limit_behaviour():
    """Error behaviour of the C entry points and the driver's own stop tests (iteration and evaluation limits, callback per iterate)."""
    import ctypes as C
    from smash_amd import _lib
    from smash_amd.optimize import _lbfgsb_native
    L = _lib.lib()
    h = C.c_void_p()
    lo, up = np.zeros(3), np.ones(3)
    assert L.smashx_lbfgsb_create(0, 5, lo.ctypes.data, up.ctypes.data, 1e7, 1e-8, C.byref(h)) != 0
    assert L.smashx_lbfgsb_create(3, 0, lo.ctypes.data, up.ctypes.data, 1e7, 1e-8, C.byref(h)) != 0
    assert L.smashx_lbfgsb_create(3, 5, up.ctypes.data, lo.ctypes.data, 1e7, 1e-8, C.byref(h)) != 0      # lower > upper
    seen = []
    make = _bounded_problem(50)
    fg, log = make()
    x, f, info = _lbfgsb_native(fg, np.full(50, 0.5), 10, 10.0, 1e-14, 3, 1000, lambda xk: seen.append(xk.copy()))
    assert info["nit"] == 3 and len(seen) == 3 and "ITERATIONS" in info["task"] and np.array_equal(seen[-1], x)
    assert np.all(x >= 0) and np.all(x <= 1) and f == log[-1][0]
    fg, log = make()
    x, f, info = _lbfgsb_native(fg, np.full(50, 0.5), 10, 10.0, 1e-14, 1000, 4, None)
    assert "EVALUATIONS" in info["task"] and info["funcalls"] <= 7
    # a start outside the box is projected onto it first (lbfgsb.f projgr / active)
    fg, log = make()
    _lbfgsb_native(fg, np.full(50, 1.5), 10, 10.0, 1e-14, 1, 50, None)
    assert np.all(log[0][1] == 1.0)


def _native_minimize(fg, x0, lo, up, m=10, factr=1e7, pgtol=1e-8, maxiter=2000):
    """smashx_lbfgsb_* through ctypes with general bounds (None = no bound): (x, f, iterations, evaluations, message)."""
    import ctypes as C
    from smash_amd import _lib
    L = _lib.lib()
    n = len(x0)
    lo = np.array([-np.inf if v is None else v for v in lo], np.float64)
    up = np.array([np.inf if v is None else v for v in up], np.float64)
    h = C.c_void_p()
    _lib.check(L.smashx_lbfgsb_create(n, m, lo.ctypes.data, up.ctypes.data, factr, pgtol, C.byref(h)))
    x = np.array(x0, np.float64)
    g = np.zeros(n)
    f, task, nit, nfev = 0.0, C.c_int(0), 0, 0
    while True:
        _lib.check(L.smashx_lbfgsb_step(h, x.ctypes.data, float(f), g.ctypes.data, C.byref(task)))
        if task.value == 1:
            f, g = fg(x.copy())
            g = np.ascontiguousarray(g, np.float64)
            nfev += 1
        elif task.value == 2:
            nit += 1
            if nit >= maxiter:
                break
        else:
            break
    msg = L.smashx_lbfgsb_message(h).decode()
    L.smashx_lbfgsb_destroy(h)
    return x, f, nit, nfev, msg


@pytest.mark.parametrize("kind", ["none", "lower", "upper", "mixed"])
def test_native_lbfgsb_with_general_bounds_against_the_reference_code(kind):
    """Variables without bounds, with one bound and with two (nbd = 0, 1, 3, 2 of lbfgsb.f) on the extended Rosenbrock function:
    the library's L-BFGS-B against scipy's build of the reference's code -- same minimum, iteration and evaluation counts within a
    few (the unit first step of the unconstrained case, the Cauchy search with infinite breakpoints)."""
    from scipy.optimize import fmin_l_bfgs_b
    n = 12

    def fg(x):
        f = float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        return f, g
    lo = {"none": [None] * n, "lower": [0.3 if i % 3 == 0 else None for i in range(n)], "upper": [None] * n,
          "mixed": [(-0.5 if i % 2 else None) for i in range(n)]}[kind]
    up = {"none": [None] * n, "lower": [None] * n, "upper": [0.7 if i % 4 == 1 else None for i in range(n)],
          "mixed": [(0.6 if i % 3 == 0 else None) for i in range(n)]}[kind]
    x0 = np.linspace(-0.4, 0.9, n)
    xs, fs, d = fmin_l_bfgs_b(fg, x0, m=10, factr=1e7, pgtol=1e-8, bounds=list(zip(lo, up)), maxiter=2000, maxfun=5000)
    xn, fn, nit, nfev, msg = _native_minimize(fg, x0, lo, up)
    assert msg.startswith("CONVERGENCE"), msg
    assert abs(fn - fs) <= 1e-6 * max(1.0, abs(fs)) and np.max(np.abs(xn - xs)) <= 2e-3, (fn, fs, float(np.max(np.abs(xn - xs))))
    assert abs(nit - d["nit"]) <= max(4, d["nit"] // 8) and abs(nfev - d["funcalls"]) <= max(5, d["funcalls"] // 8), (nit, d["nit"], nfev, d["funcalls"])
    lo_a = np.array([-np.inf if v is None else v for v in lo]); up_a = np.array([np.inf if v is None else v for v in up])
    assert np.all(xn >= lo_a) and np.all(xn <= up_a)
