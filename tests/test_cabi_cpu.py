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


def test_lbfgsb_driver_reproduces_scipy_bit_for_bit():
    """smash_amd.optimize._lbfgsb_box (the host L-BFGS-B loop with vectorised bound arrays) against scipy's own
    fmin_l_bfgs_b on a bounded, badly scaled problem: same iterates, same function values, same number of evaluations."""
    from scipy.optimize import fmin_l_bfgs_b
    from smash_amd.optimize import _lbfgsb_box
    rng = np.random.default_rng(3)
    n = 400
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
    x0 = np.full(n, 0.5)
    for maxiter in (1, 4, 25):
        fa, la = make()
        fb, lb = make()
        xa, va, ia = fmin_l_bfgs_b(fa, x0, m=10, factr=10.0, pgtol=1e-12, bounds=[(0.0, 1.0)] * n, maxiter=maxiter, maxfun=10 * maxiter + 20)
        xb, vb, ib = _lbfgsb_box(fb, x0, 10, 10.0, 1e-12, maxiter, 10 * maxiter + 20, None)
        assert len(la) == len(lb) and ia["nit"] == ib["nit"]
        assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(la, lb))
        assert np.array_equal(xa, xb) and va == vb
