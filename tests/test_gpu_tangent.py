"""GPU: the tangent-linear model smashx_forward_d (reference base_forward_d, forward_db.f90:10517-10601) against the
golden vectors of the reference's own forward_d (tests/golden/tangent/*.npz, made by make_golden.py::main_tangent
along a direction aligned with the reference gradient) and against the adjoint (scalar product test,
mw_adjoint_test.f90:26-105).  Bar: 1e-6 relative on qsim_d per gauge and on cost_d, relaxed to 3x the reference's own
flag-to-flag noise (cost_d: 1e-5, see below); the scalar product ties tangent and adjoint to fp32 accumulation
accuracy (2e-5)."""
import os
import sys

import numpy as np
import pytest

import golden_util as gu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg  # noqa: E402
from test_gpu_parity import _types  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", mg.TANGENT_CASES)
def test_tangent_vs_reference_golden(name):
    import smash_amd
    g = gu.load(name)
    z = np.load(os.path.join(gu.GOLDEN_DIR, "tangent", name + ".npz"))
    pd, sd = mg.tangent_direction(g)
    setup, mesh, inp, par, sta, out = _types(g)
    par_d, sta_d = smash_amd.ParametersDT.from_dict(mesh, pd), smash_amd.StatesDT.from_dict(mesh, sd)
    out_d = smash_amd.OutputDT(setup, mesh)
    cost, cost_d = smash_amd.forward_d(setup, mesh, inp, par, par_d, inp._bgd[0], par.copy(), sta, sta_d, inp._bgd[1], sta.copy(),
                                       out, out_d)
    for i in range(g.mesh.ng):
        e = gu.rel_l2(out_d.qsim[i], z["qsim_d"][i])
        assert e <= gu.tol(z["noise_qsim_d"][i]), (i, e)
        assert gu.rel_l2(out.qsim[i], g.fwd["qsim"][i]) <= gu.tol(g.noise["qsim"][i], base=2e-6)
    ref = float(z["cost_d"])
    # cost_d: the criteria derivatives (kge above all) are differences of nearly equal fp32 sums, so a discharge that is
    # within 2e-7 of the reference's -- but not bit-identical -- moves them by up to 5e-6; the reference's own tangent
    # and adjoint agree to 1e-6 .. 1e-5 on these cases (scalar product, printed by make_golden.py).  Bar: 1e-5.
    assert abs(cost_d - ref) <= gu.tol(float(z["noise_cost_d"]), base=1e-5) * abs(ref), (cost_d, ref)
    # the primal of a tangent sweep is evaluated in forward_d's re-associated form (last-bit differences in q)
    assert abs(cost - g.fwd["cost"]) <= 1e-4 * abs(g.fwd["cost"]) + 1e-6


@pytest.mark.parametrize("name", ["gr_b_64x64x720_nse", "gr_c_32x32x240_d8_ragged", "vic_a_24x24x240_d8_kge"])
def test_scalar_product(name):
    import smash_amd
    g = gu.load(name)
    setup, mesh, inp, par, sta, out = _types(g)
    sp1, sp2 = smash_amd.scalar_product_test(setup, mesh, inp, par, sta, out)
    assert abs(sp1 - sp2) <= 2e-5 * abs(sp1), (sp1, sp2)


def test_tangent_across_storage_chunks():
    """The tangent sweep cut into storage chunks (state tangents and hlr_d carried from chunk to chunk) is bit-identical
    to the single-chunk sweep."""
    import smash_amd
    from smash_amd.solver import Solver
    g = gu.load("gr_c_32x32x240_d8_ragged")
    pd, sd = mg.tangent_direction(g)
    res = []
    for chunk in (0, 32):
        setup, mesh, inp, par, sta, out = _types(g, chunk_steps=chunk) if chunk else _types(g)
        par_d, sta_d = smash_amd.ParametersDT.from_dict(mesh, pd), smash_amd.StatesDT.from_dict(mesh, sd)
        out_d = smash_amd.OutputDT(setup, mesh)
        _, cost_d = smash_amd.forward_d(setup, mesh, inp, par, par_d, inp._bgd[0], par.copy(), sta, sta_d, inp._bgd[1], sta.copy(), out, out_d)
        res.append((cost_d, out_d.qsim.copy(), out.qsim.copy()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_gradient_test_api():
    """smash_amd.gradient_test = mw_adjoint_test::gradient_test (mw_adjoint_test.f90:108-189): the finite-difference ratio
    Ia of GPU forward sweeps against the GPU adjoint along dk = 1 goes to 1 in the mid range of a (large a: nonlinearity,
    small a: fp32 cancellation), and agrees with the same ratios formed from the CPU oracle's costs and gradient."""
    import smash_amd
    from oracle import pyoracle
    g = gu.load("gr_b_16x16x96_nse_gaps")
    setup, mesh, inp, par, sta, out = _types(g)
    res = smash_amd.gradient_test(setup, mesh, inp, par, sta, out, nstep=12)
    assert [a for a, _ in res] == [2.0 ** -n for n in range(12)]
    e = [x for _, x in res]
    assert all(e[i + 1] < 0.6 * e[i] for i in range(5)), res          # first order in a
    assert min(e) < 1e-2, res                                         # down to the fp32 floor of (Y(k + a dk) - Y(k))
    P = {k: np.asarray(getattr(par, k)) for k in g.params}
    S = {k: np.asarray(getattr(sta, k)) for k in g.states}
    ob = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, P, S, adjoint=True, **g.opts)
    dot = sum(float(np.sum(v.astype(np.float64))) for v in ob["parameters_b"].values())
    for a, e in res[0:6]:
        Pa = {k: (v + np.float32(a)).astype(np.float32) for k, v in P.items()}
        oa = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, Pa, S, **g.opts)
        ia = (oa["cost"] - ob["cost"]) / (a * dot)
        assert abs(abs(ia - 1.0) - e) <= 3e-3, (a, e, ia)
