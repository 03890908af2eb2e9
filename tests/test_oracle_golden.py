"""CPU: the plain-C oracle (oracle/smash_oracle.c) against the golden vectors produced by the
unmodified reference Fortran (tests/golden/make_golden.py).  The oracle replays the reference's
operation order with the same libm, so the bar here is BIT-EXACT, forward and adjoint."""
import numpy as np
import pytest

import golden_util as gu
from oracle import pyoracle


@pytest.mark.parametrize("name", gu.names())
def test_oracle_forward_bit_exact(name):
    g = gu.load(name)
    o = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, **g.opts)
    assert np.array_equal(o["qsim"], g.fwd["qsim"])
    assert o["cost"] == g.fwd["cost"] and o["cost_jobs"] == g.fwd["cost_jobs"] and o["cost_jreg"] == g.fwd["cost_jreg"]
    for k in gu.STRUCT_STATES[g.structure]:
        assert np.array_equal(o["fstates"][k], g.fwd["fstates"][k]), k
        assert np.array_equal(o["states"][k], g.fwd["states"][k]), k
    for k in gu.STRUCT_PARAMS[g.structure]:
        assert np.array_equal(o["parameters"][k], g.fwd["parameters"][k]), k


@pytest.mark.parametrize("name", gu.names())
def test_oracle_adjoint_bit_exact(name):
    g = gu.load(name)
    o = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, adjoint=True, **g.opts)
    assert np.array_equal(o["qsim"], g.adj["qsim"])
    assert o["cost"] == g.adj["cost"]
    for k in g.adj["parameters_b"]:
        assert np.array_equal(o["parameters_b"][k], g.adj["parameters_b"][k]), k
    for k in g.adj["states_b"]:
        assert np.array_equal(o["states_b"][k], g.adj["states_b"][k]), k


def test_oracle_gradient_taylor():
    """Restatement of the reference's gradient_test idea (mw_adjoint_test.f90:108-189): the adjoint
    gradient agrees with a central finite difference of the cost along a random direction."""
    g = gu.load("gr_a_12x12x48_nse_cold")
    o = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, adjoint=True)
    rng = np.random.default_rng(0)
    act = g.mesh.active_cell == 1
    for k in ("cp", "lr"):   # cft/exc sensitivities are below fp32 FD noise on this short case
        d = np.where(act, rng.standard_normal(act.shape), 0.0).astype(np.float32)
        eps = 3e-3 * float(np.mean(g.params[k]))
        pp = dict(g.params); pm = dict(g.params)
        pp[k] = np.asfortranarray(g.params[k] + eps * d); pm[k] = np.asfortranarray(g.params[k] - eps * d)
        cp_ = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, pp, g.states)["cost"]
        cm_ = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, pm, g.states)["cost"]
        fd = (cp_ - cm_) / (2 * eps)
        ad = float(np.sum(o["parameters_b"][k].astype(np.float64) * d))
        assert abs(fd - ad) <= 2e-2 * max(abs(fd), abs(ad)) + 1e-7, (k, fd, ad)


def test_tangent_oracle_is_bit_identical_to_reference_forward_d():
    """orc_forward_d (oracle/smash_oracle_d.c) against the reference's own forward_d (tests/golden/tangent/*.npz,
    make_golden.py::main_tangent): cost_d and qsim_d bit for bit on every case."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    from oracle import pyoracle
    for name in mg.TANGENT_CASES:
        g = gu.load(name)
        z = np.load(os.path.join(gu.GOLDEN_DIR, "tangent", name + ".npz"))
        pd, sd = mg.tangent_direction(g)
        r = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, params_d=pd, states_d=sd, **g.opts)
        assert np.float32(r["cost_d"]) == z["cost_d"], name
        assert np.array_equal(r["qsim_d"], z["qsim_d"]), name


def test_python_lbfgsb_loop_reproduces_reference_trajectory():
    """The host loop of smash_amd.optimize_lbfgsb (the reference's control-vector order and settings) fed by the CPU oracle
    instead of the GPU, under scipy's build of lbfgsb.f and under the library's own L-BFGS-B: the cost after 1 and 3 iterations
    equals the reference's own optimize_lbfgsb bit for bit (tests/golden/lbfgsb)."""
    import os
    from scipy.optimize import fmin_l_bfgs_b
    from oracle import pyoracle
    from oracle.refbind import GLB_P, GLB_S, GUB_P, GUB_S
    from smash_amd import synth
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "opt_gr_b_24x24x120.npz"))
    g = gu.load("gr_b_24x24x120_norm_jreg")
    mesh, op = g.mesh, z["optim_parameters"]
    nrm = lambda d, lb, ub, names: {k: np.asfortranarray(((d[k] - lb[i]) / (ub[i] - lb[i])).astype(np.float32)) for i, k in enumerate(names)}
    Pn = nrm(synth.make_parameters(24, 24), GLB_P, GUB_P, synth.PARAM_NAMES)
    Sn = nrm(synth.make_states(24, 24, warm=True), GLB_S, GUB_S, synth.STATE_NAMES)
    fields = [k for i, k in enumerate(synth.PARAM_NAMES) if op[i] > 0]
    cr = np.argwhere((mesh.active_cell == 1).T)
    cols, rows = cr[:, 0], cr[:, 1]
    m = len(rows)
    kw = dict(params_bgd=Pn, states_bgd=Sn, denormalize_forward=True, optim_parameters=op, jobs_fun=("nse",), wjobs_fun=(1.0,))

    def unpack(x):
        out = {k: v.copy() for k, v in Pn.items()}
        for j, k in enumerate(fields):
            out[k][rows, cols] = x[j * m:(j + 1) * m].astype(np.float32)
        return out

    def fg(x):
        r = pyoracle.run("gr-b", mesh, g.dt, g.prcp, g.pet, z["qobs"], unpack(x), Sn, adjoint=True, **kw)
        return float(np.float32(r["cost"])), np.concatenate([r["parameters_b"][k][rows, cols].astype(np.float64) for k in fields])

    x0 = np.concatenate([Pn[k][rows, cols].astype(np.float64) for k in fields])
    for it in (1, 3):
        x, f, d = fmin_l_bfgs_b(fg, x0, m=10, factr=10.0, pgtol=1e-12, bounds=[(0.0, 1.0)] * len(x0), maxiter=it, maxfun=100)
        r = pyoracle.run("gr-b", mesh, g.dt, g.prcp, g.pet, z["qobs"], unpack(x), Sn, **kw)
        assert np.float32(r["cost"]) == z["costs"][it], (it, r["cost"], z["costs"][it])
        # the library's own L-BFGS-B (the default of smash_amd.optimize) from the same start: the same iterates to rounding of its
        # inner products (it sums in a different order than lbfgsb.f; measured 7e-18), the same fp32 cost, the same evaluations
        from smash_amd.optimize import _lbfgsb_native
        xn, fn, dn = _lbfgsb_native(fg, x0, 10, 10.0, 1e-12, it, 100, None)
        rn = pyoracle.run("gr-b", mesh, g.dt, g.prcp, g.pet, z["qobs"], unpack(xn), Sn, **kw)
        print(it, "native", float(rn["cost"]), "reference", float(z["costs"][it]), "max |dx|", float(np.max(np.abs(xn - x))), dn["funcalls"], d["funcalls"])
        assert dn["nit"] == d["nit"] and dn["funcalls"] == d["funcalls"]
        assert np.max(np.abs(xn - x)) <= 1e-12 and np.float32(rn["cost"]) == z["costs"][it]


def test_cance_fixture_is_pinned_by_the_values_the_reference_publishes():
    """The reference's own test data is the Cance catchment (smash/tests/*, baseline.hdf5 unreadable here); what its
    sources print about that case pins the fixture (read from the dataset files by tests/golden/cance_io.py and run
    through the flang-built reference) and the oracle:
      * mesh: 28 x 28, 383 active cells, gauge 0 at (20, 27)               (dataset/load.py:58, core/model.py:775-777)
      * run with Model() defaults: qsim[0, :3]                             (core/model.py:475-477)
      * run at the uniform SBS optimum (model.py:784): qsim[0, :3], qsim[0, -3:]   (core/model.py:767-769)
    The last three values of the DEFAULT run printed there (20.9165, 20.7623, 20.6105) are 2.1 % above what the current
    sources give on the current dataset (20.4842, 20.3349, 20.1874; flang -O2 and -O3 agree to 2e-7) while the optimum run
    agrees to 4e-5: that docstring predates the defaults / forcing treatment of this revision and is not asserted."""
    g = gu.load("gr_a_cance_28x28x1440")
    assert (g.mesh.nrow, g.mesh.ncol, g.mesh.nac, g.nt) == (28, 28, 383, 1440)
    assert tuple(np.asarray(g.mesh.gauge_pos)[0]) == (20, 27)
    q = g.fwd["qsim"][0]
    np.testing.assert_allclose(q[:3], [5.7140866e-04, 4.7018618e-04, 3.5345653e-04], rtol=2e-7)
    np.testing.assert_allclose(q[-3:], [1.9017689e+01, 1.8781073e+01, 1.8549627e+01], rtol=1e-4)
    # Model() defaults (mwd_parameters.f90:150-167, mwd_states.f90:117-126, lr = dt * 5 / 3600) through the oracle
    pv = dict(ci=1e-6, cp=200.0, beta=1000.0, cft=500.0, cst=500.0, alpha=0.9, exc=0.0, lr=5.0)
    P = {k: (np.full_like(v, pv[k]) if k in pv else v) for k, v in g.params.items()}
    o = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, P, g.states, **g.opts)
    np.testing.assert_allclose(o["qsim"][0][:3], [1.9826449e-03, 1.3466686e-07, 6.7618025e-12], rtol=2e-7)


def test_default_build_bars_are_capped_and_the_unasserted_outputs_are_pinned():
    """No default-build PARITY bar is above 1e-4 (golden_util.CAP).  The outputs whose reference self-noise puts 3 x noise beyond that
    keep that finite bar as a sanity check (a gross regression of the default build must not pass) and have their parity asserted in the
    exact-libm build only (bit-identity, tests/test_gpu_exact.py); the list is pinned here so that it cannot grow silently."""
    un = {}
    for name in gu.names():
        g = gu.load(name)
        for v in list(g.noise["qsim"]) + [g.noise["cost"]] + [x for grp in ("fstates", "parameters_b", "states_b") for x in g.noise[grp].values()]:
            b = gu.tol(v)
            assert np.isfinite(b) and b >= 1e-6 and (b <= gu.CAP or gu.sanity_only(v))
        u = gu.unasserted_outputs(g)
        if u:
            un[name] = u
    assert un == {
        "gr_a_12x12x48_nse_cold": ["qsim[0]", "qsim[1]", "cost", "fstates.hlr", "parameters_b.cp", "parameters_b.cft", "parameters_b.exc",
                                   "parameters_b.lr", "states_b.hp", "states_b.hft", "states_b.hlr"],
        "gr_d_12x12x48_rmse_kge2_start": ["parameters_b.cp", "states_b.hp"],
        "vic_a_16x16x96_nse_gaps": ["fstates.hlr", "parameters_b.b", "parameters_b.cusl1", "parameters_b.cusl2"],
        "vic_a_24x24x240_d8_kge": ["parameters_b.b", "parameters_b.lr", "states_b.hlr"],
    }, un
    assert sum(len(v) for v in un.values()) == 20       # of 318 outputs


def test_fp64_truth_oracle_agrees_with_the_fp32_oracle_to_rounding():
    """oracle/liboracle64.so (the same statements in double) is the common truth of tools/accuracy_report.py: on a well-conditioned
    fixture the fp32 oracle (= the reference, bit for bit) must sit within fp32 rounding accumulation of it, forward and adjoint."""
    g = gu.load("gr_b_16x16x96_nse_gaps")
    r32 = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, adjoint=True, **g.opts)
    r64 = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, adjoint=True, fp64=True, **g.opts)
    assert r64["qsim"].dtype == np.float64
    assert gu.rel_l2(r32["qsim"], r64["qsim"]) < 5e-6
    assert abs(r32["cost"] - r64["cost"]) < 1e-4 * abs(r64["cost"])
    for k in ("cp", "cft", "exc", "lr"):
        e = gu.rel_l2(r32["parameters_b"][k], r64["parameters_b"][k])
        assert 1e-8 < e < 1e-3, (k, e)        # close, and not identical: the fp32 program's own rounding error is what is measured


def test_reference_optimize_lbfgsb_on_the_library_setulb():
    """fortran/smashx_setulb.f90: `setulb` of the reference's lbfgsb.f (argument list, task strings, isave(30) / isave(34) /
    dsave(13)) on top of the library's own L-BFGS-B.  oracle/_ref/libsmash_ref_lbfgsb.so is the unmodified reference with only that
    one symbol replaced (oracle/ref/build_ref.sh): its own mw_optimize::optimize_lbfgsb loop, all on the CPU, must walk the golden
    cost trajectory (tests/golden/lbfgsb, made with the reference's lbfgsb.f) -- bit for bit, 0 to 4 iterations."""
    import os
    from oracle import refbind
    from smash_amd import synth
    if not refbind.available("ref_lbfgsb"):
        pytest.skip("oracle/_ref/libsmash_ref_lbfgsb.so not built")
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "opt_gr_b_24x24x120.npz"))
    g = gu.load("gr_b_24x24x120_norm_jreg")
    P, S = synth.make_parameters(24, 24), synth.make_states(24, 24, warm=True)
    for it, ref in zip(z["maxiters"], z["costs"]):
        r = refbind.run("gr-b", g.mesh, g.dt, g.prcp, g.pet, z["qobs"], P, S, optimize_maxiter=int(it),
                        optim_parameters=z["optim_parameters"], jobs_fun=("nse",), wjobs_fun=(1.0,), fast="ref_lbfgsb")
        assert np.float32(r["cost"]) == np.float32(ref), (int(it), r["cost"], float(ref))


def test_oracle_is_bit_identical_to_the_reference_on_a_real_river_network():
    """The oracle's pin so far is the 18 golden cases (grids up to 64 x 64).  Here: the largest basin of the reference's own 1-km D8 raster
    of France (139 742 cells, all eight codes; tests/golden/mesh/france_d8.npz -> synth.make_mesh_france), 48 steps, gr-b -- the compiled
    reference (oracle/_ref/libsmash_ref.so, built in place by oracle/ref/build_ref.sh) against the C restatement, forward and adjoint,
    bit for bit.  tests/test_gpu_parity.py::test_real_river_network_vs_oracle then holds the GPU to the oracle on the same network."""
    from oracle import refbind
    from smash_amd import synth
    if not refbind.available():
        pytest.skip("oracle/_ref/libsmash_ref.so not built (needs /root/reference and flang)")
    m = synth.make_mesh_france(1, ng=4)
    nt = 48
    prcp, pet = synth.dense_forcing(m, nt, gap_per_million=2000)
    P, S = synth.make_parameters(m.nrow, m.ncol), synth.make_states(m.nrow, m.ncol, warm=True)
    qobs = np.asfortranarray(np.abs(np.random.default_rng(5).standard_normal((m.ng, nt))).astype(np.float32) + 0.1)
    ref = refbind.run("gr-b", m, 3600.0, prcp, pet, qobs, P, S, adjoint=True)
    orc = pyoracle.run("gr-b", m, 3600.0, prcp, pet, qobs, P, S, adjoint=True)
    assert np.array_equal(ref["qsim"], orc["qsim"]) and np.float32(ref["cost"]) == np.float32(orc["cost"])
    for k in gu.STRUCT_PARAMS["gr-b"]:
        assert np.array_equal(ref["parameters_b"][k], orc["parameters_b"][k]), k
    for k in gu.STRUCT_STATES["gr-b"]:
        assert np.array_equal(ref["states_b"][k], orc["states_b"][k]), k
