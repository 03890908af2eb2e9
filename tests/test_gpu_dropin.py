"""GPU: the reference's OWN Fortran boundary running on the HIP path.

oracle/_ref/libsmash_dropin.so is the unmodified reference solver with only base_forward / base_forward_b /
base_forward_d replaced by the ISO_C_BINDING shim fortran/smashx_dropin.f90 (built by oracle/ref/build_ref.sh in the build
container; the prebuilt library travels to the GPU box).  The same bind(C) driver that produced the golden
vectors (oracle/ref/ref_capi.f90 -> mw_forward::forward / forward_b, mw_optimize::optimize_lbfgsb) is called
on it, so these tests exercise exactly what a maintainer gets by relinking the reference against libsmashx.
"""
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import refbind

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refbind.available("dropin"), reason="oracle/_ref/libsmash_dropin.so not built")]

CASES = ["gr_a_12x12x48_nse", "gr_b_16x16x96_nse_gaps", "gr_c_16x16x96_kge_se_log_mask", "gr_d_48x48x480_nse",
         "vic_a_16x16x96_nse_gaps"]


@pytest.mark.parametrize("name", CASES)
def test_reference_forward_b_through_dropin(name):
    g = gu.load(name)
    f = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, fast="dropin", **g.opts)
    b = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, adjoint=True, fast="dropin", **g.opts)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(f["qsim"][i], g.fwd["qsim"][i]) <= gu.tol(g.noise["qsim"][i])
        assert gu.rel_l2(b["qsim"][i], g.adj["qsim"][i]) <= gu.tol(g.noise["qsim"][i])
    assert abs(f["cost"] - g.fwd["cost"]) <= gu.tol_cost(g.noise["cost"], g.fwd["cost"])
    for k in gu.STRUCT_STATES[g.structure]:
        assert gu.rel_l2(f["fstates"][k], g.fwd["fstates"][k]) <= gu.tol_fstate(k, g.noise["fstates"][k]), k
        assert gu.rel_l2(b["states_b"][k], g.adj["states_b"][k]) <= gu.tol(g.noise["states_b"][k]), k
    for k in gu.STRUCT_PARAMS[g.structure]:
        assert gu.rel_l2(b["parameters_b"][k], g.adj["parameters_b"][k]) <= gu.tol(g.noise["parameters_b"][k]), k


def test_sparse_storage_through_dropin():
    g = gu.load("gr_c_16x16x96_kge_se_log_mask")
    f = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, fast="dropin", sparse_storage=True, **g.opts)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(f["qsim"][i], g.fwd["qsim"][i]) <= gu.tol(g.noise["qsim"][i])


def test_reference_lbfgsb_loop_through_dropin():
    """SURVEY row f1 / BASELINE config #4 in miniature: the reference's optimize_lbfgsb (host, fp64 L-BFGS-B,
    mw_optimize.f90:484-676) driving GPU forward / forward_b sweeps through the drop-in boundary.  The iterates
    are a chaotic function of the gradient's last bits, so the check is on the cost trajectory: identical at
    iteration 0, and the same decrease (within 2 %) after 4 iterations as the all-CPU reference."""
    from smash_amd import synth
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "opt_gr_b_24x24x120.npz"))
    g = gu.load("gr_b_24x24x120_norm_jreg")
    P = synth.make_parameters(24, 24)
    S = synth.make_states(24, 24, warm=True)
    costs = []
    for it in z["maxiters"]:
        r = refbind.run("gr-b", g.mesh, g.dt, g.prcp, g.pet, z["qobs"], P, S, optimize_maxiter=int(it),
                        optim_parameters=z["optim_parameters"], jobs_fun=("nse",), wjobs_fun=(1.0,), fast="dropin")
        costs.append(r["cost"])
    ref = z["costs"]
    assert abs(costs[0] - ref[0]) <= 3e-7 + 1e-5 * abs(ref[0]), (costs, ref)
    assert costs[-1] < 0.8 * costs[0]
    assert abs(costs[-1] - ref[-1]) <= 0.02 * abs(ref[0]), (costs, ref)


def test_reference_calibration_entirely_on_the_library():
    """The reference's optimize_lbfgsb with BOTH halves replaced: GPU forward / forward_b sweeps through fortran/smashx_dropin.f90 and
    `setulb` through fortran/smashx_setulb.f90 (the library's own L-BFGS-B): oracle/_ref/libsmash_dropin_lbfgsb.so.  Against the same
    loop on the reference's lbfgsb.f (libsmash_dropin.so): the sweeps are the same deterministic kernels, so the costs agree to the
    rounding of the optimiser's inner products."""
    from smash_amd import synth
    if not refbind.available("dropin_lbfgsb"):
        pytest.skip("oracle/_ref/libsmash_dropin_lbfgsb.so not built")
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "opt_gr_b_24x24x120.npz"))
    g = gu.load("gr_b_24x24x120_norm_jreg")
    P = synth.make_parameters(24, 24)
    S = synth.make_states(24, 24, warm=True)
    for it in (1, 4):
        kw = dict(optimize_maxiter=it, optim_parameters=z["optim_parameters"], jobs_fun=("nse",), wjobs_fun=(1.0,))
        a = refbind.run("gr-b", g.mesh, g.dt, g.prcp, g.pet, z["qobs"], P, S, fast="dropin", **kw)
        b = refbind.run("gr-b", g.mesh, g.dt, g.prcp, g.pet, z["qobs"], P, S, fast="dropin_lbfgsb", **kw)
        assert abs(a["cost"] - b["cost"]) <= 1e-6 * abs(a["cost"]), (it, a["cost"], b["cost"])
        for k in ("cp", "cft", "exc", "lr"):
            if k in a.get("parameters", {}):
                assert np.max(np.abs(a["parameters"][k] - b["parameters"][k])) <= 1e-4 * np.max(np.abs(a["parameters"][k])), k


def test_reference_lbfgsb_on_cance_through_dropin():
    """The user guide's distributed calibration on the real Cance data (real_case_cance.rst:470-552): the reference's
    optimize_lbfgsb over cp, cft, exc, lr from the uniform SBS optimum, GPU sweeps through the drop-in, against the
    all-CPU reference's cost trajectory (tests/golden/lbfgsb/opt_gr_a_cance.npz)."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "opt_gr_a_cance.npz"))
    g = gu.load("gr_a_cance_28x28x1440")
    costs = []
    for it in z["maxiters"]:
        r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, optimize_maxiter=int(it),
                        optim_parameters=z["optim_parameters"], fast="dropin", **g.opts)
        costs.append(r["cost"])
    ref = z["costs"]
    assert abs(costs[0] - ref[0]) <= 3e-7 + 1e-5 * abs(ref[0]), (costs, ref)
    assert costs[-1] < costs[0]
    assert abs(costs[-1] - ref[-1]) <= 0.02 * abs(ref[0]), (costs, ref)


def test_reference_sbs_on_cance_through_dropin():
    """The user guide's first calibration (real_case_cance.rst:396-430): the reference's optimize_sbs (uniform cp, cft, exc,
    lr from the Model() defaults, nse at the downstream gauge, maxiter 2) with every forward sweep on the GPU, against the
    all-CPU reference (tests/golden/lbfgsb/sbs_gr_a_cance.npz: J = 0.6996 -> 0.1097 -> 0.0439; the guide, on its revision of
    the data, prints 0.6774 -> 0.1300 -> 0.0437)."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "sbs_gr_a_cance.npz"))
    g = gu.load("gr_a_cance_28x28x1440")
    pv = dict(ci=1e-6, cp=200.0, beta=1000.0, cft=500.0, cst=500.0, alpha=0.9, exc=0.0, lr=5.0)
    P = {k: (np.full_like(v, pv[k]) if k in pv else v) for k, v in g.params.items()}
    costs = []
    for it in z["maxiters"]:
        r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, P, g.states, optimize_sbs_maxiter=int(it),
                        optim_parameters=z["optim_parameters"], fast="dropin", **g.opts)
        costs.append(r["cost"])
    ref = z["costs"]
    assert abs(costs[0] - ref[0]) <= 3e-7 + 1e-5 * abs(ref[0]), (costs, ref)
    for a, b in zip(costs[1:], ref[1:]):
        assert abs(a - b) <= 0.02 * abs(b), (costs, ref)
    for k in ("cp", "cft", "exc", "lr"):
        assert abs(float(r["parameters"][k][20, 27]) - float(z["final_" + k])) <= 0.02 * abs(float(z["final_" + k])) + 1e-3, k


@pytest.mark.parametrize("name", ["gr_b_16x16x96_nse_gaps", "gr_b_24x24x120_norm_jreg"])
def test_reference_forward_d_through_dropin(name):
    """mw_forward::forward_d of the reference (mw_forward.f90:70-97) on the GPU tangent sweep: base_forward_d replaced by
    the shim; against the golden vectors of the all-CPU reference (tests/golden/tangent)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    g = gu.load(name)
    z = np.load(os.path.join(gu.GOLDEN_DIR, "tangent", name + ".npz"))
    pd, sd = mg.tangent_direction(g)
    r = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, params_d=pd, states_d=sd, fast="dropin", **g.opts)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(r["qsim_d"][i], z["qsim_d"][i]) <= gu.tol(z["noise_qsim_d"][i]), i
    ref = float(z["cost_d"])
    assert abs(r["cost_d"] - ref) <= 1e-5 * abs(ref), (r["cost_d"], ref)


def _hyper_cases():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    return mg.HYPER_CASES


@pytest.mark.parametrize("name,mapping", _hyper_cases())
def test_reference_hyper_forward_through_dropin(name, mapping):
    """mw_forward::hyper_forward / hyper_forward_b (mw_forward.f90:99-152): the descriptor -> parameter mapping and its
    adjoint stay the reference's host code, the time loop, cost and their adjoints run on the GPU (base_hyper_forward /
    base_hyper_forward_b in fortran/smashx_dropin.f90); against the golden vectors of the all-CPU reference."""
    import make_golden as mg
    from smash_amd import synth
    g = gu.load(name)
    z = np.load(os.path.join(gu.GOLDEN_DIR, "hyper", f"{name}__{mapping}.npz"))
    desc, hp, hs = mg.hyper_inputs(g, mapping)
    kw = dict(descriptor=desc, hyper_params=hp, hyper_states=hs, mapping=mapping, fast="dropin",
              **{k: v for k, v in g.opts.items() if k in ("jobs_fun", "wjobs_fun", "optimize_start_step", "wgauge")})
    f = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, **kw)
    b = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, adjoint=True, **kw)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(f["qsim"][i], z["fwd_qsim"][i]) <= gu.tol(z["noise_qsim"][i]), i
    assert abs(f["cost"] - float(z["fwd_cost"])) <= gu.tol_cost(float(z["noise_cost"]), float(z["fwd_cost"]))
    assert abs(b["cost"] - float(z["adj_cost"])) <= gu.tol_cost(float(z["noise_cost"]), float(z["adj_cost"]))
    for k in synth.PARAM_NAMES:          # the mapped fields come from the reference's own host code: identical
        assert np.array_equal(f["parameters"][k], z["fwd_p_" + k]), k
    for k in gu.STRUCT_STATES[g.structure]:   # base_hyper_forward leaves the states at their final values
        assert gu.rel_l2(f["states"][k], z["fwd_s_" + k]) <= gu.tol_fstate(k, 1e-5), k
    for k in gu.STRUCT_PARAMS[g.structure]:
        assert gu.rel_l2(b["hyper_parameters_b"][k], z["adj_hp_b_" + k]) <= gu.tol(float(z["noise_hp_b_" + k]), base=5e-6), k
    for k in gu.STRUCT_STATES[g.structure]:
        assert gu.rel_l2(b["hyper_states_b"][k], z["adj_hs_b_" + k]) <= gu.tol(float(z["noise_hs_b_" + k]), base=5e-6), k


@pytest.mark.parametrize("name,mapping", _hyper_cases())
def test_reference_hyper_forward_d_through_dropin(name, mapping):
    """mw_forward::hyper_forward_d (mw_forward.f90:154-181): tangent of the mapping in the reference's host code, tangent
    sweep and cost tangent on the GPU (base_hyper_forward_d in fortran/smashx_dropin.f90).  Bars as in
    test_gpu_tangent.py: qsim_d noise-aware 1e-6, cost_d 1e-5 (reference-order sums of cancelling terms)."""
    import make_golden as mg
    g = gu.load(name)
    z = np.load(os.path.join(gu.GOLDEN_DIR, "hyper", f"{name}__{mapping}.npz"))
    desc, hp, hs = mg.hyper_inputs(g, mapping)
    adj = dict(hyper_parameters_b={k: z["adj_hp_b_" + k] for k in hp}, hyper_states_b={k: z["adj_hs_b_" + k] for k in hs})
    hd, sd = mg.hyper_direction(adj)
    kw = dict(descriptor=desc, hyper_params=hp, hyper_states=hs, mapping=mapping, fast="dropin", hyper_params_d=hd, hyper_states_d=sd,
              **{k: v for k, v in g.opts.items() if k in ("jobs_fun", "wjobs_fun", "optimize_start_step", "wgauge")})
    t = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, **kw)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(t["qsim_d"][i], z["tan_qsim_d"][i]) <= gu.tol(z["noise_tan_qsim_d"][i]), i
    ref = float(z["tan_cost_d"])
    assert abs(t["cost_d"] - ref) <= max(1e-5, 3 * float(z["noise_tan_cost_d"])) * abs(ref), (t["cost_d"], ref)
    # and the adjoint / tangent pair of the whole chain agree: cost_d = <gradient, direction>
    dot = sum(float(np.dot(z["adj_hp_b_" + k].astype(np.float64), hd[k])) for k in hp) + \
          sum(float(np.dot(z["adj_hs_b_" + k].astype(np.float64), sd[k])) for k in hs)
    assert abs(t["cost_d"] - dot) <= 2e-4 * abs(dot), (t["cost_d"], dot)


def test_dropin_does_not_serve_stale_forcing_or_mesh():
    """Two models of the same shape in one process: the shim caches its plan (forcing resident in HBM) across calls, and the
    second model's arrays can land at the freed addresses of the first.  The cache key therefore carries dt, dx, a hash of the
    mesh arrays and a strided hash of the forcing recomputed on every call (fortran/smashx_dropin.f90)."""
    g = gu.load("gr_b_16x16x96_nse_gaps")
    a = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, fast="dropin", **g.opts)
    wet = np.asfortranarray(np.where(g.prcp < 0, g.prcp, g.prcp * np.float32(1.5)).astype(np.float32))
    b = refbind.run(g.structure, g.mesh, g.dt, wet, g.pet, g.qobs, g.params, g.states, fast="dropin", **g.opts)
    ref = refbind.run(g.structure, g.mesh, g.dt, wet, g.pet, g.qobs, g.params, g.states, **g.opts)      # the all-CPU reference
    assert not np.array_equal(a["qsim"], b["qsim"])
    for i in range(g.mesh.ng):
        assert gu.rel_l2(b["qsim"][i], ref["qsim"][i]) <= 1e-5
    c = refbind.run(g.structure, g.mesh, 1800.0, wet, g.pet, g.qobs, g.params, g.states, fast="dropin", **g.opts)   # another dt
    refc = refbind.run(g.structure, g.mesh, 1800.0, wet, g.pet, g.qobs, g.params, g.states, **g.opts)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(c["qsim"][i], refc["qsim"][i]) <= 1e-5


def test_dropin_keeps_reader_form_forcing_compact(capfd, monkeypatch):
    """A Model whose forcing came from the reference's reader with daily inter-annual PET (setup%daily_interannual_pet, hourly steps,
    prcp_conversion_factor) reaches the shim with setup fields that say so; the shim then asks the library for the lossless compact
    layout (rain counts x factor, daily PET x RATIO_PET_HOURLY).  Same discharge and gradients as without it; and forcing that is NOT
    of that form (the synthetic generator's own diurnal weights) silently stays in fp32 rows with the same results."""
    from smash_amd.solver import RATIO_PET_HOURLY as R
    monkeypatch.setenv("SMASHX_VERBOSE", "1")
    g = gu.load("gr_b_16x16x96_nse_gaps")
    # put the fixture's PET on the reader's form: daily value x RATIO_PET_HOURLY, first step = 01:00
    pet = g.pet.copy(order="F")
    for d in range((g.nt + 1 + 23) // 24):
        ts = list(range(max(0, d * 24 - 1), min(g.nt, (d + 1) * 24 - 1)))
        daily = np.float32(1.0) + pet[:, :, ts].max(axis=2) * np.float32(8.0)
        for t in ts:
            pet[:, :, t] = daily * R[(t + 1) % 24]
    ref = refbind.run(g.structure, g.mesh, g.dt, g.prcp, pet, g.qobs, g.params, g.states, adjoint=True, **g.opts)      # all-CPU reference
    capfd.readouterr()
    for form in (True, False):
        b = refbind.run(g.structure, g.mesh, g.dt, g.prcp, pet, g.qobs, g.params, g.states, adjoint=True, fast="dropin", reader_form=form, **g.opts)
        err = capfd.readouterr().err
        assert ("forcing resident as compact" in err) == form and ("forcing resident as fp32 rows" in err) == (not form), err
        for i in range(g.mesh.ng):
            assert gu.rel_l2(b["qsim"][i], ref["qsim"][i]) <= 1e-5
        for k in gu.STRUCT_PARAMS[g.structure]:
            assert gu.rel_l2(b["parameters_b"][k], ref["parameters_b"][k]) <= 1e-4, k
        if form:
            keep = b
        else:
            assert np.array_equal(keep["qsim"], b["qsim"]) and all(np.array_equal(keep["parameters_b"][k], b["parameters_b"][k]) for k in gu.STRUCT_PARAMS[g.structure])
    # reader-form flags on forcing that is not of the form: falls back, results as ever
    c = refbind.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, g.params, g.states, fast="dropin", reader_form=True, **g.opts)
    assert "forcing resident as fp32 rows" in capfd.readouterr().err
    for i in range(g.mesh.ng):
        assert gu.rel_l2(c["qsim"][i], g.fwd["qsim"][i]) <= gu.tol(g.noise["qsim"][i])
