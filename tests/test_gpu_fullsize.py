"""GPU, BASELINE.json's full single-GPU size (configs[2]: 1024 x 1024 cells x 8760 hourly steps, gr-b, forward +
adjoint): the oracle cannot run this, so the sweep is checked through size-independent properties of the path:

  * determinism        two sweeps give bit-identical cost and gradient fields
  * linearity          the adjoint is linear in its seed: cost_b = 2 doubles every gradient field exactly
                       (a power-of-two factor commutes with every rounding)
  * directional        (J(theta + e d) - J(theta - e d)) / 2e  against  <grad J, d>  for d = sign(grad) * theta over
    derivative         cp, cft and lr (two extra forward sweeps; fp32 finite differences: 5 % bar)
  * storage invariance the chunked checkpoint/recompute adjoint (4 storage chunks) is bit-identical to store-all

BASELINE.json configs[3] (full VDA L-BFGS-B loop, distributed mapping, 2048 x 2048 grid, 1 GPU) at its full length: a year of
hourly fp32 forcing on 2048^2 cells is 294 GB, more than the card; in the lossless compact layout (uint16 rain counts + daily
PET, smashx_set_forcing_layout) it is 80 GB and the adjoint runs checkpointed in storage chunks.  smash_amd.optimize_lbfgsb
(host mirror of mw_optimize::optimize_lbfgsb) drives GPU sweeps over 16.8 M control variables; checked: the cost decreases
over the iterations and the final forward run reproduces the last evaluated cost.

Chained routing rounds (sx_kernels.h): bit-identity with one launch per round at the 2048 x 1024 tile of configs[4], and the
stall path -- a chained group that never publishes makes its consumers give up, the sweep is repeated unchained, same bits.

SMASHX_FULLSIZE_GRID / SMASHX_FULLSIZE_NT (and SMASHX_VDA_GRID / SMASHX_VDA_NT) shrink the cases for a quick run.
"""
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = int(os.environ.get("SMASHX_FULLSIZE_GRID", "1024"))
NT = int(os.environ.get("SMASHX_FULLSIZE_NT", "8760"))
FIELDS_P = ("ci", "cp", "cft", "exc", "lr")
FIELDS_S = ("hi", "hp", "hft", "hlr")


def _problem(chunk, N=N, NT=NT, ncol=None, compact=False):
    import torch
    import smash_amd
    from smash_amd import synth
    from smash_amd.solver import Solver
    dev = torch.device("cuda", 0)
    M = ncol or N
    m = synth.make_mesh(N, M, ng=8)
    setup = smash_amd.SetupDT(0, m.ng, structure="gr-b", dt=3600.0, ntime_step=NT)
    setup.optimize.jobs_fun, setup.optimize.wjobs_fun = ["nse"], [1.0]
    mesh = smash_amd.MeshDT.from_synth(setup, m)
    sol = Solver(setup, mesh, chunk_steps=chunk)
    if compact:                                      # uint16 rain counts + daily PET: a year of 2048^2 forcing is 80 GB instead of 294
        sol.set_forcing_layout(compact=True, prcp_factor=0.1, pet_ratio=synth._pet_tables()[1], pet_hour0=0)
    rows, cols = sol.cell_order()
    d_rows = torch.from_numpy(rows.astype(np.int64)).to(dev)
    d_cols = torch.from_numpy(cols.astype(np.int64)).to(dev)
    tb = max(24, (1 << 26) // max(sol.ncells, 1) // 24 * 24)
    for t0 in range(0, NT, tb):
        t1 = min(NT, t0 + tb)
        prcp, pet = synth.forcing_block(d_rows, d_cols, t0, t1, xp=torch, device=dev)
        torch.cuda.synchronize()
        sol.set_forcing_device_block(t0, t1, prcp.data_ptr(), pet.data_ptr())
        del prcp, pet
    del d_rows, d_cols
    torch.cuda.empty_cache()
    par = smash_amd.ParametersDT.from_dict(mesh, synth.make_parameters(N, M))
    sta = smash_amd.StatesDT.from_dict(mesh, synth.make_states(N, M, warm=True))
    parq = smash_amd.ParametersDT.from_dict(mesh, synth.make_parameters(N, M, perturb=0.1))
    out = smash_amd.OutputDT(setup, mesh)
    sol.set_options(setup.optimize)
    sol.upload(parq, sta)
    sol.sweep(False)
    sol.download(False, parq, sta, out)
    sol.set_qobs(out.qsim)                       # observations = run with parameters + 10 %
    return sol, setup, mesh, par, sta


def _adjoint(sol, setup, mesh, par, sta, cost_b=1.0):
    import smash_amd
    out = smash_amd.OutputDT(setup, mesh)
    par_b, sta_b = par.copy(), sta.copy()
    sol.upload(par, sta)
    sol.sweep(True, cost_b)
    cost = sol.download(True, par, sta, out, par_b, sta_b)
    return cost, {k: getattr(par_b, k).copy() for k in FIELDS_P}, {k: getattr(sta_b, k).copy() for k in FIELDS_S}, out.qsim.copy()


def _cost(sol, setup, mesh, par, sta):
    import smash_amd
    out = smash_amd.OutputDT(setup, mesh)
    sol.upload(par, sta)
    sol.sweep(False)
    return sol.download(False, par, sta, out)


def test_fullsize_properties():
    sol, setup, mesh, par, sta = _problem(0)
    assert sol.chunking()[0] >= NT, "store-all expected at this size (288 GB HBM)"
    c1, p1, s1, q1 = _adjoint(sol, setup, mesh, par, sta)
    assert np.isfinite(c1) and 0.0 < c1 < 10.0
    for k in FIELDS_P:
        assert np.all(np.isfinite(p1[k])) and np.any(p1[k] != 0.0), k
    # determinism
    c1b, p1b, s1b, q1b = _adjoint(sol, setup, mesh, par, sta)
    assert c1b == c1 and np.array_equal(q1, q1b)
    assert all(np.array_equal(p1[k], p1b[k]) for k in FIELDS_P) and all(np.array_equal(s1[k], s1b[k]) for k in FIELDS_S)
    # linearity in the seed
    c2, p2, s2, _ = _adjoint(sol, setup, mesh, par, sta, cost_b=2.0)
    assert c2 == c1
    for k in FIELDS_P:
        assert np.array_equal(p2[k], np.float32(2.0) * p1[k]), k
    for k in FIELDS_S:
        assert np.array_equal(s2[k], np.float32(2.0) * s1[k]), k
    # directional derivative along d = sign(grad) * theta over cp, cft and lr: <grad, d> = sum |grad * theta|
    eps = 1e-2
    keep = {k: getattr(par, k).copy() for k in ("cp", "cft", "lr")}
    sgn = {k: np.sign(p1[k]).astype(np.float32) for k in keep}
    for k in keep:
        setattr(par, k, np.asfortranarray(keep[k] * (np.float32(1.0) + np.float32(eps) * sgn[k])))
    jp = _cost(sol, setup, mesh, par, sta)
    for k in keep:
        setattr(par, k, np.asfortranarray(keep[k] * (np.float32(1.0) - np.float32(eps) * sgn[k])))
    jm = _cost(sol, setup, mesh, par, sta)
    for k in keep:
        setattr(par, k, keep[k])
    fd = (jp - jm) / (2.0 * eps)
    ad = float(sum(np.sum(np.abs(p1[k].astype(np.float64) * keep[k].astype(np.float64))) for k in keep))
    assert abs(fd - ad) <= 5e-2 * abs(ad) + 1e-6, (fd, ad)
    # storage invariance: 4 storage chunks, checkpoints + recomputation
    del sol
    gc.collect()
    sol, setup, mesh, par, sta = _problem(((NT + 3) // 4 + 15) // 16 * 16)
    assert sol.chunking()[0] < NT
    c3, p3, s3, q3 = _adjoint(sol, setup, mesh, par, sta)
    assert c3 == c1 and np.array_equal(q3, q1)
    for k in FIELDS_P:
        assert np.array_equal(p3[k], p1[k]), k
    for k in FIELDS_S:
        assert np.array_equal(s3[k], s1[k]), k
    del sol
    gc.collect()


def test_vda_lbfgsb_loop_2048():
    import types
    import smash_amd
    n2 = int(os.environ.get("SMASHX_VDA_GRID", "2048"))
    nt2 = int(os.environ.get("SMASHX_VDA_NT", "8760"))
    sol, setup, mesh, par, sta = _problem(0, n2, nt2, compact=True)
    assert sol.forcing_info()["layout"].startswith("compact")
    out = smash_amd.OutputDT(setup, mesh)
    sol.upload(par, sta)
    sol.sweep(False)
    j_start = sol.download(False, par, sta, out)
    qobs = np.asfortranarray(sol_qobs(sol, setup, mesh, par, sta))
    inp = types.SimpleNamespace(qobs=qobs, _smashx_solver=sol)         # forcing is already resident in the plan
    op = np.zeros(16, np.int32)
    op[[1, 3, 6, 15]] = 1                                               # cp, cft, exc, lr: 4 x 4.2 M control variables
    setup.optimize.optim_parameters = op
    setup.optimize.maxiter = 2
    h = smash_amd.optimize_lbfgsb(setup, mesh, inp, par, sta, out)
    assert len(h["cost"]) == 2 and h["nfg"] >= 2
    assert h["cost"][0] < j_start and h["cost"][1] < h["cost"][0], (j_start, h)
    assert abs(h["final_cost"] - h["cost"][-1]) <= 1e-5 * abs(h["cost"][-1]) + 1e-7, h
    lb, ub = setup.optimize.lb_parameters, setup.optimize.ub_parameters
    for i, k in ((1, "cp"), (3, "cft"), (6, "exc"), (15, "lr")):        # calibrated fields come back denormalised, inside the bounds
        a = getattr(par, k)
        assert np.all(a >= lb[i] - 1e-3 * abs(ub[i] - lb[i])) and np.all(a <= ub[i] + 1e-3 * abs(ub[i] - lb[i])), k
    del sol
    gc.collect()


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_chained_rounds_equal_launch_per_round_at_the_2048x1024_tile():
    """Twice the groups of the 1024^2 case in the chained launch (more than the chip holds at once): the progress-counter
    protocol must neither stall nor change a bit."""
    nt = int(os.environ.get("SMASHX_CHAIN_NT", "192"))
    res = {}
    for chain in ("1", "0"):
        def run():
            sol, setup, mesh, par, sta = _problem(0, 2048, nt, ncol=1024)
            r = _adjoint(sol, setup, mesh, par, sta)
            tm = sol.timing()
            del sol
            gc.collect()
            return r, tm
        res[chain] = _with_env({"SMASHX_CHAIN_ROUNDS": chain}, run)
    (c1, p1, s1, q1), t1 = res["1"]
    (c0, p0, s0, q0), t0 = res["0"]
    assert t1["route_fwd_launches"] == 2 and t0["route_fwd_launches"] == t0["n_rounds"] > 2
    assert c1 == c0 and np.array_equal(q1, q0)
    assert all(np.array_equal(p1[k], p0[k]) for k in FIELDS_P) and all(np.array_equal(s1[k], s0[k]) for k in FIELDS_S)


def test_stalled_chain_falls_back_to_one_launch_per_round(capfd):
    """SMASHX_DEBUG_MUTE_GROUP=-2: the first chained group never publishes its progress; its consumers hit the poll limit
    (SMASHX_SPIN_LIMIT) and raise the stall flag; smashx_sweep drops the plan to one launch per round, repeats the sweep and
    returns the same bits as a healthy plan."""
    ref, tref = _with_env({}, lambda: (lambda P: (_adjoint(*P), P[0].timing()))(_problem(0, 256, 96)))
    def run():
        P = _problem(0, 256, 96)                     # the forward run of _problem already stalls and falls back
        r = _adjoint(*P)
        return r, P[0].timing()
    got, tgot = _with_env({"SMASHX_DEBUG_MUTE_GROUP": "-2", "SMASHX_SPIN_LIMIT": "2000"}, run)
    assert "stalled" in capfd.readouterr().err
    assert tref["route_fwd_launches"] == 2 and tgot["route_fwd_launches"] == tgot["n_rounds"] > 2
    assert got[0] == ref[0] and np.array_equal(got[3], ref[3])
    assert all(np.array_equal(got[1][k], ref[1][k]) for k in FIELDS_P) and all(np.array_equal(got[2][k], ref[2][k]) for k in FIELDS_S)


def sol_qobs(sol, setup, mesh, par, sta):
    """The observations _problem() installed (forward run at parameters + 10 %), read back for the host loop."""
    import smash_amd
    from smash_amd import synth
    n = mesh.nrow
    parq = smash_amd.ParametersDT.from_dict(mesh, synth.make_parameters(n, n, perturb=0.1))
    out = smash_amd.OutputDT(setup, mesh)
    sol.upload(parq, sta)
    sol.sweep(False)
    sol.download(False, parq, sta, out)
    return out.qsim.copy()
