"""Variational calibration loop on the host: the mirror of mw_optimize::optimize_lbfgsb
(smash/solver/optimize/mw_optimize.f90:484-676) over GPU sweeps.

Same steps as the reference: normalise the control (mwd_parameters_manipulation.f90:154-178), background = normalised
start, control vector = optimised fields x active cells in column-major order (var_to_control_lbfgsb :679-718, fp32 ->
fp64), L-BFGS-B with m = 10, factr = 10, pgtol = 1e-12 and bounds [0, 1] (:505-512, 541-543), every function/gradient
evaluation one forward_b with denormalize_forward on (:590-606), the two extra stop tests (:625-633), a final forward
(:646-647).  The L-BFGS-B driver is scipy's (the same Zhu-Byrd-Lu-Nocedal 3.0 code the reference carries as lbfgsb.f);
with the CPU oracle as the gradient provider the cost trajectory is bit-identical to the reference's
(tests/test_oracle_golden.py), so on the GPU the only difference is the sweep itself.
"""
from __future__ import annotations

import numpy as np

from .solver import forward, forward_b
from .synth import PARAM_NAMES, STATE_NAMES


def _normalize(obj, names, lb, ub):
    for i, k in enumerate(names):
        a = getattr(obj, k)
        setattr(obj, k, np.asfortranarray(((a - np.float32(lb[i])) / (np.float32(ub[i]) - np.float32(lb[i]))).astype(np.float32)))


class _Stop(Exception):
    pass


def optimize_lbfgsb(setup, mesh, input_data, parameters, states, output, verbose=False):
    """In-place like the reference: parameters / states come back calibrated (denormalised), output holds the final run.
    Returns a dict with the cost trajectory.  setup.optimize.maxiter bounds the iterations (default 100)."""
    from scipy.optimize import fmin_l_bfgs_b
    o = setup.optimize
    maxiter = int(getattr(o, "maxiter", 100))
    act = np.asarray(mesh.active_cell) == 1
    cr = np.argwhere(act.T)                                # column outer, row inner
    cols, rows = cr[:, 0], cr[:, 1]
    pf = [k for i, k in enumerate(PARAM_NAMES) if o.optim_parameters[i] > 0]
    sf = [k for i, k in enumerate(STATE_NAMES) if o.optim_states[i] > 0]
    m = len(rows)
    n = m * (len(pf) + len(sf))
    if n == 0:
        raise ValueError("nothing to optimise: optim_parameters / optim_states are all zero")

    _normalize(parameters, PARAM_NAMES, o.lb_parameters, o.ub_parameters)
    _normalize(states, STATE_NAMES, o.lb_states, o.ub_states)
    par_bgd, sta_bgd = parameters.copy(), states.copy()
    par_b, sta_b = parameters.copy(), states.copy()
    out_b = output.copy()
    was = o.denormalize_forward
    o.denormalize_forward = True

    def to_control(p, s):
        return np.concatenate([getattr(p, k)[rows, cols].astype(np.float64) for k in pf] +
                              [getattr(s, k)[rows, cols].astype(np.float64) for k in sf])

    def to_var(x):
        for j, k in enumerate(pf):
            a = getattr(parameters, k); a[rows, cols] = x[j * m:(j + 1) * m].astype(np.float32)
        for j, k in enumerate(sf):
            a = getattr(states, k); a[rows, cols] = x[(len(pf) + j) * m:(len(pf) + j + 1) * m].astype(np.float32)

    def renormalize():
        _normalize(parameters, PARAM_NAMES, o.lb_parameters, o.ub_parameters)
        _normalize(states, STATE_NAMES, o.lb_states, o.ub_states)

    hist = {"cost": [], "nfg": 0}
    x = to_control(parameters, states)
    try:
        forward(setup, mesh, input_data, parameters, par_bgd, states, sta_bgd, output, np.float32(0))
        renormalize()
        hist["cost_jobs_initial"], hist["cost_jreg_initial"] = output.cost_jobs, output.cost_jreg
        last = {}

        def fg(xc):
            to_var(xc)
            forward_b(setup, mesh, input_data, parameters, par_b, par_bgd, par_bgd.copy(), states, sta_b, sta_bgd, sta_bgd.copy(),
                      output, out_b, np.float32(0), np.float32(1))
            renormalize()
            hist["nfg"] += 1
            g = to_control(par_b, sta_b)
            last["f"], last["g"] = float(np.float32(output.cost)), g
            return last["f"], g

        def cb(xk):
            hist["cost"].append(last["f"])
            if verbose:
                print(f"    At iterate {len(hist['cost']):3d}    nfg = {hist['nfg']:5d}    J = {last['f']:14.6f}")
            pg = np.max(np.abs(xk - np.clip(xk - last["g"], 0.0, 1.0)))       # |proj g| (lbfgsb.f projgr)
            if pg <= 1e-10 * (1.0 + abs(last["f"])):
                last["x"] = xk.copy()
                raise _Stop

        try:
            x, f, info = fmin_l_bfgs_b(fg, x, m=10, factr=10.0, pgtol=1e-12, bounds=[(0.0, 1.0)] * n, maxiter=maxiter,
                                       maxfun=10 * maxiter + 20, callback=cb)
            hist["task"] = str(info.get("task", ""))
        except _Stop:
            x = last["x"]
            hist["task"] = "STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL"
        to_var(x)
        forward(setup, mesh, input_data, parameters, par_bgd, states, sta_bgd, output, np.float32(0))
        hist["final_cost"] = float(output.cost)
    finally:
        o.denormalize_forward = was
    return hist
