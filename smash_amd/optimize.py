"""Variational calibration loop on the host: the mirror of mw_optimize::optimize_lbfgsb
(smash/solver/optimize/mw_optimize.f90:484-676) over GPU sweeps.

Same steps as the reference: normalise the control (mwd_parameters_manipulation.f90:154-178), background = normalised
start, control vector = optimised fields x active cells in column-major order (var_to_control_lbfgsb :679-718, fp32 ->
fp64), L-BFGS-B with m = 10, factr = 10, pgtol = 1e-12 and bounds [0, 1] (:505-512, 541-543), every function/gradient
evaluation one forward_b with denormalize_forward on (:590-606), the two extra stop tests (:625-633), a final forward
(:646-647).  The L-BFGS-B driver is the library's own (smashx_lbfgsb_*, written from the papers the reference's lbfgsb.f
implements; round 3) -- scipy's build of that very code is one environment variable away (SMASHX_LBFGSB=scipy) and, fed by
the CPU oracle, reproduces the reference's cost trajectory bit for bit (tests/test_oracle_golden.py); the native driver
follows it to rounding of the inner products.
"""
from __future__ import annotations

import numpy as np

from .solver import _solver_for, forward
from .synth import PARAM_NAMES, STATE_NAMES


def _normalize(obj, names, lb, ub):
    for i, k in enumerate(names):
        a = getattr(obj, k)
        setattr(obj, k, np.asfortranarray(((a - np.float32(lb[i])) / (np.float32(ub[i]) - np.float32(lb[i]))).astype(np.float32)))


class _Held:
    """a trial point a worker rank has already received (Decomposition.bcast_point hands it back unchanged)"""

    def __init__(self, x):
        self.x = x


class _Stop(Exception):
    pass


def _lbfgsb_native(fg, x0, m, factr, pgtol, maxiter, maxfun, callback):
    """The library's own L-BFGS-B (include/smashx.h smashx_lbfgsb_*: written from Byrd-Lu-Nocedal-Zhu 1995 / Morales-Nocedal 2011 /
    More'-Thuente, threaded host C++) on the box [0, 1]^n.  No scipy in the loop.  Same method, parameters and stopping tests as the
    lbfgsb.f the reference calls; on test problems its iterates agree with scipy's build of that code to 1e-15 per iteration
    (tests/test_cabi_cpu.py).  Returns (x, f, info) like _lbfgsb_scipy."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    n = x0.size
    lo, up = np.zeros(n, np.float64), np.ones(n, np.float64)
    h = C.c_void_p()
    _lib.check(L.smashx_lbfgsb_create(n, int(m), lo.ctypes.data, up.ctypes.data, float(factr), float(pgtol), C.byref(h)))
    try:
        x = np.array(x0, dtype=np.float64)
        task, f, g = C.c_int(0), 0.0, np.zeros(n, np.float64)
        nit = nfev = 0
        stop = None
        while True:
            _lib.check(L.smashx_lbfgsb_step(h, x.ctypes.data, float(f), g.ctypes.data, C.byref(task)))
            if task.value == 1:                              # f and g wanted at x
                # (x itself, not a copy: fg reads it before it returns and the optimiser only moves it in its next step -- 134 MB
                # per evaluation at 1.7e7 variables)
                f, g = fg(x)
                g = np.ascontiguousarray(g, np.float64)
                nfev += 1
            elif task.value == 2:                            # new iterate
                nit += 1
                if callback is not None:
                    if getattr(callback, "wants_pg", False):
                        # the optimiser's own |projected gradient| at this iterate (lbfgsb.f dsave(13)) saves the callback five passes
                        # over x; such a callback copies x itself if it keeps it
                        callback(x, float(L.smashx_lbfgsb_projected_gradient(h)))
                    else:
                        callback(np.copy(x))
                if nit >= maxiter:
                    stop = "STOP: TOTAL NO. of ITERATIONS REACHED LIMIT"
                    break
                if nfev > maxfun:
                    stop = "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
                    break
            else:
                break
        msg = stop or L.smashx_lbfgsb_message(h).decode()
        return x, float(f), {"task": msg, "nit": nit, "funcalls": nfev, "grad": g}
    finally:
        L.smashx_lbfgsb_destroy(h)


def _lbfgsb_scipy(fg, x0, m, factr, pgtol, maxiter, maxfun, callback):
    """scipy's build of the Zhu-Byrd-Lu-Nocedal 3.0 code the reference carries as lbfgsb.f, for comparison (SMASHX_LBFGSB=scipy):
    bit-identical iterates to the reference's.  scipy's public drivers walk the bounds in Python loops over n before they start --
    tens of seconds for the 1.7e7 control variables of a 2048 x 2048 grid -- so where its `setulb` entry point has the signature of
    scipy 1.15 the driver loop (scipy/optimize/_lbfgsb_py.py::_minimize_lbfgsb) is restated around it with the bound arrays built in
    one go; any other scipy goes through the public fmin_l_bfgs_b.  Returns (x, f, info)."""
    import scipy
    from scipy.optimize import fmin_l_bfgs_b
    n = x0.size
    try:
        from scipy.optimize import _lbfgsb
        direct = tuple(int(v) for v in scipy.__version__.split(".")[:2]) == (1, 15) and hasattr(_lbfgsb, "setulb")
    except Exception:  # pragma: no cover
        direct = False
    if not direct:  # pragma: no cover
        return fmin_l_bfgs_b(fg, x0, m=m, factr=factr, pgtol=pgtol, bounds=[(0.0, 1.0)] * n, maxiter=maxiter, maxfun=maxfun,
                             callback=callback)
    x = np.clip(np.array(x0, dtype=np.float64), 0.0, 1.0)
    nbd = np.full(n, 2, np.int32)                       # both bounds present
    low, up = np.zeros(n, np.float64), np.ones(n, np.float64)
    f = np.array(0.0, dtype=np.int32)                   # (scipy's own driver starts from this placeholder; setulb never reads it before task 3)
    g = np.zeros(n, np.float64)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, np.int32)
    task, ln_task, lsave = np.zeros(2, np.int32), np.zeros(2, np.int32), np.zeros(4, np.int32)
    isave, dsave = np.zeros(44, np.int32), np.zeros(29, np.float64)
    nit = nfev = 0
    while True:
        g = np.asarray(g, np.float64)
        _lbfgsb.setulb(m, x, low, up, nbd, f, g, factr, pgtol, wa, iwa, task, lsave, isave, dsave, 20, ln_task)
        if task[0] == 3:                                # f and g wanted at x
            f, g = fg(np.copy(x))
            nfev += 1
        elif task[0] == 1:                              # new iterate
            nit += 1
            if callback is not None:
                callback(np.copy(x))
            if nit >= maxiter:
                task[0], task[1] = 5, 504
            elif nfev > maxfun:
                task[0], task[1] = 5, 502
        else:
            break
    msg = {4: "CONVERGENCE", 5: "STOP", 6: "WARNING", 7: "ERROR", 8: "ABNORMAL"}.get(int(task[0]), str(int(task[0])))
    return x, float(f), {"task": f"{msg} ({int(task[1])})", "nit": nit, "funcalls": nfev, "grad": g}


def _lbfgsb_box(fg, x0, m, factr, pgtol, maxiter, maxfun, callback):
    """L-BFGS-B on the box [0, 1]^n: the library's own implementation (default), or scipy's build of the reference's lbfgsb.f with
    SMASHX_LBFGSB=scipy."""
    import os
    if os.environ.get("SMASHX_LBFGSB", "native") == "scipy":
        return _lbfgsb_scipy(fg, x0, m, factr, pgtol, maxiter, maxfun, callback)
    return _lbfgsb_native(fg, x0, m, factr, pgtol, maxiter, maxfun, callback)


def wjreg_range(wjreg_opt, nb_wjreg_lcurve):
    """Regularisation weights tried by the L-curve, centred on the `fast` estimate (core/simulation/_optimize.py:911-945,
    `_compute_wjreg_range`): five weights a third of a decade apart around it, and for nb_wjreg_lcurve > 6 whole decades
    below and above (the extra cycles split half / half, the odd one above).  float32 like the reference's."""
    centre = np.log10(wjreg_opt)
    span = 0.66
    around = np.array(10 ** np.arange(centre - span, centre + span + 0.01, span / 2), dtype=np.float32)
    extra = int(nb_wjreg_lcurve) - 6
    if extra <= 0:
        return around
    below_from = centre - span - (extra - np.ceil(extra / 2.0))
    above_to = centre + span + 1.0 + (extra - np.floor(extra / 2.0))
    below = np.array(10 ** np.arange(below_from, centre - span), dtype=np.float32)
    above = np.array(10 ** np.arange(centre + span + 1.0, above_to), dtype=np.float32)
    return np.hstack((below, around, above))


def best_lcurve_weight(cost_jobs, cost_jreg, wjreg, jobs_min, jobs_max, jreg_min, jreg_max):
    """The weight at the corner of the L-curve (core/simulation/_optimize.py:948-1003, `_compute_best_lcurve_weight`).  In the
    plane x = share of the attainable misfit reduction a cycle kept, y = share of the largest regularisation term it paid, the
    corner is the point farthest below the diagonal y = x, measured the reference's way: hyp * sin(pi/4 - acos(x / hyp)) with
    hyp = sqrt(x^2 + y^2).  Operand types are the caller's (the reference's cycles hand float32 arrays and float32 extrema), so
    the `y < x` test next to the diagonal and the `>=` tie-break (of equal distances the LAST wins) fall as they do there.
    Points on or above the diagonal get NaN, cycles that did not reduce the misfit 0.  Returns (distance float32 array, weight
    or None)."""
    cost_jobs, cost_jreg = np.asarray(cost_jobs), np.asarray(cost_jreg)
    if not (cost_jobs.size > 2 and (jreg_max - jreg_min) > 0.0 and (jobs_max - jobs_min) > 0.0):
        return np.empty(0, np.float32), None
    x = (jobs_max - cost_jobs) / (jobs_max - jobs_min)         # element type of the inputs
    y = (cost_jreg - jreg_min) / (jreg_max - jreg_min)
    dist = np.zeros(cost_jobs.size, np.float32)
    best, far = None, 0.0
    for i in range(cost_jobs.size):
        if not (y[i] < x[i]):
            dist[i] = np.nan
            continue
        if cost_jobs[i] < jobs_max:
            hyp = (x[i] ** 2.0 + y[i] ** 2.0) ** 0.5
            dist[i] = hyp * np.sin(np.pi * 0.25 - np.arccos(x[i] / hyp))
        if dist[i] >= far:
            far, best = dist[i], wjreg[i]
    return dist, best


def auto_wjreg_cycles(run_cycle, restore, auto_wjreg, nb_wjreg_lcurve=6, verbose=False):
    """The calibration cycles behind auto_wjreg = 'fast' | 'lcurve' (core/simulation/_optimize.py:257-453, the part of
    `_optimize_lbfgsb` that chooses the weight of the regularisation term).
    run_cycle(wjreg) -> dict(cost, cost_jobs, cost_jreg, cost_jobs_initial): one complete optimize_lbfgsb with that weight;
    restore(): parameters / states back to the first guess.  Returns (wjreg chosen or None, lcurve dict or None): the last
    run_cycle call made here is the final calibration with the chosen weight (no weight chosen: the caller runs the model
    as it is, like the reference)."""
    def say(tag, w):
        if verbose:
            print(f"    {tag}: wJreg = {w:.6f}\n")

    say("CYCLE 1", 0.0)
    first = run_cycle(0.0)
    if auto_wjreg == "fast":
        # the misfit the unregularised calibration removed, per unit of regularisation term it ended with
        w = (first["cost_jobs_initial"] - first["cost_jobs"]) / first["cost_jreg"]
        restore()
        say("FINAL CYCLE", w)
        run_cycle(w)
        return w, None
    if auto_wjreg != "lcurve":
        raise ValueError(f"Unknown auto_wjreg '{auto_wjreg}'. Choices: ['fast', 'lcurve']")
    jobs_min, jobs_max, jreg_max = first["cost_jobs"], first["cost_jobs_initial"], first["cost_jreg"]
    if (jobs_min / jobs_max) < 0.95 and jreg_max > 0.0:
        w_fast = (jobs_max - jobs_min) / jreg_max
        tries = wjreg_range(w_fast, nb_wjreg_lcurve)
    else:
        w_fast, tries = 0.0, np.empty(0, np.float32)
    rec = {k: np.zeros(tries.size + 1, np.float32) for k in ("cost", "cost_jobs", "cost_jreg", "wjreg")}
    for k in ("cost", "cost_jobs", "cost_jreg"):
        rec[k][0] = first[k]
    for i, w in enumerate(tries):
        restore()
        say(f"CYCLE {i + 2}", w)
        r = run_cycle(w)
        for k in ("cost", "cost_jobs", "cost_jreg"):
            rec[k][i + 1] = r[k]
        rec["wjreg"][i + 1] = w
    jobs_min, jobs_max = np.min(rec["cost_jobs"]), np.max(rec["cost_jobs"])
    jreg_min, jreg_max = np.min(rec["cost_jreg"]), np.max(rec["cost_jreg"])
    dist, w_best = best_lcurve_weight(rec["cost_jobs"], rec["cost_jreg"], rec["wjreg"], jobs_min, jobs_max, jreg_min, jreg_max)
    lcurve = {"cost_jobs_initial": jobs_max, "cost_jreg_initial": jreg_min, "wjreg_lcurve_opt": w_best, "wjreg_fast": w_fast,
              "wjreg": rec["wjreg"], "distance": dist, "cost": rec["cost"], "cost_jobs": rec["cost_jobs"], "cost_jreg": rec["cost_jreg"]}
    restore()
    if w_best is not None:
        say("FINAL CYCLE", w_best)
        run_cycle(w_best)
    return w_best, lcurve


def optimize_lbfgsb(setup, mesh, input_data, parameters, states, output, verbose=False, auto_wjreg=None, nb_wjreg_lcurve=6,
                    return_lcurve=False, decomposition=None):
    """In-place like the reference: parameters / states come back calibrated (denormalised), output holds the final run.
    Returns a dict with the cost trajectory.  setup.optimize.maxiter bounds the iterations (default 100).
    auto_wjreg = 'fast' | 'lcurve' (only with jreg_fun set): the weight of the regularisation term is found by calibration
    cycles first (auto_wjreg_cycles above) and left in setup.optimize.wjreg; the returned dict is the final cycle's, with
    'wjreg' and, for return_lcurve, 'lcurve' added.

    decomposition (multi-GPU, smash_amd.tiles.Decomposition): the calibration over the ranks of a tile decomposition.  Every rank
    calls this function with ITS setup / mesh (its gauges), whole-grid parameter planes and input_data._smashx_solver = the
    plan of its part (exchange set).  The control vector is the whole grid's, in the reference's order; rank 0 runs L-BFGS-B and
    hands every trial point to the others, each evaluation is one collective forward_b: the parts' cost_jobs are summed, the
    regulariser's term -- evaluated over the whole grid by every rank -- counted once, every rank contributes the gradient of the
    cells it owns.  The iterates are those of the single domain (the sweep is bit-identical, the sums are not reordered).
    auto_wjreg works over a decomposition too: every cycle is a collective calibration and every rank takes the same decisions."""
    if auto_wjreg is not None and setup.optimize.njr > 0:
        o = setup.optimize
        if auto_wjreg == "lcurve" and nb_wjreg_lcurve < 6:
            raise ValueError("nb_wjreg_lcurve option must be greater or equal to 6")
        par0, sta0 = parameters.copy(), states.copy()
        last = {}

        def restore():
            for k in PARAM_NAMES:
                setattr(parameters, k, np.asfortranarray(getattr(par0, k)).copy(order="F"))
            for k in STATE_NAMES:
                setattr(states, k, np.asfortranarray(getattr(sta0, k)).copy(order="F"))

        def run_cycle(w):
            o.wjreg = float(w)
            # over a decomposition every cycle is a collective calibration; what the cycles decide on -- the decomposition's cost_jobs
            # (all-reduced), the common cost_jreg, the initial cost_jobs -- is the same numbers on every rank, so every rank tries the
            # same weights and picks the same one
            h = optimize_lbfgsb(setup, mesh, input_data, parameters, states, output, verbose=verbose, decomposition=decomposition)
            last["h"] = h
            return dict(cost=output.cost, cost_jobs=output.cost_jobs, cost_jreg=output.cost_jreg, cost_jobs_initial=h["cost_jobs_initial"])

        w, lcurve = auto_wjreg_cycles(run_cycle, restore, auto_wjreg, nb_wjreg_lcurve, verbose)
        if w is None:
            # no corner found: the model is run as it is; like the reference, wjreg stays at the last weight a cycle tried
            # (core/simulation/_optimize.py:434-450 does not reset it), so output.cost carries that weight's regularisation term
            forward(setup, mesh, input_data, parameters, parameters.copy(), states, states.copy(), output, np.float32(0))
            if decomposition is not None:
                v = np.array([float(output.cost_jobs)], np.float64)
                decomposition.allreduce(v)
                output.cost_jobs = np.float32(v[0])
                output.cost = np.float32(np.float32(v[0]) + np.float32(o.wjreg) * np.float32(output.cost_jreg))
        h = dict(last["h"], wjreg=w)
        if return_lcurve and lcurve is not None:
            h["lcurve"] = lcurve
        return h
    o = setup.optimize
    maxiter = int(getattr(o, "maxiter", 100))
    act = np.asarray(mesh.active_cell) == 1
    mask_f = act.reshape(-1, order="F")                    # column outer, row inner = the reference's control-vector order
    pf = [k for i, k in enumerate(PARAM_NAMES) if o.optim_parameters[i] > 0]
    sf = [k for i, k in enumerate(STATE_NAMES) if o.optim_states[i] > 0]
    m = int(np.count_nonzero(mask_f))
    n = m * (len(pf) + len(sf))
    if n == 0:
        raise ValueError("nothing to optimise: optim_parameters / optim_states are all zero")

    _normalize(parameters, PARAM_NAMES, o.lb_parameters, o.ub_parameters)
    _normalize(states, STATE_NAMES, o.lb_states, o.ub_states)
    par_bgd, sta_bgd = parameters.copy(), states.copy()
    par_b, sta_b = parameters.copy(), states.copy()
    was = o.denormalize_forward
    o.denormalize_forward = True

    def flat(obj, k):
        a = getattr(obj, k)
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.f_contiguous):
            a = np.asfortranarray(a, dtype=np.float32)
            setattr(obj, k, a)
        return a.reshape(-1, order="F")                   # a view of the Fortran-ordered field

    def to_control(p, s):
        return np.concatenate([flat(p, k)[mask_f].astype(np.float64) for k in pf] + [flat(s, k)[mask_f].astype(np.float64) for k in sf])

    def to_var(x):
        for j, k in enumerate(pf):
            flat(parameters, k)[mask_f] = x[j * m:(j + 1) * m]
        for j, k in enumerate(sf):
            flat(states, k)[mask_f] = x[(len(pf) + j) * m:(len(pf) + j + 1) * m]

    def renormalize():
        _normalize(parameters, PARAM_NAMES, o.lb_parameters, o.ub_parameters)
        _normalize(states, STATE_NAMES, o.lb_states, o.ub_states)

    hist = {"cost": [], "nfg": 0}
    x = to_control(parameters, states)
    moved = set(pf + sf)
    dec = decomposition
    if dec is not None:
        own_f = np.asarray(dec.owned, bool).reshape(-1, order="F")[mask_f]          # control-vector order
        own_ctl = np.tile(own_f, len(pf) + len(sf))

    def whole_cost():
        """the decomposition's cost from this rank's output: sum of the parts' cost_jobs + wjreg x the (common) cost_jreg"""
        if dec is None:
            return float(np.float32(output.cost))
        v = np.array([float(output.cost_jobs)], np.float64)
        dec.allreduce(v)
        output.cost_jobs = np.float32(v[0])
        output.cost = np.float32(np.float32(v[0]) + np.float32(o.wjreg) * np.float32(output.cost_jreg))
        return float(output.cost)

    try:
        forward(setup, mesh, input_data, parameters, par_bgd, states, sta_bgd, output, np.float32(0))
        renormalize()
        whole_cost()
        hist["cost_jobs_initial"], hist["cost_jreg_initial"] = output.cost_jobs, output.cost_jreg
        last = {}
        # every evaluation is one forward_b with denormalize_forward on (mw_optimize.f90:590-606).  The plan keeps the fields
        # that do not move on the device: an evaluation uploads the optimised fields only and brings back cost, discharge
        # and the gradient of those fields -- nothing is denormalised and renormalised on the host in between.
        sol = _solver_for(setup, mesh, input_data)
        sol.upload(parameters, states, par_bgd, sta_bgd)

        device_pack = dec is None and sol.control_size() == n       # control_to_var / var_to_control on the device (smashx_control_*)

        def fg(xc):
            if dec is not None:
                # one collective evaluation: rank 0's trial point on every rank, the parts' costs and gradients put together
                xc = dec.bcast_point(xc)
                to_var(xc)
                sol.upload(parameters, states, par_bgd, sta_bgd, only=moved)
                sol.sweep(True, 1.0)
                sol.download(True, None, None, output, par_b, sta_b, only_b=moved)
                g = to_control(par_b, sta_b)
                g[~own_ctl] = 0.0                                   # (cells of other parts carry the regulariser's gradient here too)
                dec.allreduce(g)
                hist["nfg"] += 1
                last["f"], last["g"] = whole_cost(), g
                return last["f"], g
            if device_pack:
                # mw_optimize.f90:590-606 with the packing on the device: one contiguous fp64 vector goes up, the fields are
                # unpacked, cast and denormalised there; cost + discharge and one contiguous gradient vector come back
                sol.control_set(xc)
                sol.sweep(True, 1.0)
                sol.cost_and_qsim(output)
                g = sol.control_gradient()
            else:
                to_var(xc)
                sol.upload(parameters, states, par_bgd, sta_bgd, only=moved)
                sol.sweep(True, 1.0)
                sol.download(True, None, None, output, par_b, sta_b, only_b=moved)
                g = to_control(par_b, sta_b)
            hist["nfg"] += 1
            last["f"], last["g"] = float(np.float32(output.cost)), g
            return last["f"], g

        def cb(xk, pg=None):
            hist["cost"].append(last["f"])
            if verbose:
                print(f"    At iterate {len(hist['cost']):3d}    nfg = {hist['nfg']:5d}    J = {last['f']:14.6f}")
            if pg is None:
                pg = np.max(np.abs(xk - np.clip(xk - last["g"], 0.0, 1.0)))   # |proj g| (lbfgsb.f projgr)
            if pg <= 1e-10 * (1.0 + abs(last["f"])):
                last["x"] = xk.copy()
                raise _Stop

        cb.wants_pg = True
        if dec is None or dec.rank == 0:
            try:
                x, f, info = _lbfgsb_box(fg, x, 10, 10.0, 1e-12, maxiter, 10 * maxiter + 20, cb)
                hist["task"] = str(info.get("task", ""))
            except _Stop:
                x = last["x"]
                hist["task"] = "STOP: THE PROJECTED GRADIENT IS SUFFICIENTLY SMALL"
            if dec is not None:
                x = dec.bcast_point(x, done=True)
        else:
            # the other ranks evaluate what rank 0 asks for until it says the search is over
            while True:
                xw = dec.bcast_point(None)
                if xw is None:
                    break
                fg(_Held(xw))
            x = dec.final_point
        to_var(x)
        forward(setup, mesh, input_data, parameters, par_bgd, states, sta_bgd, output, np.float32(0))
        hist["final_cost"] = whole_cost()
    finally:
        o.denormalize_forward = was
    return hist
