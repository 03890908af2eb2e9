"""ctypes binding to oracle/_ref/libsmash_ref.so -- the UNMODIFIED reference Fortran solver built by
oracle/ref/build_ref.sh, driven through oracle/ref/ref_capi.f90.

TEST INFRASTRUCTURE: imported only by tests/, tests/golden/make_golden.py, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  Never imported by the product package smash_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GNP, GNS = 16, 8
STRUCTURES = {"gr-a": 1, "gr-b": 2, "gr-c": 3, "gr-d": 4, "vic-a": 5}
JOBS = {"nse": 1, "kge": 2, "kge2": 3, "se": 4, "rmse": 5, "logarithmic": 6}
JREG = {"prior": 1, "smoothing": 2, "hard_smoothing": 3}
GLB_P = np.array([1e-6] * 6 + [-50.0] + [1e-6] * 9, dtype=np.float32)
GUB_P = np.array([1e2, 1e3, 1e3, 1e3, 1e4, 0.999999, 50.0, 1e1, 2e3, 2e3, 2e3, 1e4, 0.999999, 30.0,
                  0.999999, 1e3], dtype=np.float32)
GLB_S = np.array([1e-6] * 8, dtype=np.float32)
GUB_S = np.array([0.999999] * 7 + [10000.0], dtype=np.float32)


def lib_path(fast=False) -> str:
    """fast: False -> parity build (-O2 -ffp-contract=off); True -> the reference's own optimisation level;
    "dropin" -> the reference with base_forward/base_forward_b replaced by fortran/smashx_dropin.f90 (GPU)."""
    name = {False: "libsmash_ref.so", True: "libsmash_ref_fast.so", "dropin": "libsmash_dropin.so",
            "ref_lbfgsb": "libsmash_ref_lbfgsb.so", "dropin_lbfgsb": "libsmash_dropin_lbfgsb.so"}[fast]
    return os.path.join(_HERE, "_ref", name)


def available(fast=False) -> bool:
    return os.path.exists(lib_path(fast))


_libs = {}


def _lib(fast):
    if fast not in _libs:
        _libs[fast] = C.CDLL(lib_path(fast))
    return _libs[fast]


def _f(a, dtype):
    return np.asfortranarray(a, dtype=dtype)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(fields: dict, names, nrow, ncol):
    out = np.zeros((nrow, ncol, len(names)), dtype=np.float32, order="F")
    for i, n in enumerate(names):
        out[:, :, i] = fields[n]
    return out


def unpack(a, names):
    return {n: np.asfortranarray(a[:, :, i]) for i, n in enumerate(names)}


def run(structure, mesh, dt, prcp, pet, qobs, params, states, *, adjoint=False, params_bgd=None,
        states_bgd=None, sparse_storage=False, denormalize_forward=False, optimize_start_step=1,
        jobs_fun=("nse",), wjobs_fun=(1.0,), jreg_fun=(), wjreg_fun=(), wjreg=0.0, wgauge=None,
        optim_parameters=None, optim_states=None, lb_parameters=None, ub_parameters=None,
        lb_states=None, ub_states=None, cost_b=1.0, nrep=1, fast=False, optimize_maxiter=None,
        params_d=None, states_d=None, params_bgd_d=None, states_bgd_d=None,
        descriptor=None, hyper_params=None, hyper_states=None, mapping=None, hyper_params_d=None, hyper_states_d=None,
        optimize_sbs_maxiter=None, reader_form=False):
    """Call the reference forward / forward_b (mw_forward.f90:18-68) on flat arrays.

    mesh: object with nrow, ncol, dx, flwdir, flwacc, path (0-based), active_cell, gauge_pos (0-based),
    area.  params/states: dict name -> (nrow, ncol) float32.  Returns a dict of outputs.
    params_d / states_d (dicts): run the tangent model forward_d (mw_forward.f90:70-97) along that direction
    instead; the result also holds cost_d and qsim_d.
    mapping ("hyper-linear" | "hyper-polynomial") with descriptor (nrow, ncol, nd), hyper_params / hyper_states
    (dict name -> (nhyper,)): mw_forward::hyper_forward / hyper_forward_b (mw_forward.f90:99-152); the result also holds
    hyper_parameters_b / hyper_states_b; with hyper_params_d / hyper_states_d it is hyper_forward_d (mw_forward.f90:154-181)
    along that direction (result: cost_d, qsim_d)."""
    from smash_amd.synth import PARAM_NAMES, STATE_NAMES
    lib = _lib(fast)
    nrow, ncol, ng = mesh.nrow, mesh.ncol, mesh.ng
    nt = prcp.shape[2]
    jobs_fun, jreg_fun = list(jobs_fun), list(jreg_fun)
    tangent = params_d is not None or states_d is not None
    hyper = mapping is not None
    mode = 3 if tangent else (2 if optimize_maxiter is not None else 7 if optimize_sbs_maxiter is not None else int(adjoint))
    if optimize_sbs_maxiter is not None:        # mw_optimize::optimize_sbs, uniform mapping (mw_optimize.f90:53-294)
        optimize_maxiter = optimize_sbs_maxiter
    nd = mcode = 0
    if hyper:
        mode = 6 if hyper_params_d is not None else 5 if adjoint else 4
        descriptor = np.asfortranarray(descriptor, dtype=np.float32)
        nd = descriptor.shape[2]
        mcode = {"hyper-linear": 1, "hyper-polynomial": 2}[mapping]
    icfg = np.array([STRUCTURES[structure], nrow, ncol, nt, ng, int(sparse_storage),
                     int(denormalize_forward), optimize_start_step, len(jobs_fun), len(jreg_fun),
                     mode, nrep, optimize_maxiter or 0, nd, mcode, int(bool(reader_form))], dtype=np.int32)
    rcfg = np.array([dt, mesh.dx, wjreg, cost_b], dtype=np.float32)
    P = pack(params, PARAM_NAMES, nrow, ncol)
    S = pack(states, STATE_NAMES, nrow, ncol)
    Pb = pack(params_bgd, PARAM_NAMES, nrow, ncol) if params_bgd is not None else P.copy(order="F")
    Sb = pack(states_bgd, STATE_NAMES, nrow, ncol) if states_bgd is not None else S.copy(order="F")
    wg = np.full(max(ng, 1), 1.0 / max(ng, 1), np.float32) if wgauge is None else np.asarray(wgauge, np.float32)
    jc = np.array([JOBS[j] for j in jobs_fun] + [0], dtype=np.int32)
    wj = np.array(list(wjobs_fun) + [0], dtype=np.float32)
    rc = np.array([JREG[j] for j in jreg_fun] + [0], dtype=np.int32)
    wr = np.array(list(wjreg_fun) + [0], dtype=np.float32)
    op = np.zeros(GNP, np.int32) if optim_parameters is None else np.asarray(optim_parameters, np.int32)
    os_ = np.zeros(GNS, np.int32) if optim_states is None else np.asarray(optim_states, np.int32)
    lbp = GLB_P if lb_parameters is None else np.asarray(lb_parameters, np.float32)
    ubp = GUB_P if ub_parameters is None else np.asarray(ub_parameters, np.float32)
    lbs = GLB_S if lb_states is None else np.asarray(lb_states, np.float32)
    ubs = GUB_S if ub_states is None else np.asarray(ub_states, np.float32)
    qsim = np.zeros((max(ng, 1), nt), np.float32, order="F")
    costs = np.zeros(3, np.float32)
    fstates = np.zeros((nrow, ncol, GNS), np.float32, order="F")
    pout = np.zeros((nrow, ncol, GNP), np.float32, order="F")
    sout = np.zeros((nrow, ncol, GNS), np.float32, order="F")
    p_b = np.zeros((nrow, ncol, GNP), np.float32, order="F")
    s_b = np.zeros((nrow, ncol, GNS), np.float32, order="F")
    elapsed = C.c_double(0.0)
    args = [icfg, rcfg, _f(mesh.flwdir, np.int32), _f(mesh.flwacc, np.int32),
            _f(np.asarray(mesh.path) + 1, np.int32), _f(mesh.active_cell, np.int32),
            _f(np.asarray(mesh.gauge_pos) + 1, np.int32), np.ascontiguousarray(mesh.area, np.float32),
            _f(prcp, np.float32), _f(pet, np.float32), _f(qobs, np.float32), P, Pb, S, Sb, wg, jc, wj,
            rc, wr, op, os_, lbp, ubp, lbs, ubs, qsim, costs, fstates, pout, sout, p_b, s_b]
    cargs = [_ptr(a) for a in args] + [C.byref(elapsed)]
    if tangent:
        zp, zs = np.zeros((nrow, ncol, GNP), np.float32, order="F"), np.zeros((nrow, ncol, GNS), np.float32, order="F")
        P_d = pack(params_d, PARAM_NAMES, nrow, ncol) if params_d is not None else zp
        S_d = pack(states_d, STATE_NAMES, nrow, ncol) if states_d is not None else zs
        Pb_d = pack(params_bgd_d, PARAM_NAMES, nrow, ncol) if params_bgd_d is not None else zp
        Sb_d = pack(states_bgd_d, STATE_NAMES, nrow, ncol) if states_bgd_d is not None else zs
        qsim_d = np.zeros((max(ng, 1), nt), np.float32, order="F")
        cost_d = C.c_float(0.0)
        cargs += [_ptr(P_d), _ptr(S_d), _ptr(Pb_d), _ptr(Sb_d), _ptr(qsim_d), C.byref(cost_d)]
    if hyper:
        nh = 1 + mcode * nd
        HP = np.zeros((nh, 1, GNP), np.float32, order="F")
        HS = np.zeros((nh, 1, GNS), np.float32, order="F")
        for i, k in enumerate(PARAM_NAMES):
            HP[:, 0, i] = hyper_params[k]
        for i, k in enumerate(STATE_NAMES):
            HS[:, 0, i] = hyper_states[k]
        HP_b, HS_b = np.zeros_like(HP, order="F"), np.zeros_like(HS, order="F")
        if mode == 6:
            for i, k in enumerate(PARAM_NAMES):
                HP_b[:, 0, i] = hyper_params_d[k]
            for i, k in enumerate(STATE_NAMES):
                HS_b[:, 0, i] = hyper_states_d[k]
        qsim_d = np.zeros((max(ng, 1), nt), np.float32, order="F")
        cost_d = C.c_float(0.0)
        cargs += [_ptr(descriptor), _ptr(HP), _ptr(HS), _ptr(HP_b), _ptr(HS_b), _ptr(qsim_d), C.byref(cost_d)]
        tangent = mode == 6
    err = []

    def call():
        try:
            (lib.ref_run_hyper if hyper else lib.ref_run_d if tangent else lib.ref_run)(*cargs)
        except Exception as e:  # pragma: no cover
            err.append(e)

    # the reference keeps (nrow,ncol,16) automatic arrays on the stack (mwd_cost.f90:185-186)
    old = threading.stack_size(min(2 << 30, max(64 << 20, 64 * nrow * ncol * 4 * 12)))
    th = threading.Thread(target=call)
    th.start()
    th.join()
    threading.stack_size(old)
    if err:
        raise err[0]
    return dict(qsim=qsim[:ng], cost=float(costs[0]), cost_jobs=float(costs[1]), cost_jreg=float(costs[2]),
                fstates=unpack(fstates, STATE_NAMES), parameters=unpack(pout, PARAM_NAMES),
                states=unpack(sout, STATE_NAMES), parameters_b=unpack(p_b, PARAM_NAMES),
                states_b=unpack(s_b, STATE_NAMES), elapsed=elapsed.value,
                cost_d=float(cost_d.value) if tangent else None, qsim_d=qsim_d[:ng] if tangent else None,
                hyper_parameters_b={k: HP_b[:, 0, i].copy() for i, k in enumerate(PARAM_NAMES)} if hyper else None,
                hyper_states_b={k: HS_b[:, 0, i].copy() for i, k in enumerate(STATE_NAMES)} if hyper else None)
