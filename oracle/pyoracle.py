"""ctypes binding to oracle/liboracle.so (the plain-C CPU restatement, oracle/smash_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .refbind import GLB_P, GLB_S, GNP, GNS, GUB_P, GUB_S, JOBS, JREG, STRUCTURES, pack, unpack

_HERE = os.path.dirname(os.path.abspath(__file__))


def _config_class(real):
    class Cfg(C.Structure):
        _fields_ = [("structure", C.c_int), ("nrow", C.c_int), ("ncol", C.c_int), ("nt", C.c_int), ("ng", C.c_int),
                    ("denormalize_forward", C.c_int), ("optimize_start_step", C.c_int), ("njf", C.c_int),
                    ("jobs_fun", C.c_int * 8), ("wjobs_fun", real * 8), ("njr", C.c_int),
                    ("jreg_fun", C.c_int * 4), ("wjreg_fun", real * 4), ("wjreg", real),
                    ("dt", real), ("dx", real), ("optim_parameters", C.c_int * GNP),
                    ("optim_states", C.c_int * GNS), ("lb_parameters", real * GNP),
                    ("ub_parameters", real * GNP), ("lb_states", real * GNS), ("ub_states", real * GNS)]
    return Cfg


OrcConfig = _config_class(C.c_float)
OrcConfig64 = _config_class(C.c_double)      # liboracle64.so: the same statements compiled in double (oracle/Makefile FP64_DEFS)

_lib = None
_lib64 = None


class _Prefixed:
    """liboracle64.so exports orc64_*: present them under the names run() uses."""

    def __init__(self, dll):
        for k in ("forward", "forward_b", "forward_d", "set_domain_outputs"):
            f = getattr(dll, "orc64_" + k)
            f.restype = C.c_int if k != "set_domain_outputs" else None
            setattr(self, "orc_" + k, f)


def lib64():
    global _lib64
    if _lib64 is None:
        path = os.path.join(_HERE, "liboracle64.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle64.so"])
        _lib64 = _Prefixed(C.CDLL(path))
    return _lib64


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_forward.restype = C.c_int
        _lib.orc_forward_b.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run(structure, mesh, dt, prcp, pet, qobs, params, states, *, adjoint=False, params_bgd=None,
        states_bgd=None, denormalize_forward=False, optimize_start_step=1, jobs_fun=("nse",),
        wjobs_fun=(1.0,), jreg_fun=(), wjreg_fun=(), wjreg=0.0, wgauge=None, optim_parameters=None,
        optim_states=None, lb_parameters=None, ub_parameters=None, lb_states=None, ub_states=None,
        cost_b=1.0, save_domain=False, params_d=None, states_d=None, fp64=False):
    """Same calling convention and result dict as oracle.refbind.run (dense forcing only).
    save_domain: also return qsim_domain / net_prcp_domain (nrow, ncol, nt), -99 on inactive cells.
    params_d / states_d (dicts): run the tangent model (orc_forward_d) along that direction -> cost_d, qsim_d.
    fp64: evaluate the same statements in double (liboracle64.so) -- the common "truth" the fp32 builds are ranked against
    (tools/accuracy_report.py); inputs are the fp32 values, widened."""
    from smash_amd.synth import PARAM_NAMES, STATE_NAMES
    L = lib64() if fp64 else lib()
    real, npreal = (C.c_double, np.float64) if fp64 else (C.c_float, np.float32)
    nrow, ncol, ng = mesh.nrow, mesh.ncol, mesh.ng
    nt = prcp.shape[2]
    cfg = (OrcConfig64 if fp64 else OrcConfig)()
    cfg.structure, cfg.nrow, cfg.ncol, cfg.nt, cfg.ng = STRUCTURES[structure], nrow, ncol, nt, ng
    cfg.denormalize_forward, cfg.optimize_start_step = int(denormalize_forward), optimize_start_step
    cfg.njf = len(jobs_fun)
    for i, j in enumerate(jobs_fun):
        cfg.jobs_fun[i], cfg.wjobs_fun[i] = JOBS[j], wjobs_fun[i]
    cfg.njr = len(jreg_fun)
    for i, j in enumerate(jreg_fun):
        cfg.jreg_fun[i], cfg.wjreg_fun[i] = JREG[j], wjreg_fun[i]
    cfg.wjreg, cfg.dt, cfg.dx = wjreg, dt, mesh.dx
    for i in range(GNP):
        cfg.optim_parameters[i] = 0 if optim_parameters is None else int(optim_parameters[i])
        cfg.lb_parameters[i] = (GLB_P if lb_parameters is None else lb_parameters)[i]
        cfg.ub_parameters[i] = (GUB_P if ub_parameters is None else ub_parameters)[i]
    for i in range(GNS):
        cfg.optim_states[i] = 0 if optim_states is None else int(optim_states[i])
        cfg.lb_states[i] = (GLB_S if lb_states is None else lb_states)[i]
        cfg.ub_states[i] = (GUB_S if ub_states is None else ub_states)[i]
    P = np.asfortranarray(pack(params, PARAM_NAMES, nrow, ncol), dtype=npreal)
    S = np.asfortranarray(pack(states, STATE_NAMES, nrow, ncol), dtype=npreal)
    Pb = np.asfortranarray(pack(params_bgd, PARAM_NAMES, nrow, ncol), dtype=npreal) if params_bgd is not None else P.copy(order="F")
    Sb = np.asfortranarray(pack(states_bgd, STATE_NAMES, nrow, ncol), dtype=npreal) if states_bgd is not None else S.copy(order="F")
    wg = np.full(max(ng, 1), 1.0 / max(ng, 1), npreal) if wgauge is None else np.ascontiguousarray(wgauge, npreal)
    f32 = lambda a: np.asfortranarray(a, dtype=npreal)
    i32 = lambda a: np.asfortranarray(a, dtype=np.int32)
    flwdir, flwacc, path, active, gpos = i32(mesh.flwdir), i32(mesh.flwacc), i32(mesh.path), i32(mesh.active_cell), i32(mesh.gauge_pos)
    area = np.ascontiguousarray(mesh.area, npreal)
    prcp, pet, qobs = f32(prcp), f32(pet), f32(qobs)
    qsim = np.zeros((max(ng, 1), nt), npreal, order="F")
    costs = np.zeros(3, npreal)
    fstates = np.zeros((nrow, ncol, GNS), npreal, order="F")
    p_b = np.zeros((nrow, ncol, GNP), npreal, order="F")
    s_b = np.zeros((nrow, ncol, GNS), npreal, order="F")
    common = [C.byref(cfg), _p(flwdir), _p(flwacc), _p(path), _p(active), _p(gpos), _p(area), _p(prcp), _p(pet),
              _p(qobs), _p(wg), _p(P), _p(Pb), _p(S), _p(Sb)]
    if params_d is not None or states_d is not None:
        zp, zs = np.zeros((nrow, ncol, GNP), npreal, order="F"), np.zeros((nrow, ncol, GNS), npreal, order="F")
        P_d = np.asfortranarray(pack(params_d, PARAM_NAMES, nrow, ncol), dtype=npreal) if params_d is not None else zp
        S_d = np.asfortranarray(pack(states_d, STATE_NAMES, nrow, ncol), dtype=npreal) if states_d is not None else zs
        qsim_d = np.zeros((max(ng, 1), nt), npreal, order="F")
        cost_d = real(0.0)
        L.orc_forward_d.restype = C.c_int
        rc = L.orc_forward_d(C.byref(cfg), _p(flwdir), _p(flwacc), _p(path), _p(active), _p(gpos), _p(area), _p(prcp), _p(pet),
                             _p(qobs), _p(wg), _p(P), _p(P_d), _p(Pb), _p(S), _p(S_d), _p(Sb), _p(qsim), _p(qsim_d), _p(costs),
                             C.byref(cost_d))
        if rc != 0:
            raise RuntimeError(f"oracle returned {rc}")
        return dict(qsim=qsim[:ng], qsim_d=qsim_d[:ng], cost_d=float(cost_d.value), cost_jobs=float(costs[1]),
                    parameters=unpack(P, PARAM_NAMES), states=unpack(S, STATE_NAMES),
                    parameters_d=unpack(P_d, PARAM_NAMES), states_d=unpack(S_d, STATE_NAMES))
    qdom = pdom = None
    if save_domain and not adjoint:
        qdom = np.full((nrow, ncol, nt), -99.0, npreal, order="F")
        pdom = np.full((nrow, ncol, nt), -99.0, npreal, order="F")
        L.orc_set_domain_outputs(_p(qdom), _p(pdom))
    try:
        if adjoint:
            rc = L.orc_forward_b(*common, real(cost_b), _p(qsim), _p(costs), _p(p_b), _p(s_b))
        else:
            rc = L.orc_forward(*common, _p(qsim), _p(costs), _p(fstates))
    finally:
        L.orc_set_domain_outputs(None, None)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return dict(qsim_domain=qdom, net_prcp_domain=pdom,
                qsim=qsim[:ng], cost=float(costs[0]), cost_jobs=float(costs[1]), cost_jreg=float(costs[2]),
                fstates=unpack(fstates, STATE_NAMES), parameters=unpack(P, PARAM_NAMES),
                states=unpack(S, STATE_NAMES), parameters_b=unpack(p_b, PARAM_NAMES),
                states_b=unpack(s_b, STATE_NAMES))
