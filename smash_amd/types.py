"""Python-side mirrors of the derived types the path touches.  Same field names as
the reference's f90wrap classes so call sites read the same:

    SetupDT / Optimize_SetupDT   smash/solver/derived_type/mwd_setup.f90:57-157
    MeshDT                       mwd_mesh.f90:45-72       (path / gauge_pos 0-based as Python sees them)
    Input_DataDT                 mwd_input_data.f90:32-50
    ParametersDT / StatesDT      mwd_parameters.f90:58-86 / mwd_states.f90:49-69
    OutputDT                     mwd_output.f90:36-57
"""
from __future__ import annotations

import copy

import numpy as np

from .synth import PARAM_DEFAULTS, PARAM_NAMES, STATE_DEFAULTS, STATE_NAMES

STRUCTURES = {"gr-a": 1, "gr-b": 2, "gr-c": 3, "gr-d": 4, "vic-a": 5}
JOBS_FUN = {"nse": 1, "kge": 2, "kge2": 3, "se": 4, "rmse": 5, "logarithmic": 6}
JREG_FUN = {"prior": 1, "smoothing": 2, "hard_smoothing": 3}

# md_constant.f90:70-136
GLB_PARAMETERS = np.array([1e-6] * 6 + [-50.0] + [1e-6] * 9, dtype=np.float32)
GUB_PARAMETERS = np.array([1e2, 1e3, 1e3, 1e3, 1e4, 0.999999, 50.0, 1e1, 2e3, 2e3, 2e3, 1e4, 0.999999, 30.0, 0.999999, 1e3],
                          dtype=np.float32)
GLB_STATES = np.array([1e-6] * 8, dtype=np.float32)
GUB_STATES = np.array([0.999999] * 7 + [10000.0], dtype=np.float32)


class Optimize_SetupDT:
    def __init__(self, ng: int):
        self.jobs_fun = []            # njf = len(jobs_fun); plain Model.run() has none (mwd_setup.f90:236)
        self.wjobs_fun = []
        self.wjreg = 0.0
        self.jreg_fun = []
        self.wjreg_fun = []
        self.denormalize_forward = False
        self.optimize_start_step = 1  # 1-based like the Fortran field
        self.maxiter = 100
        self.optim_parameters = np.zeros(16, np.int32)
        self.optim_states = np.zeros(8, np.int32)
        self.lb_parameters = GLB_PARAMETERS.copy()
        self.ub_parameters = GUB_PARAMETERS.copy()
        self.lb_states = GLB_STATES.copy()
        self.ub_states = GUB_STATES.copy()
        self.wgauge = np.full(max(ng, 1), 1.0 / max(ng, 1), np.float32)[:ng]
        self.mapping = "uniform"      # "hyper-linear" / "hyper-polynomial": smash_amd.hyper_forward(_b, _d)
        self.nhyper = 0               # 1 + nd (hyper-linear) or 1 + 2 nd (hyper-polynomial), mwd_setup.f90 / _optimize.py

    @property
    def njf(self):
        return len(self.jobs_fun)

    @property
    def njr(self):
        return len(self.jreg_fun)


class SetupDT:
    def __init__(self, nd: int = 0, ng: int = 0, *, structure: str = "gr-a", dt: float = 3600.0, ntime_step: int = 0,
                 sparse_storage: bool = False, save_qsim_domain: bool = False, save_net_prcp_domain: bool = False):
        self.structure = structure
        self.nd = int(nd)                                       # catchment descriptors (input_data.descriptor (nrow, ncol, nd))
        self.dt = float(dt)
        self.ntime_step = int(ntime_step)
        self.sparse_storage = bool(sparse_storage)
        self.save_qsim_domain = bool(save_qsim_domain)          # mwd_setup.f90:144-145
        self.save_net_prcp_domain = bool(save_net_prcp_domain)
        self.optimize = Optimize_SetupDT(ng)

    def copy(self):
        return copy.deepcopy(self)


class MeshDT:
    def __init__(self, setup: SetupDT, nrow: int, ncol: int, ng: int):
        self.nrow, self.ncol, self.ng = int(nrow), int(ncol), int(ng)
        self.dx = 1000.0
        self.flwdir = np.full((nrow, ncol), -99, np.int32, order="F")
        self.flwacc = np.full((nrow, ncol), -99, np.int32, order="F")
        self.path = np.full((2, nrow * ncol), -99, np.int32, order="F")
        self.active_cell = np.ones((nrow, ncol), np.int32, order="F")
        self.gauge_pos = np.zeros((ng, 2), np.int32, order="F")
        self.area = np.zeros(ng, np.float32)

    @property
    def nac(self):
        return int(np.count_nonzero(self.active_cell == 1))

    @classmethod
    def from_synth(cls, setup, m):
        o = cls(setup, m.nrow, m.ncol, m.ng)
        o.dx = m.dx
        o.flwdir, o.flwacc, o.path, o.active_cell = m.flwdir, m.flwacc, m.path, m.active_cell
        o.gauge_pos, o.area = m.gauge_pos, m.area
        return o


class Input_DataDT:
    def __init__(self, setup: SetupDT, mesh: MeshDT):
        nt = setup.ntime_step
        self.qobs = np.full((mesh.ng, nt), -99.0, np.float32, order="F")
        self.prcp = self.pet = self.sparse_prcp = self.sparse_pet = None
        if setup.sparse_storage:
            self.sparse_prcp = np.full((mesh.nac, nt), -99.0, np.float32, order="F")
            self.sparse_pet = np.full((mesh.nac, nt), -99.0, np.float32, order="F")
        else:
            self.prcp = np.full((mesh.nrow, mesh.ncol, nt), -99.0, np.float32, order="F")
            self.pet = np.full((mesh.nrow, mesh.ncol, nt), -99.0, np.float32, order="F")
        self.descriptor = np.full((mesh.nrow, mesh.ncol, getattr(setup, "nd", 0)), -99.0, np.float32, order="F")


class _Fields:
    _names = ()
    _defaults = {}

    def __init__(self, mesh):
        for k in self._names:
            setattr(self, k, np.full((mesh.nrow, mesh.ncol), self._defaults[k], np.float32, order="F"))

    def copy(self):
        o = object.__new__(type(self))
        for k in self._names:
            setattr(o, k, np.asfortranarray(getattr(self, k).copy(order="F")))
        return o

    def as_dict(self):
        return {k: getattr(self, k) for k in self._names}

    @classmethod
    def from_dict(cls, mesh, d):
        o = cls(mesh)
        for k in cls._names:
            if k in d:
                setattr(o, k, np.asfortranarray(d[k], dtype=np.float32).copy(order="F"))
        return o


class ParametersDT(_Fields):
    _names = PARAM_NAMES
    _defaults = PARAM_DEFAULTS


class StatesDT(_Fields):
    _names = STATE_NAMES
    _defaults = STATE_DEFAULTS


class _HyperFields:
    """Hyper_ParametersDT / Hyper_StatesDT (mwd_parameters.f90:88-116, mwd_states.f90:71-91): one (nhyper, 1) column of coefficients
    per field; row 0 the intercept, then (a_j) for hyper-linear or (a_j, b_j) pairs for hyper-polynomial, j over the descriptors."""
    _names = ()

    def __init__(self, setup):
        nh = setup.optimize.nhyper
        for k in self._names:
            setattr(self, k, np.zeros((nh, 1), np.float32, order="F"))

    def copy(self):
        o = object.__new__(type(self))
        for k in self._names:
            setattr(o, k, np.asfortranarray(getattr(self, k).copy(order="F")))
        return o

    def matrix(self):
        """(nhyper, nfields) column-major, md_constant field order: what include/smashx.h calls a hyper matrix."""
        return np.asfortranarray(np.stack([np.asarray(getattr(self, k), np.float32).reshape(-1) for k in self._names], axis=1))

    def set_matrix(self, a):
        for i, k in enumerate(self._names):
            setattr(self, k, np.asfortranarray(np.asarray(a, np.float32)[:, i].reshape(-1, 1)))

    @classmethod
    def from_dict(cls, setup, d):
        o = cls(setup)
        for k in cls._names:
            if k in d:
                setattr(o, k, np.asfortranarray(np.asarray(d[k], np.float32).reshape(-1, 1)))
        return o


class Hyper_ParametersDT(_HyperFields):
    _names = PARAM_NAMES


class Hyper_StatesDT(_HyperFields):
    _names = STATE_NAMES


class OutputDT:
    def __init__(self, setup: SetupDT, mesh: MeshDT):
        self.qsim = np.full((mesh.ng, setup.ntime_step), -99.0, np.float32, order="F")
        # optional whole-domain stores (OutputDT_initialise, mwd_output.f90:80-112)
        shp = (mesh.nac, setup.ntime_step) if setup.sparse_storage else (mesh.nrow, mesh.ncol, setup.ntime_step)
        key = ("sparse_" if setup.sparse_storage else "")
        if setup.save_qsim_domain:
            setattr(self, key + "qsim_domain", np.full(shp, -99.0, np.float32, order="F"))
        if setup.save_net_prcp_domain:
            setattr(self, key + "net_prcp_domain", np.full(shp, -99.0, np.float32, order="F"))
        self.cost = 0.0
        self.cost_jobs = 0.0
        self.cost_jreg = 0.0
        self.fstates = StatesDT(mesh)

    def copy(self):
        return copy.deepcopy(self)
