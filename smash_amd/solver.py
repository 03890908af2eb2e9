"""Host-side mirror of the reference's wrapped boundary for the hot path:

    forward    smash/solver/forward/mw_forward.f90:18-39   -> base_forward   (forward.f90:1-80)
    forward_b  smash/solver/forward/mw_forward.f90:41-68   -> base_forward_b (forward_db.f90:10648-10936)

Same names, argument order and in/out behaviour as the f90wrap functions the reference's Python calls
(smash/core/model.py:490, smash/core/net.py:1044): parameters / states are modified in place
(denormalised on return when setup.optimize.denormalize_forward), results land in ``output`` and in the
``*_b`` objects, and -- like the reference -- nothing is raised for a cost that is NaN.  Differences a
caller can see: unsupported options raise SmashxError instead of being silently ignored, and only the
fields the structure uses are touched.

All arithmetic happens in libsmashx (HIP, gfx950) behind the C ABI of include/smashx.h.
"""
from __future__ import annotations

import ctypes as C

import zlib

import numpy as np

from . import _lib
from .types import JOBS_FUN, JREG_FUN, STRUCTURES
from .synth import PARAM_NAMES, STATE_NAMES


# hourly share of the daily PET in the reference's reader (smash/core/_constant.py:47-75)
RATIO_PET_HOURLY = np.array([0, 0, 0, 0, 0, 0, 0, 0.035, 0.062, 0.079, 0.097, 0.11, 0.117, 0.117, 0.11, 0.097, 0.079, 0.062, 0.035,
                             0, 0, 0, 0, 0], dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_config(setup, mesh, chunk_steps=0, pipe_steps=0, group_size=0, device=-1, tile=None):
    cfg = _lib.Config()
    cfg.structure, cfg.nrow, cfg.ncol = STRUCTURES[setup.structure], mesh.nrow, mesh.ncol
    cfg.nt, cfg.ng, cfg.dt, cfg.dx = setup.ntime_step, mesh.ng, setup.dt, mesh.dx
    cfg.chunk_steps, cfg.pipe_steps, cfg.group_size, cfg.device = chunk_steps, pipe_steps, group_size, device
    for i in range(4):
        cfg.tile[i] = int(tile[i]) if tile is not None else 0
    return cfg


def _f32(a):
    return np.asfortranarray(a, dtype=np.float32)


def _i32(a):
    return np.asfortranarray(a, dtype=np.int32)


class Comm:
    """One RCCL communicator per process (= per GPU) for the native exchange of boundary series (include/smashx.h
    "native exchange").  Rank 0 draws the id (Comm.unique_id()); the launcher hands it to every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * 128)()
        _lib.check(_lib.lib().smashx_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes, rank: int, nranks: int, device: int = -1):
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        self.handle = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        _lib.check(_lib.lib().smashx_comm_create(buf, int(rank), int(nranks), int(device), C.byref(self.handle)))

    def allreduce_sum(self, values):
        v = np.ascontiguousarray(values, np.float64).copy()
        _lib.check(_lib.lib().smashx_comm_allreduce_sum(self.handle, _ptr(v), int(v.size)))
        return v

    def info(self):
        """{nranks, version}: what the communicator itself reports (ncclCommCount, ncclGetVersion)."""
        n, v = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().smashx_comm_info(self.handle, C.byref(n), C.byref(v)))
        return {"nranks": n.value, "version": v.value}

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().smashx_comm_destroy(self.handle)
            self.handle = None


class Solver:
    """A libsmashx plan: routing schedule + HBM-resident forcing for one (setup, mesh, input_data)."""

    def __init__(self, setup, mesh, *, chunk_steps: int = 0, pipe_steps: int = 0, group_size: int = 0, device: int = -1,
                 tile=None, owner_mask=None):
        """tile = (row0, row1, col0, col1): this plan only owns that rectangle of the grid (multi-GPU, see
        smash_amd.tiles); owner_mask (nrow, ncol), 1 = owned, does the same for an arbitrary partition (sub-catchments);
        mesh and field arrays stay global-sized."""
        L = _lib.lib()
        self.nrow, self.ncol, self.nt, self.ng = mesh.nrow, mesh.ncol, setup.ntime_step, mesh.ng
        self.structure = setup.structure
        if setup.structure not in STRUCTURES:
            raise _lib.SmashxError(_lib.E_UNSUPPORTED, f"structure {setup.structure!r} is not on the hot path yet")
        cfg = make_config(setup, mesh, chunk_steps, pipe_steps, group_size, device, tile)
        self._keep = [_i32(mesh.flwdir), _i32(mesh.flwacc), _i32(mesh.active_cell), _i32(mesh.path),
                      _i32(np.asarray(mesh.gauge_pos).reshape(-1, 2)), np.ascontiguousarray(mesh.area, np.float32)]
        m = _lib.Mesh(*[_ptr(a) for a in self._keep])
        if owner_mask is not None:
            self._keep.append(_i32(owner_mask))
            m.owner_mask = self._keep[-1].ctypes.data
        self._h = C.c_void_p()
        _lib.check(L.smashx_plan_create(C.byref(cfg), C.byref(m), C.byref(self._h)))
        self.ncells = L.smashx_plan_ncells(self._h)
        self._sig = self.signature(setup, mesh)

    @staticmethod
    def signature(setup, mesh):
        """Everything a cached plan was built from: sizes, dt, dx and the contents of the mesh arrays."""
        # (checksums of the arrays in their own memory order: no transposed copy of a Fortran-ordered 2048^2 plane per call)
        def crc(a):
            a = np.asarray(a)
            if not (a.flags.c_contiguous or a.flags.f_contiguous):
                a = np.ascontiguousarray(a)
            return (a.shape, a.dtype.str, bool(a.flags.f_contiguous and not a.flags.c_contiguous), zlib.crc32(memoryview(a.ravel(order="K")).cast("B")))
        h = hash(tuple(crc(getattr(mesh, k)) for k in ("flwdir", "flwacc", "active_cell", "gauge_pos", "area")))
        return (setup.structure, setup.ntime_step, float(setup.dt), mesh.nrow, mesh.ncol, mesh.ng, float(mesh.dx), bool(setup.sparse_storage), h)

    FULL_HASH_ELEMENTS = 1 << 24        # fields up to 64 MB are hashed whole

    @staticmethod
    def forcing_fingerprint(prcp, pet):
        """Hash of the forcing arrays, recomputed on every call: values written in place, or new arrays that happen to land at the
        old address, must not be served the forcing already resident in HBM.  Fields up to FULL_HASH_ELEMENTS values are hashed
        whole; larger ones by a strided sample of ~65 k values per field (hashing 70 GB per call would cost more than the sweep), so
        an in-place edit of a small window of a LARGE field can go unnoticed: after such an edit call invalidate_forcing(input_data)
        (or Solver.set_forcing again) -- the reference has no such cache because it has no device copy."""
        out = []
        for a in (prcp, pet):
            a = np.asarray(a)
            flat = a.reshape(-1, order="A") if (a.flags.f_contiguous or a.flags.c_contiguous) else a.ravel()
            step = 1 if flat.size <= Solver.FULL_HASH_ELEMENTS else max(1, flat.size // 65536)
            out.append((a.shape, flat[::step].tobytes()))
        return hash(tuple(out))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().smashx_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- residency ---------------------------------------------------------------------------
    def cell_order(self):
        rows = np.zeros(self.ncells, np.int32)
        cols = np.zeros(self.ncells, np.int32)
        _lib.check(_lib.lib().smashx_plan_cell_order(self._h, _ptr(rows), _ptr(cols)))
        return rows, cols

    def set_forcing(self, prcp, pet, sparse=False):
        p, e = _f32(prcp), _f32(pet)
        self._fp = Solver.forcing_fingerprint(prcp, pet)
        _lib.check(_lib.lib().smashx_set_forcing(self._h, _ptr(p), _ptr(e), int(bool(sparse))))

    def set_forcing_device_block(self, t0, t1, d_prcp_ptr, d_pet_ptr):
        self._forcing_external = True
        _lib.check(_lib.lib().smashx_set_forcing_device_block(self._h, int(t0), int(t1), C.c_void_p(d_prcp_ptr),
                                                              C.c_void_p(d_pet_ptr)))

    def set_forcing_layout(self, compact=True, prcp_factor=0.1, pet_ratio=None, pet_hour0=1):
        """Lossless compact residency of the forcing (include/smashx.h smashx_set_forcing_layout): uint16 rain counts x
        prcp_factor + daily PET x pet_ratio[(t + pet_hour0) % 24], verified bit for bit when the forcing is set.
        pet_ratio defaults to the reference's RATIO_PET_HOURLY (smash/core/_constant.py:47-75); pet_hour0 = 1 is a run that
        starts at midnight (the first step is start_time + dt)."""
        lay = _lib.ForcingLayout()
        lay.compact, lay.prcp_factor, lay.pet_hour0 = int(bool(compact)), float(np.float32(prcp_factor)), int(pet_hour0)
        r = RATIO_PET_HOURLY if pet_ratio is None else np.asarray(pet_ratio, np.float32)
        for h in range(24):
            lay.pet_ratio[h] = float(r[h])
        _lib.check(_lib.lib().smashx_set_forcing_layout(self._h, C.byref(lay)))

    def forcing_info(self):
        c, b = C.c_int(0), C.c_double(0.0)
        _lib.check(_lib.lib().smashx_forcing_info(self._h, C.byref(c), C.byref(b)))
        return {"layout": "compact: uint16 rain counts + daily PET x hourly ratio (lossless, verified bit for bit)" if c.value
                else "fp32 rows", "resident_bytes_per_cellstep": round(b.value, 4)}

    def set_qobs(self, qobs):
        q = _f32(qobs)
        _lib.check(_lib.lib().smashx_set_qobs(self._h, _ptr(q)))

    # -- tiles ---------------------------------------------------------------------------------
    def halo_counts(self):
        a, b = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().smashx_halo_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def halo_edges(self):
        no, ni = self.halo_counts()
        arr = [np.zeros(max(n, 1), np.int32) for n in (no, no, ni, ni)]
        _lib.check(_lib.lib().smashx_halo_edges(self._h, *[_ptr(a) for a in arr]))
        return arr[0][:no], arr[1][:no], arr[2][:ni], arr[3][:ni]

    def chunking(self):
        a, b = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().smashx_plan_chunking(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def hbm(self):
        """{free_at_plan, total, held} in bytes (smashx_plan_hbm): free HBM when the storage-chunk length was chosen, the device's
        total, what the plan holds now."""
        v = (C.c_double * 3)()
        _lib.check(_lib.lib().smashx_plan_hbm(self._h, v))
        return {"free_at_plan": v[0], "total": v[1], "held": v[2]}

    def set_domain_outputs(self, qsim_domain=None, net_prcp_domain=None, sparse=False):
        """Host arrays the following forward sweeps fill (OutputDT%qsim_domain / net_prcp_domain or sparse_ forms)."""
        for a in (qsim_domain, net_prcp_domain):
            if a is not None and not (a.dtype == np.float32 and a.flags.f_contiguous):
                raise _lib.SmashxError(_lib.E_ARG, "domain outputs must be Fortran-ordered float32 arrays")
        self._dom_keep = (qsim_domain, net_prcp_domain)
        fn = _lib.lib().smashx_set_domain_outputs
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.check(fn(self._h, None if qsim_domain is None else qsim_domain.ctypes.data,
                      None if net_prcp_domain is None else net_prcp_domain.ctypes.data, int(bool(sparse))))

    def tangent_terms(self):
        """(jobs_d, jreg_d) of the last forward_d on this plan (include/smashx.h smashx_tangent_terms): over a decomposition the
        parts' jobs_d add up and jreg_d -- the whole grid's on every part -- enters once."""
        a, b = C.c_float(0.0), C.c_float(0.0)
        _lib.check(_lib.lib().smashx_tangent_terms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def group_times(self):
        """Diagnostics (SMASHX_TRACE_GROUPS=1): (ticks[2][groups][2] at 100 MHz, round_of_group[groups])."""
        ng = self.timing()["n_groups"]
        out = np.zeros((2, ng, 2), np.int64)
        rnd = np.zeros(ng, np.int32)
        _lib.check(_lib.lib().smashx_debug_group_times(self._h, out.ctypes.data_as(C.POINTER(C.c_longlong)), _ptr(rnd)))
        return out, rnd

    def set_halo(self, out_ptr, in_ptr, fn):
        """fn(phase, t0, nsteps) -> 0; out_ptr / in_ptr: device addresses of the message buffers."""
        def tramp(user, phase, t0, nsteps):
            try:
                return int(fn(phase, t0, nsteps) or 0)
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        self._halo_cb = _lib.HALO_FN(tramp)
        _lib.check(_lib.lib().smashx_set_halo(self._h, C.c_void_p(out_ptr), C.c_void_p(in_ptr), self._halo_cb, None))

    def set_median_slots(self, nslots, slot_of_gauge, reduce_fn=None):
        """The median over the negative-weight gauges of a decomposition (include/smashx.h "cost terms that span the tiles";
        tiles.median_slots builds the arguments).  reduce_fn(values: float32 array, in place) sums the slot values over the tiles;
        not needed with the native exchange (set_exchange), which all-reduces them on the routing stream.  Call before set_options."""
        sl = np.ascontiguousarray(slot_of_gauge, np.int32) if self.ng else np.full(1, -1, np.int32)
        cb = None
        if reduce_fn is not None:
            def tramp(user, vals, n):
                try:
                    a = np.ctypeslib.as_array(vals, shape=(n,))
                    reduce_fn(a)
                    return 0
                except Exception:  # pragma: no cover
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = _lib.REDUCE_FN(tramp)
        self._median_cb = cb
        fn = _lib.lib().smashx_set_median_slots
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.check(fn(self._h, int(nslots), _ptr(sl), C.cast(cb, C.c_void_p) if cb is not None else None, None))

    def set_exchange(self, comm, out_peer, in_peer):
        """Native exchange (smashx_set_exchange): comm = a Comm (or None to unset); out_peer / in_peer = the rank owning the
        other end of every out / in boundary edge, in the order of halo_edges()."""
        op, ip = np.ascontiguousarray(out_peer, np.int32), np.ascontiguousarray(in_peer, np.int32)
        self._comm = comm
        _lib.check(_lib.lib().smashx_set_exchange(self._h, comm.handle if comm is not None else None, _ptr(op), _ptr(ip)))

    def set_options(self, opt):
        o = _lib.Options()
        o.denormalize_forward = int(bool(opt.denormalize_forward))
        o.optimize_start_step = int(opt.optimize_start_step)
        o.njf = len(opt.jobs_fun)
        for i, j in enumerate(opt.jobs_fun):
            if j not in JOBS_FUN:
                raise _lib.SmashxError(_lib.E_UNSUPPORTED, f"jobs_fun {j!r} is outside the hot path")
            o.jobs_fun[i] = JOBS_FUN[j]
            o.wjobs_fun[i] = float(opt.wjobs_fun[i])
        o.njr = len(opt.jreg_fun)
        for i, j in enumerate(opt.jreg_fun):
            o.jreg_fun[i] = JREG_FUN.get(j, 0)
            o.wjreg_fun[i] = float(opt.wjreg_fun[i])
        o.wjreg = float(opt.wjreg)
        for i in range(16):
            o.optim_parameters[i] = int(opt.optim_parameters[i])
            o.lb_parameters[i] = float(opt.lb_parameters[i])
            o.ub_parameters[i] = float(opt.ub_parameters[i])
        for i in range(8):
            o.optim_states[i] = int(opt.optim_states[i])
            o.lb_states[i] = float(opt.lb_states[i])
            o.ub_states[i] = float(opt.ub_states[i])
        wg = np.ascontiguousarray(opt.wgauge, np.float32) if self.ng else np.zeros(1, np.float32)
        o.wgauge = _ptr(wg)
        _lib.check(_lib.lib().smashx_set_options(self._h, C.byref(o)))

    # -- calls ---------------------------------------------------------------------------------
    @staticmethod
    def _pack(obj, names, struct_cls, only=None):
        s = struct_cls()
        keep = []
        for i, k in enumerate(names):
            a = getattr(obj, k, None) if obj is not None and (only is None or k in only) else None
            if a is None:
                s.f[i] = None
                continue
            if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.f_contiguous):
                a = np.asfortranarray(a, dtype=np.float32)
                setattr(obj, k, a)
            keep.append(a)
            s.f[i] = a.ctypes.data
        return s, keep

    def upload(self, parameters, states, parameters_bgd=None, states_bgd=None, only=None):
        """only = names of the fields that changed since the last upload (the others keep their device copies)."""
        P, k1 = self._pack(parameters, PARAM_NAMES, _lib.Parameters, only)
        S, k2 = self._pack(states, STATE_NAMES, _lib.States, only)
        PB, k3 = self._pack(parameters_bgd, PARAM_NAMES, _lib.Parameters) if parameters_bgd is not None else (None, None)
        SB, k4 = self._pack(states_bgd, STATE_NAMES, _lib.States) if states_bgd is not None else (None, None)
        _lib.check(_lib.lib().smashx_upload(self._h, C.byref(P), C.byref(PB) if PB is not None else None, C.byref(S),
                                            C.byref(SB) if SB is not None else None))

    # -- control vector of the calibration, packed / unpacked on the device (mw_optimize.f90:679-777) ----------------------------
    def control_size(self):
        return int(_lib.lib().smashx_control_size(self._h))

    def control_set(self, x):
        x = np.ascontiguousarray(x, np.float64)
        _lib.check(_lib.lib().smashx_control_set(self._h, _ptr(x)))

    def control_get(self):
        x = np.zeros(self.control_size(), np.float64)
        _lib.check(_lib.lib().smashx_control_get(self._h, _ptr(x)))
        return x

    def control_gradient(self):
        g = np.zeros(self.control_size(), np.float64)
        _lib.check(_lib.lib().smashx_control_gradient(self._h, _ptr(g)))
        return g

    def cost_and_qsim(self, output):
        """cost + discharge of the last sweep only (no field comes back)."""
        qs = np.zeros((self.ng, self.nt), np.float32, order="F") if self.ng else None
        costs = _lib.Costs()
        _lib.check(_lib.lib().smashx_download(self._h, 0, None, None, _ptr(qs), C.byref(costs), None, None, None))
        if output is not None:
            if qs is not None:
                output.qsim = qs
            output.cost, output.cost_jobs, output.cost_jreg = float(costs.cost), float(costs.cost_jobs), float(costs.cost_jreg)
        return float(costs.cost)

    def sweep(self, adjoint=False, cost_b=1.0):
        _lib.check(_lib.lib().smashx_sweep(self._h, int(bool(adjoint)), C.c_float(cost_b)))

    def timing(self):
        t = _lib.Timing()
        _lib.check(_lib.lib().smashx_get_timing(self._h, C.byref(t)))
        d = {k: getattr(t, k) for k, _ in _lib.Timing._fields_ if k != "cellsteps"}
        for i, k in enumerate(("vert_fwd", "route_fwd", "route_adj", "vert_adj")):
            d[k + "_cellsteps"] = float(t.cellsteps[i])
        return d

    def download(self, adjoint, parameters, states, output, parameters_b=None, states_b=None, only_b=None):
        """parameters / states None: nothing but cost, discharge and gradients comes back; only_b = the gradient fields wanted."""
        P, k1 = self._pack(parameters, PARAM_NAMES, _lib.Parameters)
        S, k2 = self._pack(states, STATE_NAMES, _lib.States)
        qs = np.zeros((self.ng, self.nt), np.float32, order="F") if self.ng else None
        costs = _lib.Costs()
        F = PB = SB = None
        kf = kp = ks = None
        if not adjoint and output is not None:
            F, kf = self._pack(output.fstates, STATE_NAMES, _lib.States)
        if adjoint:
            PB, kp = self._pack(parameters_b, PARAM_NAMES, _lib.Parameters, only_b)
            SB, ks = self._pack(states_b, STATE_NAMES, _lib.States, only_b)
        _lib.check(_lib.lib().smashx_download(self._h, int(bool(adjoint)), C.byref(P), C.byref(S), _ptr(qs), C.byref(costs),
                                              C.byref(F) if F is not None else None,
                                              C.byref(PB) if PB is not None else None,
                                              C.byref(SB) if SB is not None else None))
        if output is not None:
            if qs is not None:
                output.qsim = qs
            output.cost, output.cost_jobs, output.cost_jreg = float(costs.cost), float(costs.cost_jobs), float(costs.cost_jreg)
        return float(costs.cost)


def _tangent_call(s, parameters, parameters_d, parameters_bgd, states, states_d, states_bgd, output, output_d):
    P, k1 = s._pack(parameters, PARAM_NAMES, _lib.Parameters)
    PD, k2 = s._pack(parameters_d, PARAM_NAMES, _lib.Parameters)
    PB, k3 = s._pack(parameters_bgd, PARAM_NAMES, _lib.Parameters)
    S, k4 = s._pack(states, STATE_NAMES, _lib.States)
    SD, k5 = s._pack(states_d, STATE_NAMES, _lib.States)
    SB, k6 = s._pack(states_bgd, STATE_NAMES, _lib.States)
    qs = np.zeros((s.ng, s.nt), np.float32, order="F") if s.ng else None
    qd = np.zeros((s.ng, s.nt), np.float32, order="F") if s.ng else None
    costs = _lib.Costs()
    cost_d = C.c_float(0.0)
    _lib.check(_lib.lib().smashx_forward_d(s._h, C.byref(P), C.byref(PD), C.byref(PB), C.byref(S), C.byref(SD), C.byref(SB),
                                           _ptr(qs), _ptr(qd), C.byref(costs), C.byref(cost_d)))
    if output is not None:
        if qs is not None:
            output.qsim = qs
        output.cost, output.cost_jobs, output.cost_jreg = float(costs.cost), float(costs.cost_jobs), float(costs.cost_jreg)
    if output_d is not None and qd is not None:
        output_d.qsim = qd
    return float(costs.cost), float(cost_d.value)


def invalidate_forcing(input_data):
    """The forcing arrays of input_data were edited in place: the next forward / forward_b / forward_d call uploads them again."""
    s = getattr(input_data, "_smashx_solver", None)
    if s is not None:
        s._fp = None
        s._forcing_external = False


def _solver_for(setup, mesh, input_data, **kw):
    s = getattr(input_data, "_smashx_solver", None)
    if s is not None and getattr(s, "_forcing_external", False):      # the caller placed the forcing in HBM itself (device blocks)
        fp = None
    else:
        fp = Solver.forcing_fingerprint(*((input_data.sparse_prcp, input_data.sparse_pet) if setup.sparse_storage
                                          else (input_data.prcp, input_data.pet)))
    if s is None or s._sig != Solver.signature(setup, mesh):
        s = Solver(setup, mesh, **kw)
        s._fp = None
        input_data._smashx_solver = s
    if fp is not None and getattr(s, "_fp", None) != fp:
        if setup.sparse_storage:
            s.set_forcing(input_data.sparse_prcp, input_data.sparse_pet, sparse=True)
        else:
            s.set_forcing(input_data.prcp, input_data.pet, sparse=False)
        s._fp = fp
    if mesh.ng:
        s.set_qobs(input_data.qobs)
    s.set_options(setup.optimize)
    return s


def forward(setup, mesh, input_data, parameters, parameters_bgd, states, states_bgd, output, cost=None):
    """Drop-in for mw_forward::forward (mw_forward.f90:18-39).  Returns output.cost."""
    s = _solver_for(setup, mesh, input_data)
    s.upload(parameters, states, parameters_bgd, states_bgd)
    key = "sparse_" if setup.sparse_storage else ""
    dom = (getattr(output, key + "qsim_domain", None) if getattr(setup, "save_qsim_domain", False) else None,
           getattr(output, key + "net_prcp_domain", None) if getattr(setup, "save_net_prcp_domain", False) else None)
    if dom[0] is not None or dom[1] is not None:
        s.set_domain_outputs(dom[0], dom[1], setup.sparse_storage)
    try:
        s.sweep(False)
    finally:
        if dom[0] is not None or dom[1] is not None:
            s.set_domain_outputs(None, None)
    return s.download(False, parameters, states, output)


def forward_b(setup, mesh, input_data, parameters, parameters_b, parameters_bgd, parameters_bgd_b, states, states_b,
              states_bgd, states_bgd_b, output, output_b, cost=None, cost_b=1.0):
    """Drop-in for mw_forward::forward_b (mw_forward.f90:41-68): parameters_b / states_b are overwritten
    with the gradient of the cost (times cost_b); *_bgd_b and output_b are scratch in the reference and
    are left untouched here."""
    s = _solver_for(setup, mesh, input_data)
    s.upload(parameters, states, parameters_bgd, states_bgd)
    s.sweep(True, float(cost_b))
    return s.download(True, parameters, states, output, parameters_b, states_b)


def forward_d(setup, mesh, input_data, parameters, parameters_d, parameters_bgd, parameters_bgd_d, states, states_d,
              states_bgd, states_bgd_d, output, output_d, cost=None, cost_d=None):
    """Drop-in for mw_forward::forward_d (mw_forward.f90:70-97), the tangent-linear model: returns (cost, cost_d) and
    fills output.qsim / output_d.qsim.  *_bgd_d are passive in the reference and are ignored."""
    s = _solver_for(setup, mesh, input_data)
    return _tangent_call(s, parameters, parameters_d, parameters_bgd, states, states_d, states_bgd, output, output_d)


# ---- hyper mappings: mw_forward::hyper_forward / hyper_forward_b / hyper_forward_d (mw_forward.f90:99-181) -------------------------
_HYPER = {"hyper-linear": 1, "hyper-polynomial": 2}


def _hyper_map(setup, mesh, input_data, nfields, lb, ub):
    if setup.optimize.mapping not in _HYPER:
        raise _lib.SmashxError(_lib.E_ARG, f"setup.optimize.mapping = {setup.optimize.mapping!r}: hyper-linear or hyper-polynomial expected")
    desc = np.asfortranarray(input_data.descriptor, dtype=np.float32)
    m = _lib.HyperMap(_HYPER[setup.optimize.mapping], mesh.nrow, mesh.ncol, int(desc.shape[2]) if desc.ndim == 3 else 0, nfields,
                      _ptr(desc), None, None)
    keep = (desc, np.ascontiguousarray(lb, np.float32), np.ascontiguousarray(ub, np.float32))
    m.lb, m.ub = _ptr(keep[1]), _ptr(keep[2])
    return m, keep


def _plane_ptrs(fields, names):
    arr = (C.c_void_p * len(names))()
    for i, k in enumerate(names):
        a = getattr(fields, k)
        if not (a.dtype == np.float32 and a.flags.f_contiguous):
            a = np.asfortranarray(a, dtype=np.float32)
            setattr(fields, k, a)
        arr[i] = a.ctypes.data
    return arr


def _hyper_to_fields(setup, mesh, input_data, parameters, hyper_parameters, states, hyper_states, direction=None):
    """hyper_parameters_to_parameters + hyper_states_to_states (mwd_parameters_manipulation.f90:304-362, mwd_states_manipulation.f90:
    270-329) -- or, with direction = (hyper_parameters_d, parameters_d, hyper_states_d, states_d), their tangents (_D)."""
    L, o = _lib.lib(), setup.optimize
    for names, fields, hyp, lb, ub, k in ((PARAM_NAMES, parameters, hyper_parameters, o.lb_parameters, o.ub_parameters, 0),
                                          (STATE_NAMES, states, hyper_states, o.lb_states, o.ub_states, 2)):
        m, keep = _hyper_map(setup, mesh, input_data, len(names), lb, ub)
        h = hyp.matrix()
        if direction is None:
            _lib.check(L.smashx_hyper_map_forward(C.byref(m), _ptr(h), _plane_ptrs(fields, names)))
        else:
            hd = direction[k].matrix()
            _lib.check(L.smashx_hyper_map_d(C.byref(m), _ptr(h), _ptr(hd), _plane_ptrs(fields, names), _plane_ptrs(direction[k + 1], names)))


def _plain(setup):
    """base_hyper_forward knows neither denormalize_forward nor the regularisers (hyper_compute_cost: cost = jobs)."""
    s = setup.copy()
    s.optimize.denormalize_forward = False
    s.optimize.jreg_fun, s.optimize.wjreg_fun, s.optimize.wjreg = [], [], 0.0
    return s


def hyper_forward(setup, mesh, input_data, parameters, hyper_parameters, hyper_parameters_bgd, states, hyper_states, hyper_states_bgd,
                  output, cost=None):
    """Drop-in for mw_forward::hyper_forward (mw_forward.f90:99-123 -> base_hyper_forward, forward.f90:82-157): the descriptor ->
    field maps on the host (include/smashx.h "hyper mappings"), the time loop and the cost on the GPU.  parameters / states come back
    as the mapped fields / the FINAL states (forward.f90:150: no restore).  Returns output.cost."""
    _hyper_to_fields(setup, mesh, input_data, parameters, hyper_parameters, states, hyper_states)
    s = _solver_for(_plain(setup), mesh, input_data)
    s.upload(parameters, states, parameters, states)
    s.sweep(False)
    cost = s.download(False, None, None, output)
    for k in STATE_NAMES:
        getattr(states, k)[...] = getattr(output.fstates, k)
    return cost


def hyper_forward_b(setup, mesh, input_data, parameters, parameters_b, hyper_parameters, hyper_parameters_b, hyper_parameters_bgd,
                    states, states_b, hyper_states, hyper_states_b, hyper_states_bgd, output, output_b, cost=None, cost_b=1.0):
    """Drop-in for mw_forward::hyper_forward_b (mw_forward.f90:125-152 -> BASE_HYPER_FORWARD_B, forward_db.f90:11231-11560):
    hyper_parameters_b / hyper_states_b are overwritten with the gradient of the cost w.r.t. the coefficients (times cost_b);
    parameters_b / states_b hold the gradient w.r.t. the mapped fields."""
    L, o = _lib.lib(), setup.optimize
    _hyper_to_fields(setup, mesh, input_data, parameters, hyper_parameters, states, hyper_states)
    s = _solver_for(_plain(setup), mesh, input_data)
    s.upload(parameters, states, parameters, states)
    s.sweep(True, float(cost_b))
    cost = s.download(True, None, None, output, parameters_b, states_b)
    # (the gradient planes of fields the structure does not read come back as zeros: they add nothing to any sum)
    for names, grads, hyp, hyp_b, lb, ub in ((STATE_NAMES, states_b, hyper_states, hyper_states_b, o.lb_states, o.ub_states),
                                             (PARAM_NAMES, parameters_b, hyper_parameters, hyper_parameters_b, o.lb_parameters, o.ub_parameters)):
        m, keep = _hyper_map(setup, mesh, input_data, len(names), lb, ub)
        ptrs = _plane_ptrs(grads, names)
        hb = np.zeros((o.nhyper, len(names)), np.float32, order="F")
        _lib.check(L.smashx_hyper_map_b(C.byref(m), _ptr(hyp.matrix()), ptrs, _ptr(hb)))
        hyp_b.set_matrix(hb)
    return cost


def hyper_forward_d(setup, mesh, input_data, parameters, parameters_d, hyper_parameters, hyper_parameters_d, hyper_parameters_bgd,
                    states, states_d, hyper_states, hyper_states_d, hyper_states_bgd, output, output_d, cost=None, cost_d=None):
    """Drop-in for mw_forward::hyper_forward_d (mw_forward.f90:154-181 -> BASE_HYPER_FORWARD_D, forward_db.f90:11079-11162).
    Returns (cost, cost_d)."""
    _hyper_to_fields(setup, mesh, input_data, parameters, hyper_parameters, states, hyper_states,
                     direction=(hyper_parameters_d, parameters_d, hyper_states_d, states_d))
    s = _solver_for(_plain(setup), mesh, input_data)
    return _tangent_call(s, parameters, parameters_d, parameters, states, states_d, states, output, output_d)


def scalar_product_test(setup, mesh, input_data, parameters, states, output):
    """mw_adjoint_test::scalar_product_test (mw_adjoint_test.f90:26-105): <dY*, dY> = cost_b * cost_d against
    <dk*, dk> = sum(parameters_b * parameters_d) for dk = 1 on every parameter field and 0 on the states.
    Returns (sp1, sp2)."""
    from .types import OutputDT
    par_d, sta_d = parameters.copy(), states.copy()
    for k in PARAM_NAMES:
        getattr(par_d, k)[...] = 1.0
    for k in STATE_NAMES:
        getattr(sta_d, k)[...] = 0.0
    par_b, sta_b = parameters.copy(), states.copy()
    out_d, out_b = OutputDT(setup, mesh), OutputDT(setup, mesh)
    _, cost_d = forward_d(setup, mesh, input_data, parameters.copy(), par_d, parameters.copy(), parameters.copy(), states.copy(),
                          sta_d, states.copy(), states.copy(), output, out_d)
    forward_b(setup, mesh, input_data, parameters.copy(), par_b, parameters.copy(), parameters.copy(), states.copy(), sta_b,
              states.copy(), states.copy(), output, out_b, 0.0, 1.0)
    sp2 = float(sum(np.sum(getattr(par_b, k).astype(np.float64) * getattr(par_d, k)) for k in PARAM_NAMES))
    return 1.0 * cost_d, sp2


def gradient_test(setup, mesh, input_data, parameters, states, output, nstep=16):
    """mw_adjoint_test::gradient_test (mw_adjoint_test.f90:108-189): Ia = (Y(k + a dk) - Y(k)) / (a dk* . dk) for dk = 1 on
    every parameter field and a = 2^0 .. 2^-(nstep-1); Y = cost of a forward sweep, dk* = parameters_b of one adjoint sweep.
    Returns [(a, |Ia - 1|), ...] (the reference prints them)."""
    from .types import OutputDT
    bgd_p, bgd_s = parameters.copy(), states.copy()
    yk = np.float32(forward(setup, mesh, input_data, parameters.copy(), bgd_p, states.copy(), bgd_s, output))
    par_b, sta_b = parameters.copy(), states.copy()
    forward_b(setup, mesh, input_data, parameters.copy(), par_b, bgd_p, parameters.copy(), states.copy(), sta_b, bgd_s, states.copy(),
              output, OutputDT(setup, mesh), 0.0, 1.0)
    dot = np.float32(0.0)
    for k in PARAM_NAMES:                      # sum(parameters_b_matrix * dk), fp32 like the reference
        dot = np.float32(dot + np.sum(getattr(par_b, k), dtype=np.float32))
    res = []
    for n in range(nstep):
        an = np.float32(2.0 ** (-n))
        p = bgd_p.copy()
        for k in PARAM_NAMES:
            getattr(p, k)[...] = getattr(bgd_p, k) + an
        yadk = np.float32(forward(setup, mesh, input_data, p, bgd_p, states.copy(), bgd_s, output))
        ian = np.float32((yadk - yk) / (an * dot))
        res.append((float(an), float(abs(ian - np.float32(1.0)))))
    return res
