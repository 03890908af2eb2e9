"""ctypes view of include/smashx.h.  Loading fails loudly when libsmashx.so is missing: there is no
Python / CPU implementation of the solver behind this module."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMASHX_EXACT_LIBM=1 selects the exact-libm build (glibc's float functions restated, IEEE divisions: csrc/sx_libm.h) -- the
# slower library that reproduces the reference bit for bit; SMASHX_LIB overrides the path for A/B experiments.
EXACT = os.environ.get("SMASHX_EXACT_LIBM", "0") not in ("", "0")
LIB_PATH = os.environ.get("SMASHX_LIB", os.path.join(_HERE, "libsmashx_exact.so" if EXACT else "libsmashx.so"))

GNP, GNS = 16, 8

E_OK, E_ARG, E_UNSUPPORTED, E_HIP, E_NODEVICE, E_MESH, E_STATE = 0, -1, -2, -3, -4, -5, -6

SYMBOLS = [
    "smashx_last_error", "smashx_abi_sizes", "smashx_device_count", "smashx_plan_create", "smashx_plan_destroy", "smashx_plan_ncells",
    "smashx_plan_cell_order", "smashx_set_forcing", "smashx_set_forcing_device_block", "smashx_set_qobs",
    "smashx_set_options", "smashx_forward", "smashx_forward_b", "smashx_upload", "smashx_sweep", "smashx_download",
    "smashx_get_timing", "smashx_halo_counts", "smashx_halo_edges", "smashx_plan_chunking", "smashx_set_halo", "smashx_tile_probe", "smashx_debug_group_times", "smashx_set_domain_outputs", "smashx_forward_d", "smashx_tangent_terms", "smashx_selftest_math",
    "smashx_set_forcing_layout", "smashx_forcing_info", "smashx_control_size", "smashx_control_set", "smashx_control_get",
    "smashx_control_gradient",
    "smashx_comm_unique_id", "smashx_comm_create", "smashx_comm_destroy", "smashx_comm_allreduce_sum", "smashx_set_exchange",
    "smashx_set_median_slots", "smashx_selftest_paths",
    "smashx_lbfgsb_create", "smashx_lbfgsb_step", "smashx_lbfgsb_destroy", "smashx_lbfgsb_iterations", "smashx_lbfgsb_message",
    "smashx_lbfgsb_evaluations", "smashx_lbfgsb_projected_gradient", "smashx_plan_hbm", "smashx_comm_info",
    "smashx_hyper_nhyper", "smashx_hyper_map_forward", "smashx_hyper_map_d", "smashx_hyper_map_b",
]

HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int)


class Config(C.Structure):
    _fields_ = [("structure", C.c_int), ("nrow", C.c_int), ("ncol", C.c_int), ("nt", C.c_int), ("ng", C.c_int),
                ("dt", C.c_float), ("dx", C.c_float), ("chunk_steps", C.c_int), ("pipe_steps", C.c_int), ("group_size", C.c_int),
                ("device", C.c_int), ("tile", C.c_int * 4)]


class Mesh(C.Structure):
    _fields_ = [("flwdir", C.c_void_p), ("flwacc", C.c_void_p), ("active_cell", C.c_void_p), ("path", C.c_void_p),
                ("gauge_pos", C.c_void_p), ("area", C.c_void_p), ("owner_mask", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("denormalize_forward", C.c_int), ("optimize_start_step", C.c_int), ("njf", C.c_int),
                ("jobs_fun", C.c_int * 8), ("wjobs_fun", C.c_float * 8), ("njr", C.c_int), ("jreg_fun", C.c_int * 4),
                ("wjreg_fun", C.c_float * 4), ("wjreg", C.c_float), ("optim_parameters", C.c_int * GNP),
                ("optim_states", C.c_int * GNS), ("lb_parameters", C.c_float * GNP), ("ub_parameters", C.c_float * GNP),
                ("lb_states", C.c_float * GNS), ("ub_states", C.c_float * GNS), ("wgauge", C.c_void_p)]


class ForcingLayout(C.Structure):
    _fields_ = [("compact", C.c_int), ("prcp_factor", C.c_float), ("pet_ratio", C.c_float * 24), ("pet_hour0", C.c_int)]


class HyperMap(C.Structure):
    _fields_ = [("mapping", C.c_int), ("nrow", C.c_int), ("ncol", C.c_int), ("nd", C.c_int), ("nfields", C.c_int),
                ("descriptor", C.c_void_p), ("lb", C.c_void_p), ("ub", C.c_void_p)]


class Parameters(C.Structure):
    _fields_ = [("f", C.c_void_p * GNP)]


class States(C.Structure):
    _fields_ = [("f", C.c_void_p * GNS)]


class Costs(C.Structure):
    _fields_ = [("cost", C.c_float), ("cost_jobs", C.c_float), ("cost_jreg", C.c_float)]


class Timing(C.Structure):
    _fields_ = [("sweep_ms", C.c_float), ("vert_fwd_ms", C.c_float), ("route_fwd_ms", C.c_float), ("cost_ms", C.c_float),
                ("route_adj_ms", C.c_float), ("vert_adj_ms", C.c_float), ("vert_fwd_launches", C.c_int),
                ("route_fwd_launches", C.c_int), ("route_adj_launches", C.c_int), ("vert_adj_launches", C.c_int),
                ("n_chunks", C.c_int), ("chunk_steps", C.c_int), ("pipe_steps", C.c_int), ("n_rounds", C.c_int), ("n_groups", C.c_int),
                ("device_bytes", C.c_double), ("cellsteps", C.c_double * 4),
                ("route_fwd_chained_ms", C.c_float), ("route_adj_chained_ms", C.c_float),
                ("route_fwd_chained_launches", C.c_int), ("route_adj_chained_launches", C.c_int), ("max_stage", C.c_int),
                ("n_chained_groups", C.c_int), ("chain_staged", C.c_int)]


class SmashxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsmashx error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """The C-ABI library.  Raises if the HIP extension has not been built (python -c 'import
    __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build the HIP library first (__graft_entry__.build()); "
                              "smash_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.smashx_last_error.restype = C.c_char_p
        for s in SYMBOLS[1:]:
            getattr(L, s).restype = C.c_int
        L.smashx_lbfgsb_message.restype = C.c_char_p
        L.smashx_lbfgsb_iterations.restype = C.c_long
        L.smashx_lbfgsb_evaluations.restype = C.c_long
        L.smashx_lbfgsb_projected_gradient.restype = C.c_double
        L.smashx_lbfgsb_create.argtypes = [C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
        L.smashx_lbfgsb_step.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.POINTER(C.c_int)]
        for s in ("smashx_lbfgsb_destroy", "smashx_lbfgsb_iterations", "smashx_lbfgsb_message", "smashx_lbfgsb_evaluations", "smashx_lbfgsb_projected_gradient"):
            getattr(L, s).argtypes = [C.c_void_p]
        # the structs above mirror include/smashx.h by hand: refuse a library built from another layout (a stale .so would have
        # smashx_get_timing write past the end of Timing)
        sizes = (C.c_int * 7)()
        L.smashx_abi_sizes(sizes)
        mine = [C.sizeof(t) for t in (Config, Mesh, Options, Parameters, States, Costs, Timing)]
        if list(sizes) != mine:
            raise ImportError(f"{LIB_PATH} was built from another include/smashx.h (struct sizes {list(sizes)}, this module expects {mine}): "
                              "rebuild it (__graft_entry__.build())")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SmashxError(rc, lib().smashx_last_error().decode())
