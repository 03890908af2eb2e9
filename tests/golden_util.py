"""Load tests/golden/*.npz fixtures (made by tests/golden/make_golden.py from the reference solver)."""
import glob
import hashlib
import os
import types

import numpy as np

from smash_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = types.SimpleNamespace()
    g.name = name
    g.structure = str(z["structure"])
    g.dt = float(z["dt"])
    g.nt = int(z["nt"])
    g.mesh = synth.Mesh(int(z["nrow"]), int(z["ncol"]), float(z["dx"]), z["flwdir"], z["flwacc"], z["path"],
                        z["active_cell"], z["gauge_pos"], z["area"])
    if int(z["big"]):
        g.prcp, g.pet = synth.dense_forcing(g.mesh, g.nt, gap_per_million=int(z["gaps"]))
        assert _sha(g.prcp) == str(z["prcp_sha"]) and _sha(g.pet) == str(z["pet_sha"]), "synthetic forcing drifted"
    else:
        g.prcp, g.pet = np.asfortranarray(z["prcp"]), np.asfortranarray(z["pet"])
    g.qobs = np.asfortranarray(z["qobs"])
    g.params = {k: np.asfortranarray(z["p_" + k]) for k in synth.PARAM_NAMES}
    g.states = {k: np.asfortranarray(z["s_" + k]) for k in synth.STATE_NAMES}
    opts = {}
    for key in z.files:
        if key.startswith("opt_"):
            v = z[key]
            k = key[4:]
            if k in ("jobs_fun", "jreg_fun"):
                opts[k] = tuple(str(s) for s in v)
            elif v.ndim == 0:
                opts[k] = v.item()
            else:
                opts[k] = v
    if "pbgd_cp" in z.files:
        opts["params_bgd"] = {k: np.asfortranarray(z["pbgd_" + k]) for k in synth.PARAM_NAMES}
        opts["states_bgd"] = {k: np.asfortranarray(z["sbgd_" + k]) for k in synth.STATE_NAMES}
    g.opts = opts
    g.fwd = dict(qsim=z["fwd_qsim"], cost=float(z["fwd_cost"]), cost_jobs=float(z["fwd_cost_jobs"]),
                 cost_jreg=float(z["fwd_cost_jreg"]),
                 fstates={k: z["fwd_fstates_" + k] for k in synth.STATE_NAMES},
                 states={k: z["fwd_states_out_" + k] for k in synth.STATE_NAMES},
                 parameters={k: z["fwd_parameters_out_" + k] for k in synth.PARAM_NAMES})
    g.adj = dict(qsim=z["adj_qsim"], cost=float(z["adj_cost"]),
                 parameters_b={k: z["adj_parameters_b_" + k] for k in synth.PARAM_NAMES},
                 states_b={k: z["adj_states_b_" + k] for k in synth.STATE_NAMES})
    return g


STRUCT_PARAMS = {"gr-a": ("cp", "cft", "exc", "lr"), "gr-b": ("ci", "cp", "cft", "exc", "lr"),
                 "gr-c": ("ci", "cp", "cft", "cst", "exc", "lr"), "gr-d": ("cp", "cft", "lr")}
STRUCT_STATES = {"gr-a": ("hp", "hft", "hlr"), "gr-b": ("hi", "hp", "hft", "hlr"),
                 "gr-c": ("hi", "hp", "hft", "hst", "hlr"), "gr-d": ("hp", "hft", "hlr")}


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    n = np.linalg.norm(b)
    d = np.linalg.norm(a - b)
    return d / n if n > 0 else d
