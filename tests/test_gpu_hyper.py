"""GPU: smash_amd.hyper_forward / hyper_forward_b / hyper_forward_d -- the Python host's mirror of mw_forward::hyper_forward(_b, _d)
(mw_forward.f90:99-181): descriptor -> field maps on the host (include/smashx.h "hyper mappings"), time loop, cost and their adjoint /
tangent on the GPU -- against the golden vectors of the all-CPU reference (tests/golden/hyper/*.npz).  The same fixtures and bars as
tests/test_gpu_dropin.py uses for the Fortran shim, where the reference's own host routines do the maps."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import golden_util as gu  # noqa: E402
import make_golden as mg  # noqa: E402
from test_hyper_cpu import _case  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,mapping", mg.HYPER_CASES)
def test_hyper_forward_and_adjoint_vs_reference_golden(name, mapping):
    import smash_amd
    from smash_amd import synth
    g, z, setup, mesh, inp, par, sta, HP, HS = _case(name, mapping)
    out = smash_amd.OutputDT(setup, mesh)
    cost = smash_amd.hyper_forward(setup, mesh, inp, par, HP, HP.copy(), sta, HS, HS.copy(), out, np.float32(0))
    for i in range(g.mesh.ng):
        assert gu.rel_l2(out.qsim[i], z["fwd_qsim"][i]) <= gu.tol(z["noise_qsim"][i]), i
    assert abs(cost - float(z["fwd_cost"])) <= gu.tol_cost(float(z["noise_cost"]), float(z["fwd_cost"]))
    for k in synth.PARAM_NAMES:                      # the mapped fields: identical to the reference's
        assert np.array_equal(getattr(par, k), z["fwd_p_" + k]), k
    for k in gu.STRUCT_STATES[g.structure]:           # base_hyper_forward leaves the states at their final values
        assert gu.rel_l2(getattr(sta, k), z["fwd_s_" + k]) <= gu.tol_fstate(k, 1e-5), k
    # adjoint
    g, z, setup, mesh, inp, par, sta, HP, HS = _case(name, mapping)
    out = smash_amd.OutputDT(setup, mesh)
    par_b, sta_b, HPb, HSb = par.copy(), sta.copy(), HP.copy(), HS.copy()
    cost = smash_amd.hyper_forward_b(setup, mesh, inp, par, par_b, HP, HPb, HP.copy(), sta, sta_b, HS, HSb, HS.copy(), out, out.copy(),
                                     np.float32(0), np.float32(1))
    assert abs(cost - float(z["adj_cost"])) <= gu.tol_cost(float(z["noise_cost"]), float(z["adj_cost"]))
    for k in gu.STRUCT_PARAMS[g.structure]:
        assert gu.rel_l2(getattr(HPb, k).reshape(-1), z["adj_hp_b_" + k]) <= gu.tol(float(z["noise_hp_b_" + k]), base=5e-6), k
    for k in gu.STRUCT_STATES[g.structure]:
        assert gu.rel_l2(getattr(HSb, k).reshape(-1), z["adj_hs_b_" + k]) <= gu.tol(float(z["noise_hs_b_" + k]), base=5e-6), k


@pytest.mark.parametrize("name,mapping", mg.HYPER_CASES)
def test_hyper_forward_d_vs_reference_golden(name, mapping):
    import smash_amd
    g, z, setup, mesh, inp, par, sta, HP, HS = _case(name, mapping)
    desc, hp, hs = mg.hyper_inputs(g, mapping)
    adj = dict(hyper_parameters_b={k: z["adj_hp_b_" + k] for k in hp}, hyper_states_b={k: z["adj_hs_b_" + k] for k in hs})
    hd, sd = mg.hyper_direction(adj)
    HPd, HSd = smash_amd.Hyper_ParametersDT.from_dict(setup, hd), smash_amd.Hyper_StatesDT.from_dict(setup, sd)
    out, out_d = smash_amd.OutputDT(setup, mesh), smash_amd.OutputDT(setup, mesh)
    cost, cost_d = smash_amd.hyper_forward_d(setup, mesh, inp, par, par.copy(), HP, HPd, HP.copy(), sta, sta.copy(), HS, HSd, HS.copy(),
                                             out, out_d)
    for i in range(g.mesh.ng):
        assert gu.rel_l2(out_d.qsim[i], z["tan_qsim_d"][i]) <= gu.tol(z["noise_tan_qsim_d"][i]), i
    ref = float(z["tan_cost_d"])
    assert abs(cost_d - ref) <= max(1e-5, 3 * float(z["noise_tan_cost_d"])) * abs(ref), (cost_d, ref)
    dot = sum(float(np.dot(z["adj_hp_b_" + k].astype(np.float64), hd[k])) for k in hp) + \
        sum(float(np.dot(z["adj_hs_b_" + k].astype(np.float64), sd[k])) for k in hs)
    assert abs(cost_d - dot) <= 2e-4 * abs(dot), (cost_d, dot)
