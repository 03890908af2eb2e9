"""CPU: the hyper-linear / hyper-polynomial maps of the library's host side (smash_amd/csrc/sx_hyper.cpp, include/smashx.h "hyper
mappings") against the reference's own fixtures (tests/golden/hyper/*.npz: mw_forward::hyper_forward / hyper_forward_b of the all-CPU
reference, tests/golden/make_golden.py).  No GPU: the maps are host code; the gradient planes they are fed come from the oracle."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

import golden_util as gu  # noqa: E402
import make_golden as mg  # noqa: E402
from oracle import pyoracle  # noqa: E402


def _case(name, mapping):
    import smash_amd
    g = gu.load(name)
    z = np.load(os.path.join(gu.GOLDEN_DIR, "hyper", f"{name}__{mapping}.npz"))
    desc, hp, hs = mg.hyper_inputs(g, mapping)
    setup = smash_amd.SetupDT(desc.shape[2], g.mesh.ng, structure=g.structure, dt=g.dt, ntime_step=g.nt)
    o = setup.optimize
    o.mapping, o.nhyper = mapping, len(next(iter(hp.values())))
    o.jobs_fun, o.wjobs_fun = list(g.opts.get("jobs_fun", ("nse",))), list(g.opts.get("wjobs_fun", (1.0,)))
    o.optimize_start_step = int(g.opts.get("optimize_start_step", 1))
    if "wgauge" in g.opts:
        o.wgauge = np.asarray(g.opts["wgauge"], np.float32)
    mesh = smash_amd.MeshDT.from_synth(setup, g.mesh)
    inp = smash_amd.Input_DataDT(setup, mesh)
    inp.prcp, inp.pet, inp.qobs, inp.descriptor = g.prcp, g.pet, g.qobs, desc
    par, sta = smash_amd.ParametersDT.from_dict(mesh, g.params), smash_amd.StatesDT.from_dict(mesh, g.states)
    HP, HS = smash_amd.Hyper_ParametersDT.from_dict(setup, hp), smash_amd.Hyper_StatesDT.from_dict(setup, hs)
    return g, z, setup, mesh, inp, par, sta, HP, HS


@pytest.mark.parametrize("name,mapping", mg.HYPER_CASES)
def test_mapped_fields_are_bit_identical_to_the_reference(name, mapping):
    """hyper_parameters_to_parameters / hyper_states_to_states (mwd_parameters_manipulation.f90:304-362): the sixteen mapped
    parameter planes of the reference, bit for bit -- same operation order, powf / expf of the same C library."""
    from smash_amd import synth
    from smash_amd.solver import _hyper_to_fields
    g, z, setup, mesh, inp, par, sta, HP, HS = _case(name, mapping)
    _hyper_to_fields(setup, mesh, inp, par, HP, sta, HS)
    for k in synth.PARAM_NAMES:
        assert np.array_equal(getattr(par, k), z["fwd_p_" + k]), k


@pytest.mark.parametrize("name,mapping", mg.HYPER_CASES)
def test_adjoint_and_tangent_of_the_maps(name, mapping):
    """HYPER_*_TO_*_B (forward_db.f90:1434-1537, 2272-2369) fed with the oracle's gradient planes of the mapped fields (bit-identical to
    the reference's) gives the reference's hyper gradients (whole-grid fp32 sums: the reference's own noise bar, floor 5e-6); HYPER_*_D
    (:1313-1403, 2179-2256) is its transpose: <map_b(g), h_d> = <g, map_d(h_d)>."""
    import ctypes as C
    from smash_amd import _lib, synth
    from smash_amd.solver import _hyper_map, _hyper_to_fields, _plane_ptrs, _ptr
    g, z, setup, mesh, inp, par, sta, HP, HS = _case(name, mapping)
    _hyper_to_fields(setup, mesh, inp, par, HP, sta, HS)
    r = pyoracle.run(g.structure, g.mesh, g.dt, g.prcp, g.pet, g.qobs, par.as_dict(), sta.as_dict(), adjoint=True,
                     **{k: v for k, v in g.opts.items() if k in ("jobs_fun", "wjobs_fun", "optimize_start_step", "wgauge")})
    o, L = setup.optimize, _lib.lib()
    import smash_amd
    for names, grads, hyp, lb, ub, key, nkey in ((synth.PARAM_NAMES, r["parameters_b"], HP, o.lb_parameters, o.ub_parameters, "adj_hp_b_", "noise_hp_b_"),
                                                 (synth.STATE_NAMES, r["states_b"], HS, o.lb_states, o.ub_states, "adj_hs_b_", "noise_hs_b_")):
        cls = smash_amd.ParametersDT if names is synth.PARAM_NAMES else smash_amd.StatesDT
        G = cls.from_dict(mesh, {k: grads.get(k, np.zeros((mesh.nrow, mesh.ncol), np.float32)) for k in names})
        m, keep = _hyper_map(setup, mesh, inp, len(names), lb, ub)
        hb = np.zeros((o.nhyper, len(names)), np.float32, order="F")
        _lib.check(L.smashx_hyper_map_b(C.byref(m), _ptr(hyp.matrix()), _plane_ptrs(G, names), _ptr(hb)))
        used = gu.STRUCT_PARAMS[g.structure] if names is synth.PARAM_NAMES else gu.STRUCT_STATES[g.structure]
        for i, k in enumerate(names):
            if k in used:
                assert gu.rel_l2(hb[:, i], z[key + k]) <= gu.tol(float(z[nkey + k]), base=5e-6), (k, hb[:, i], z[key + k])
            else:
                assert not hb[:, i].any(), k
        # transpose property on a random direction
        rng = np.random.default_rng(7)
        hd = np.asfortranarray(rng.normal(size=hb.shape).astype(np.float32) * 0.01)
        V, VD = cls(mesh), cls(mesh)
        _lib.check(L.smashx_hyper_map_d(C.byref(m), _ptr(hyp.matrix()), _ptr(hd), _plane_ptrs(V, names), _plane_ptrs(VD, names)))
        lhs = float(np.sum(hb.astype(np.float64) * hd))
        rhs = float(sum(np.sum(getattr(G, k).astype(np.float64) * getattr(VD, k)) for k in names))
        assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs), 1e-30), (lhs, rhs)
        # and the tangent's value is the mapped field, up to the re-association of the _D form (lb + w / (e + 1) instead of
        # w * (1 / (1 + e)) + lb: one ulp of the bounds' magnitude, visible where a field sits near zero between -50 and 50)
        for i, k in enumerate(names):
            assert np.allclose(getattr(V, k), getattr(par if names is synth.PARAM_NAMES else sta, k), rtol=3e-7,
                               atol=2.5e-7 * max(abs(float(lb[i])), abs(float(ub[i])))), k


def test_bad_arguments_are_refused():
    import ctypes as C
    from smash_amd import _lib
    L = _lib.lib()
    m = _lib.HyperMap(3, 4, 4, 1, 16, None, None, None)
    assert L.smashx_hyper_nhyper(C.byref(m)) == _lib.E_ARG
    assert L.smashx_hyper_map_forward(C.byref(m), None, None) == _lib.E_ARG
