"""GPU: lossless compact residency of the forcing (include/smashx.h smashx_set_forcing_layout).  The reference's reader forms
prcp as real(k) * prcp_conversion_factor and -- with daily_interannual_pet -- pet as daily * RATIO_PET_HOURLY(hour)
(smash/core/_read_input_data.py:176-196, 223-283); the plan then keeps uint16 counts + the daily field and the kernels rebuild
the fp32 values with the same single multiply.  Bar: results BIT-IDENTICAL to the fp32 layout (the decode reproduces the
inputs exactly, everything downstream is the same arithmetic); data that is not of the form falls back, never approximated."""
import numpy as np
import pytest

import golden_util as gu
from smash_amd import synth

pytestmark = pytest.mark.gpu

SYNTH_LAYOUT = dict(compact=True, prcp_factor=0.1, pet_ratio=synth._pet_tables()[1], pet_hour0=0)


def _same(a, b, g):
    assert np.array_equal(a[2].qsim, b[2].qsim)
    assert a[2].cost == b[2].cost
    for k in gu.STRUCT_PARAMS[g.structure]:
        assert np.array_equal(getattr(a[3], k), getattr(b[3], k)), k
    for k in gu.STRUCT_STATES[g.structure]:
        assert np.array_equal(getattr(a[4], k), getattr(b[4], k)), k


@pytest.mark.parametrize("name", [n for n in gu.names() if "cance" not in n])
def test_compact_forcing_is_bit_identical(name):
    """Every synthetic fixture (gr-a..d, vic-a; gaps, D8, chunked + lean-tape variants) with the compact layout forced on."""
    from test_gpu_parity import _run_adjoint, _types
    g = gu.load(name)
    ref = _run_adjoint(g, chunk_steps=0)
    setup, mesh, inp, par, sta, out = _types(g, layout=dict(SYNTH_LAYOUT))
    info = inp._smashx_solver.forcing_info()
    assert info["layout"].startswith("compact") and info["resident_bytes_per_cellstep"] < 2.5, info
    _same(ref, _run_adjoint(g, layout=dict(SYNTH_LAYOUT)), g)
    _same(ref, _run_adjoint(g, layout=dict(SYNTH_LAYOUT), chunk_steps=32, pipe_steps=16), g)


def test_compact_tangent_is_bit_identical():
    import smash_amd
    from test_gpu_parity import _types
    g = gu.load("gr_c_48x48x480_nse")
    res = []
    for layout in (None, dict(SYNTH_LAYOUT)):
        setup, mesh, inp, par, sta, out = _types(g, chunk_steps=0, **({"layout": layout} if layout else {}))
        par_d, sta_d = par.copy(), sta.copy()
        for k in synth.PARAM_NAMES:
            getattr(par_d, k)[...] = 1.0
        for k in synth.STATE_NAMES:
            getattr(sta_d, k)[...] = 0.5
        out_d = smash_amd.OutputDT(setup, mesh)
        c, cd = smash_amd.forward_d(setup, mesh, inp, par, par_d, inp._bgd[0], par.copy(), sta, sta_d, inp._bgd[1], sta.copy(), out, out_d)
        res.append((c, cd, out.qsim.copy(), out_d.qsim.copy()))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


def test_real_data_that_is_not_of_the_form_falls_back_to_fp32():
    """The Cance fixture: its PET is daily x RATIO_PET_HOURLY (first step 01:00 -> hour0 = 1), but the fixture's rainfall was formed
    in double precision (tests/golden/cance_io.py) and is not real(k) * 0.1 in fp32: the plan must notice and keep fp32 rows."""
    from test_gpu_parity import _run_adjoint, _types
    g = gu.load("gr_a_cance_28x28x1440")
    lay = dict(compact=True, prcp_factor=0.1, pet_ratio=None, pet_hour0=1)
    setup, mesh, inp, par, sta, out = _types(g, layout=dict(lay))
    assert inp._smashx_solver.forcing_info()["layout"] == "fp32 rows"
    _same(_run_adjoint(g, chunk_steps=0), _run_adjoint(g, layout=dict(lay)), g)
    # with both fields put on the fp32 reader's form -- real(k) * 0.1 and daily * ratio(hour) as float32 products, what the reference's
    # reader produces from Float32 / integer rasters (the fixture went through float64) -- the case compacts
    from smash_amd.solver import RATIO_PET_HOURLY as R
    g.prcp = np.asfortranarray(np.where(g.prcp < 0, g.prcp, np.rint(g.prcp / np.float32(0.1)).astype(np.float32) * np.float32(0.1)).astype(np.float32))
    pet = g.pet.copy(order="F")
    for d in range((g.nt + 1 + 23) // 24):
        ts = list(range(max(0, d * 24 - 1), min(g.nt, (d + 1) * 24 - 1)))
        tb = max(ts, key=lambda t: R[(t + 1) % 24])
        daily = (pet[:, :, tb] / R[(tb + 1) % 24]).astype(np.float32) if R[(tb + 1) % 24] > 0 else np.zeros(pet.shape[:2], np.float32)
        for t in ts:
            pet[:, :, t] = daily * R[(t + 1) % 24]
    g.pet = pet
    setup, mesh, inp, par, sta, out = _types(g, layout=dict(lay))
    assert inp._smashx_solver.forcing_info()["layout"].startswith("compact")
    _same(_run_adjoint(g, chunk_steps=0), _run_adjoint(g, layout=dict(lay)), g)


def test_gap_days_and_unaligned_device_blocks():
    """A day without a PET file is -99 at every hour (_read_input_data.py:246-251), a missing rain file -99 for the step; the
    device-block entry point takes blocks that do not end on day boundaries; a block that breaks the form is refused."""
    import torch
    import smash_amd
    from smash_amd import _lib
    from smash_amd.solver import Solver
    from test_gpu_parity import _run_adjoint, _types
    g = gu.load("gr_b_64x64x720_nse")
    g.pet = g.pet.copy(order="F"); g.prcp = g.prcp.copy(order="F")
    g.pet[:, :, 48:72] = -99.0                        # a gap day everywhere
    g.pet[3:9, 5:11, 240:264] = -99.0                 # and one over a few cells only
    g.prcp[:, :, 100] = -99.0
    ref = _run_adjoint(g, chunk_steps=0)
    _same(ref, _run_adjoint(g, layout=dict(SYNTH_LAYOUT)), g)
    # device blocks of 50 steps (not whole days), cells in plan order
    setup, mesh, inp, par, sta, out = _types(g)
    sol = Solver(setup, mesh)
    sol.set_forcing_layout(**SYNTH_LAYOUT)
    rows, cols = sol.cell_order()
    sol.upload(par, sta)
    for t0 in range(0, g.nt, 50):
        t1 = min(g.nt, t0 + 50)
        bp = torch.from_numpy(np.ascontiguousarray(g.prcp[rows, cols, t0:t1].T)).cuda()
        be = torch.from_numpy(np.ascontiguousarray(g.pet[rows, cols, t0:t1].T)).cuda()
        torch.cuda.synchronize()
        if t0 > 0:
            # until device blocks have covered EVERY step the forcing does not count as set: a sweep now would close the daily PET of
            # the days still to come
            with pytest.raises(smash_amd.SmashxError) as e:
                sol.sweep(False)
            assert e.value.code == _lib.E_STATE
        sol.set_forcing_device_block(t0, t1, bp.data_ptr(), be.data_ptr())
    assert sol.forcing_info()["layout"].startswith("compact")
    inp._smashx_solver = sol
    pb, sb = par.copy(), sta.copy()
    smash_amd.forward_b(setup, mesh, inp, par, pb, par.copy(), par.copy(), sta, sb, sta.copy(), sta.copy(), out, out.copy(), np.float32(0), np.float32(1))
    _same(ref, (par, sta, out, pb, sb), g)
    bad = torch.from_numpy(np.ascontiguousarray(g.prcp[rows, cols, 0:24].T) + np.float32(0.013)).cuda()
    be = torch.from_numpy(np.ascontiguousarray(g.pet[rows, cols, 0:24].T)).cuda()
    torch.cuda.synchronize()
    with pytest.raises(smash_amd.SmashxError) as e:
        sol.set_forcing_device_block(0, 24, bad.data_ptr(), be.data_ptr())
    assert e.value.code == _lib.E_UNSUPPORTED


def test_partial_gap_day_across_blocks_is_refused():
    """A day whose night hours arrive as +0 in one device block and whose daytime hours arrive as -99 in the next is not of the
    reader's form (a gap day is -99 at every hour): the decode would turn the night hours into gaps.  The second block must be
    refused even though the first one, on its own, was acceptable."""
    import torch
    import smash_amd
    from smash_amd import _lib
    from smash_amd.solver import Solver
    from test_gpu_parity import _types
    g = gu.load("gr_b_16x16x96_nse_gaps")
    pet = g.pet.copy(order="F")
    pet[:, :, 24:30] = 0.0                          # hours 0..5 of day 1 (night: ratio 0)
    pet[:, :, 30:48] = -99.0                        # the rest of that day marked as a gap
    setup, mesh, inp, par, sta, out = _types(g)
    sol = Solver(setup, mesh)
    sol.set_forcing_layout(**SYNTH_LAYOUT)
    rows, cols = sol.cell_order()
    codes = []
    for t0 in range(0, 48, 6):
        bp = torch.from_numpy(np.ascontiguousarray(g.prcp[rows, cols, t0:t0 + 6].T)).cuda()
        be = torch.from_numpy(np.ascontiguousarray(pet[rows, cols, t0:t0 + 6].T)).cuda()
        torch.cuda.synchronize()
        try:
            sol.set_forcing_device_block(t0, t0 + 6, bp.data_ptr(), be.data_ptr())
            codes.append(0)
        except smash_amd.SmashxError as e:
            codes.append(e.code)
            break
    assert codes[:5] == [0, 0, 0, 0, 0] and codes[-1] == _lib.E_UNSUPPORTED and len(codes) == 6, codes


def test_fp32_device_blocks_in_any_order_and_refreshed():
    """fp32 rows (no compact layout): device blocks may arrive in any order -- [k, nt) before [0, k) -- and a block of an already
    complete forcing may be refreshed; neither forgets the coverage (include/smashx.h, smashx_set_forcing_device_block)."""
    import torch
    import smash_amd
    from smash_amd.solver import Solver
    from test_gpu_parity import _run_adjoint, _types
    g = gu.load("gr_b_16x16x96_nse_gaps")
    ref = _run_adjoint(g, chunk_steps=0)
    setup, mesh, inp, par, sta, out = _types(g)
    sol = Solver(setup, mesh)
    rows, cols = sol.cell_order()

    def send(t0, t1):
        bp = torch.from_numpy(np.ascontiguousarray(g.prcp[rows, cols, t0:t1].T)).cuda()
        be = torch.from_numpy(np.ascontiguousarray(g.pet[rows, cols, t0:t1].T)).cuda()
        torch.cuda.synchronize()
        sol.set_forcing_device_block(t0, t1, bp.data_ptr(), be.data_ptr())
    send(40, g.nt)
    send(0, 40)                                      # the block that starts at step 0 comes last
    send(0, 24)                                      # ... and its head is refreshed once more
    assert not sol.forcing_info()["layout"].startswith("compact")
    inp._smashx_solver = sol
    pb, sb = par.copy(), sta.copy()
    smash_amd.forward_b(setup, mesh, inp, par, pb, par.copy(), par.copy(), sta, sb, sta.copy(), sta.copy(), out, out.copy(), np.float32(0), np.float32(1))
    _same(ref, (par, sta, out, pb, sb), g)
