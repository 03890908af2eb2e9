"""GPU: the multi-GPU tile path on ONE card.  Each tile of the grid gets its own plan (as each rank would) and
runs its forward + adjoint sweep in its own thread; the halo callback moves the boundary series between the
plans through in-process queues instead of RCCL.  Everything else -- per-tile schedules, inlet / outlet
series, pack / unpack kernels, the four exchange points of the sweep -- is exactly what bench.py --gpus N runs.
Results must be BIT-IDENTICAL to the single-domain run: the decomposition only changes who computes a cell."""
import queue
import threading

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


class Loopback:
    def __init__(self, rank, solver, nrow, ncol, pr, pc, box, owner=None):
        import torch
        from smash_amd import tiles
        self.torch, self.rank, self.box = torch, rank, box
        self.peers = tiles.PeerLists(solver, nrow, ncol, pr, pc, owner)
        _, self.tp = solver.chunking()
        self.out_buf = torch.zeros(max(self.peers.n_out, 1) * self.tp, dtype=torch.float32, device="cuda")
        self.in_buf = torch.zeros(max(self.peers.n_in, 1) * self.tp, dtype=torch.float32, device="cuda")
        self.calls = 0
        solver.set_halo(self.out_buf.data_ptr(), self.in_buf.data_ptr(), self)

    def __call__(self, phase, t0, nsteps):
        torch = self.torch
        self.calls += 1
        w = 4 * ((nsteps + 3) // 4)
        use_out = phase in (1, 2)
        buf = self.out_buf if use_out else self.in_buf
        n = self.peers.n_out if use_out else self.peers.n_in
        idx = self.peers.out_peers if use_out else self.peers.in_peers
        view = buf[: n * w].view(n, w)
        kind = "f" if phase < 2 else "a"
        for p, ix in idx.items():
            ixt = torch.from_numpy(ix).cuda()
            if phase in (1, 3):
                self.box[(self.rank, p, kind)].put((t0, view[ixt].clone()))
            else:
                t0_, data = self.box[(p, self.rank, kind)].get(timeout=120)
                assert t0_ == t0 and data.shape == (len(ix), w)
                view[ixt] = data
        torch.cuda.current_stream().synchronize()
        return 0


def _tile_inputs(g, rect, ng_total, mine=None):
    import smash_amd
    gp = np.asarray(g.mesh.gauge_pos)
    if mine is not None:                          # arbitrary partition: mine[row, col] = this rank owns the cell
        loc = [i for i in range(g.mesh.ng) if mine[gp[i, 0], gp[i, 1]]]
    else:
        r0, r1, c0, c1 = rect
        loc = [i for i in range(g.mesh.ng) if r0 <= gp[i, 0] < r1 and c0 <= gp[i, 1] < c1]
    setup = smash_amd.SetupDT(0, len(loc), structure=g.structure, dt=g.dt, ntime_step=g.nt)
    setup.optimize.jobs_fun, setup.optimize.wjobs_fun = ["nse"], [1.0]
    setup.optimize.wgauge = np.full(len(loc), 1.0 / ng_total, np.float32)
    mesh = smash_amd.MeshDT(setup, g.mesh.nrow, g.mesh.ncol, len(loc))
    mesh.dx, mesh.flwdir, mesh.flwacc, mesh.path, mesh.active_cell = g.mesh.dx, g.mesh.flwdir, g.mesh.flwacc, g.mesh.path, g.mesh.active_cell
    mesh.gauge_pos = np.asfortranarray(gp[loc].reshape(-1, 2)) if loc else np.zeros((0, 2), np.int32, order="F")
    mesh.area = np.asarray(g.mesh.area)[loc] if loc else np.zeros(0, np.float32)
    return setup, mesh, loc


def _apply_opts(setup, g, loc):
    """The fixture's own calibration options on a tile's setup: criteria, start step, normalisation, regularisers, flags and bounds as
    they are, the weights of the tile's gauges; returns (nslots, slot_of_gauge) of the median over the decomposition's gauges."""
    from smash_amd import tiles
    o = setup.optimize
    o.jobs_fun = list(g.opts.get("jobs_fun", ("nse",)))
    o.wjobs_fun = list(g.opts.get("wjobs_fun", (1.0,)))
    o.optimize_start_step = int(g.opts.get("optimize_start_step", 1))
    o.denormalize_forward = bool(g.opts.get("denormalize_forward", False))
    wg = np.asarray(g.opts.get("wgauge", np.full(g.mesh.ng, 1.0 / g.mesh.ng)), np.float32)
    o.wgauge = wg[loc] if loc else np.zeros(0, np.float32)
    if "jreg_fun" in g.opts:
        o.jreg_fun, o.wjreg_fun, o.wjreg = list(g.opts["jreg_fun"]), list(g.opts["wjreg_fun"]), float(g.opts["wjreg"])
    if "optim_parameters" in g.opts:
        o.optim_parameters = np.asarray(g.opts["optim_parameters"], np.int32)
        o.optim_states = np.asarray(g.opts["optim_states"], np.int32)
    return tiles.median_slots(wg, loc)


class _SumOverTiles:
    """reduce_fn of Solver.set_median_slots for plans that live in one process: every tile's thread adds its values, all read the sum."""

    def __init__(self, world):
        self.bar, self.lock, self.acc = threading.Barrier(world), threading.Lock(), None

    def __call__(self, vals):
        with self.lock:
            self.acc = vals.copy() if self.acc is None else self.acc + vals
        self.bar.wait(timeout=120)
        vals[:] = self.acc
        if self.bar.wait(timeout=120) == 0:
            self.acc = None


def _short(name, nt):
    g = gu.load(name)
    g.nt = nt                                     # a short window keeps 8 plans on one card cheap
    g.prcp, g.pet, g.qobs = np.asfortranarray(g.prcp[:, :, :nt]), np.asfortranarray(g.pet[:, :, :nt]), np.asfortranarray(g.qobs[:, :nt])
    g.opts = {}
    return g


@pytest.mark.parametrize("world,chunk,pipe", [(2, 96, 16), (4, 32, 16), (8, 96, 32)])
def test_tiled_sweep_equals_single_domain(world, chunk, pipe):
    _check_partitioned(_short("gr_b_64x64x720_nse", 96), world, chunk, pipe, None)


@pytest.mark.parametrize("name,world,chunk,pipe,cut", [("gr_b_64x64x720_nse", 4, 32, 16, None), ("gr_b_64x64x720_nse", 2, 96, 32, None),
                                                       ("gr_b_64x64x720_nse", 3, 32, 16, "sub"), ("gr_b_64x64x720_nse", 2, 96, 32, "trunk")])
def test_tiles_with_staging_rows_equal_single_domain(name, world, chunk, pipe, cut, monkeypatch):
    """A plan with many chained groups gives the chained launches their inputs in staging rows (sx_k_chain_transpose); small plans do not by
    themselves.  Forced on (SMASHX_CHAIN_STAGE=1) over decompositions: the series received from other tiles -- inlets of chained groups
    fed through the exchange rows, kept per storage chunk for the recomputation -- pass through the copy like the local ones, and the
    adjoint series handed back leave through it.  Bit-identical to the single domain (which runs without them)."""
    from smash_amd import tiles
    g = _short(name, 96)
    owner = None if cut is None else (tiles.partition_subcatchments(g.mesh, world) if cut == "sub" else tiles.partition_trunk(g.mesh, world))
    ref = None
    import test_gpu_parity
    monkeypatch.setenv("SMASHX_CHAIN_STAGE", "0")
    ref = test_gpu_parity._run_adjoint(g)
    monkeypatch.setenv("SMASHX_CHAIN_STAGE", "1")
    tm = _check_partitioned(g, world, chunk, pipe, owner, reference=ref, group_size=64)      # (small groups: several rounds even on a 32 x 32 tile)
    assert any(t["chain_staged"] for t in tm.values()), "no part of this decomposition has chained rounds: the case tests nothing"


@pytest.mark.parametrize("name,world,chunk,pipe,cut", [("gr_c_32x32x240_d8_ragged", 3, 96, 16, "sub"), ("gr_b_20x20x96_d8", 4, 32, 16, "sub"),
                                                       ("vic_a_24x24x240_d8_kge", 5, 96, 32, "sub"),
                                                       ("gr_c_32x32x240_d8_ragged", 4, 96, 16, "trunk"), ("gr_a_cance_28x28x1440", 3, 96, 32, "trunk")])
def test_subcatchment_partition_equals_single_domain(name, world, chunk, pipe, cut):
    """A flow field with all eight D8 codes cut into sub-catchment parts (tiles.partition_subcatchments): each part is
    a plan with an owner mask, the boundary series travel between the plans as between ranks."""
    from smash_amd import tiles
    g = _short(name, 96)
    owner = tiles.partition_subcatchments(g.mesh, world) if cut == "sub" else tiles.partition_trunk(g.mesh, world)
    _check_partitioned(g, world, chunk, pipe, owner)


def _check_partitioned(g, world, chunk, pipe, owner, keep_opts=False, reference=None, group_size=128):
    import torch
    torch.zeros(1, device="cuda")                 # initialise torch's HIP context in the main thread
    import smash_amd
    from smash_amd import tiles
    from smash_amd.solver import Solver
    from test_gpu_parity import _run_adjoint
    _, _, ref_out, ref_pb, ref_sb = reference if reference is not None else _run_adjoint(g)
    summer = _SumOverTiles(world)
    pr, pc = tiles.tile_grid(world)
    nrow, ncol = g.mesh.nrow, g.mesh.ncol
    box = {(a, b, k): queue.Queue() for a in range(world) for b in range(world) for k in "fa"}
    res, errs, timings = {}, [], {}

    def run(rank):
        try:
            if owner is None:
                rect = tiles.tile_rect(rank, nrow, ncol, pr, pc)
                setup, mesh, loc = _tile_inputs(g, rect, g.mesh.ng)
                sol = Solver(setup, mesh, chunk_steps=chunk, pipe_steps=pipe, group_size=group_size, tile=rect)
            else:
                mine = np.asarray(owner) == rank
                setup, mesh, loc = _tile_inputs(g, None, g.mesh.ng, mine)
                sol = Solver(setup, mesh, chunk_steps=chunk, pipe_steps=pipe, group_size=group_size, owner_mask=mine)
                assert sol.ncells == int(mine.sum())
            rows, cols = sol.cell_order()
            sol.set_forcing(g.prcp, g.pet)
            if loc:
                sol.set_qobs(np.asfortranarray(g.qobs[loc]))
            if keep_opts:
                nslots, slots = _apply_opts(setup, g, loc)
                if nslots:
                    sol.set_median_slots(nslots, slots, summer)
            sol.set_options(setup.optimize)
            ex = Loopback(rank, sol, nrow, ncol, pr, pc, box, owner)
            par = smash_amd.ParametersDT.from_dict(mesh, g.params)
            sta = smash_amd.StatesDT.from_dict(mesh, g.states)
            out = smash_amd.OutputDT(setup, mesh)
            pb, sb = par.copy(), sta.copy()
            if keep_opts and "params_bgd" in g.opts:
                sol.upload(par, sta, smash_amd.ParametersDT.from_dict(mesh, g.opts["params_bgd"]), smash_amd.StatesDT.from_dict(mesh, g.opts["states_bgd"]))
            else:
                sol.upload(par, sta)
            sol.sweep(True, 1.0)
            sol.download(True, par, sta, out, pb, sb)
            res[rank] = (loc, out, pb, sb, rows, cols, ex.calls)
            timings[rank] = sol.timing()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs and len(res) == world
    cost = 0.0
    owned = np.zeros((nrow, ncol), bool)
    for rank, (loc, out, pb, sb, rows, cols, calls) in res.items():
        # every rank evaluates the regulariser over the whole grid: the same cost_jreg everywhere, counted once (smashx.h)
        cost += out.cost_jobs
        assert out.cost_jreg == ref_out.cost_jreg, (rank, out.cost_jreg, ref_out.cost_jreg)
        owned[rows, cols] = True
        for i, gi in enumerate(loc):
            assert np.array_equal(out.qsim[i], ref_out.qsim[gi]), (rank, gi)
        for k in gu.STRUCT_PARAMS[g.structure]:
            assert np.array_equal(getattr(pb, k)[rows, cols], getattr(ref_pb, k)[rows, cols]), (rank, k)
        for k in gu.STRUCT_STATES[g.structure]:
            assert np.array_equal(getattr(sb, k)[rows, cols], getattr(ref_sb, k)[rows, cols]), (rank, k)
    cost = tiles.decomposition_cost([cost], ref_out.cost_jreg, float(g.opts.get("wjreg", 0.0)) if keep_opts else 0.0)
    assert abs(cost - ref_out.cost) <= 1e-6 * abs(ref_out.cost) + 1e-7, (cost, ref_out.cost)
    assert sum(r[6] for r in res.values()) > 0       # the exchange really ran
    if keep_opts and "jreg_fun" in g.opts:
        # cells nobody owns (inactive ones) only carry the regulariser's gradient, which every rank holds for the whole grid
        _, _, pb0, sb0, _, _, _ = res[0]
        for k in gu.STRUCT_PARAMS[g.structure]:
            assert np.array_equal(getattr(pb0, k)[~owned], getattr(ref_pb, k)[~owned]), k
    return timings


def test_regularisation_on_tiles_equals_single_domain():
    """compute_jreg (prior + smoothing + hard smoothing, normalised control, mwd_cost.f90:159-245, 1100-1221) on a 2 x 2 decomposition:
    every plan holds whole parameter planes, evaluates the reference's ordered sums over the WHOLE grid and takes the gradient of its own
    cells -- cost_jreg and every gradient field bit-identical to the single domain, the cost = sum(cost_jobs) + wjreg * cost_jreg."""
    _check_partitioned(gu.load("gr_b_24x24x120_norm_jreg"), 4, 64, 16, None, keep_opts=True)


@pytest.mark.parametrize("name,world", [("gr_a_16x16x96_median3", 4), ("gr_b_16x16x96_median2", 2)])
def test_median_over_gauges_of_several_tiles(name, world):
    """wgauge < 0: the cost is the median of the gauges' criteria (mwd_cost.f90:139-154, quantile1d_r).  The gauges sit in different
    sub-catchment parts: each plan computes its own gauges' criteria, the slots are summed over the parts between the two phases of
    the cost kernel (smashx_set_median_slots), every plan takes the median of all -- discharge and gradients bit-identical to the
    single domain, the parts' cost_jobs add up to the reference's cost."""
    from smash_amd import tiles
    g = gu.load(name)
    owner = tiles.partition_subcatchments(g.mesh, world)
    wg = np.asarray(g.opts["wgauge"])
    gp = np.asarray(g.mesh.gauge_pos)
    assert len({int(np.asarray(owner)[r, c]) for (r, c), w in zip(gp, wg) if w < 0}) > 1      # the median really spans parts
    _check_partitioned(g, world, 96, 32, owner, keep_opts=True)


def test_tile_refuses_self_sized_chunks():
    """A tile with boundary series must be told its chunk length: sizing it from this rank's free HBM (chunk_steps = 0) would
    cut time differently on different ranks and end in mismatched messages (smashx.h "native exchange")."""
    import smash_amd
    from smash_amd import _lib, tiles
    from smash_amd.solver import Solver
    g = _short("gr_b_64x64x720_nse", 96)
    rect = tiles.tile_rect(0, g.mesh.nrow, g.mesh.ncol, 1, 2)
    setup, mesh, loc = _tile_inputs(g, rect, g.mesh.ng)
    sol = Solver(setup, mesh, chunk_steps=0, pipe_steps=16, tile=rect)
    with pytest.raises(smash_amd.SmashxError) as e:
        sol.chunking()
    assert e.value.code == _lib.E_ARG and "chunk_steps" in str(e.value)


@pytest.mark.parametrize("world,with_jreg,auto", [(4, False, None), (2, True, None), (2, True, "fast"), (2, True, "lcurve")])
def test_calibration_over_tiles_follows_the_single_domain(world, with_jreg, auto):
    """smash_amd.optimize_lbfgsb(decomposition=...): the distributed L-BFGS-B calibration (mw_optimize.f90:484-676) over the plans of
    a tile decomposition -- rank 0 runs the optimiser, every trial point is one collective forward_b, the parts' costs are summed,
    every part contributes the gradient of its own cells (the regulariser's term, evaluated over the whole grid by every part,
    counted once).  Against the same calibration on the single domain: same number of iterations and evaluations, costs and
    calibrated fields equal to the rounding of the cost sum (parts' fp32 costs added in double).
    auto = 'fast' | 'lcurve': the cycles that choose the regularisation weight (core/simulation/_optimize.py:257-453) over the
    decomposition -- every rank must try the same weights, pick the same one and end with the single domain's."""
    import os
    import torch
    torch.zeros(1, device="cuda")
    import smash_amd
    from smash_amd import synth, tiles
    from smash_amd.solver import Solver
    from test_gpu_parity import _types
    z = np.load(os.path.join(gu.GOLDEN_DIR, "lbfgsb", "opt_gr_b_24x24x120.npz"))
    g = gu.load("gr_b_24x24x120_norm_jreg")
    if not with_jreg:
        g.opts = dict(jobs_fun=("nse",), wjobs_fun=(1.0,))
    else:
        g.opts = {k: v for k, v in g.opts.items() if k not in ("params_bgd", "states_bgd")}
    g.params, g.states, g.qobs = synth.make_parameters(24, 24), synth.make_states(24, 24, warm=True), z["qobs"]
    op = np.asarray(z["optim_parameters"], np.int32)
    maxiter = 3

    setup, mesh, inp, par, sta, out = _types(g)
    setup.optimize.optim_parameters, setup.optimize.maxiter = op, maxiter
    setup.optimize.optim_states = np.zeros_like(np.asarray(setup.optimize.optim_states))
    href = smash_amd.optimize_lbfgsb(setup, mesh, inp, par, sta, out, auto_wjreg=auto, return_lcurve=True)

    pr, pc = tiles.tile_grid(world)
    nrow, ncol = g.mesh.nrow, g.mesh.ncol
    box = {(a, b, k): queue.Queue() for a in range(world) for b in range(world) for k in "fa"}
    shared = tiles.ThreadDecomposition.make(world)
    summer = _SumOverTiles(world)
    res, errs = {}, []

    def run(rank):
        try:
            rect = tiles.tile_rect(rank, nrow, ncol, pr, pc)
            st, ms, loc = _tile_inputs(g, rect, g.mesh.ng)
            nslots, slots = _apply_opts(st, g, loc)
            st.optimize.optim_parameters, st.optimize.maxiter = op, maxiter
            st.optimize.optim_states = np.zeros_like(np.asarray(st.optimize.optim_states))
            sol = Solver(st, ms, chunk_steps=64, pipe_steps=16, group_size=128, tile=rect)
            if nslots:
                sol.set_median_slots(nslots, slots, summer)
            ex = Loopback(rank, sol, nrow, ncol, pr, pc, box)
            ip = smash_amd.Input_DataDT(st, ms)
            ip.prcp, ip.pet = g.prcp, g.pet
            ip.qobs = np.asfortranarray(g.qobs[loc]) if loc else np.zeros((0, g.nt), np.float32, order="F")
            ip._smashx_solver = sol
            rows, cols = sol.cell_order()
            owned = np.zeros((nrow, ncol), bool)
            owned[rows, cols] = True
            dec = tiles.ThreadDecomposition(shared, rank, owned)
            pt = smash_amd.ParametersDT.from_dict(ms, g.params)
            stt = smash_amd.StatesDT.from_dict(ms, g.states)
            ot = smash_amd.OutputDT(st, ms)
            h = smash_amd.optimize_lbfgsb(st, ms, ip, pt, stt, ot, decomposition=dec, auto_wjreg=auto, return_lcurve=True)
            h["wjreg_left"] = float(st.optimize.wjreg)
            res[rank] = (h, pt, ot, ex.calls)
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs and len(res) == world
    h0 = res[0][0]
    if auto is not None:
        # the weight every rank ended with is one number, and the single domain's to the rounding of the cost sums it is a ratio of;
        # the L-curve tried the same weights and chose the same corner
        ws = [res[r][0]["wjreg"] for r in range(world)]
        assert all(w == ws[0] for w in ws) and all(res[r][0]["wjreg_left"] == res[0][0]["wjreg_left"] for r in range(world))
        assert (ws[0] is None) == (href["wjreg"] is None)
        if ws[0] is not None:
            assert abs(ws[0] - href["wjreg"]) <= 1e-4 * abs(href["wjreg"]), (ws[0], href["wjreg"])
        if auto == "lcurve":
            assert np.allclose(h0["lcurve"]["wjreg"], href["lcurve"]["wjreg"], rtol=1e-4)
            assert np.allclose(h0["lcurve"]["cost_jobs"], href["lcurve"]["cost_jobs"], rtol=1e-4)
            assert np.array_equal(np.isnan(h0["lcurve"]["distance"]), np.isnan(href["lcurve"]["distance"]))
        assert abs(h0["final_cost"] - href["final_cost"]) <= 1e-4 * abs(href["final_cost"]), (h0["final_cost"], href["final_cost"])
        return
    assert len(h0["cost"]) == len(href["cost"]) == maxiter and h0["nfg"] == href["nfg"]
    assert np.allclose(h0["cost"], href["cost"], rtol=2e-6, atol=0), (h0["cost"], href["cost"])
    assert abs(h0["final_cost"] - href["final_cost"]) <= 2e-6 * abs(href["final_cost"])
    act = np.asarray(g.mesh.active_cell) == 1
    for rank, (h, pt, ot, calls) in res.items():
        assert h["final_cost"] == h0["final_cost"] and calls > 0             # every rank ends with the same cost and really exchanged
        for i, k in enumerate(synth.PARAM_NAMES):
            if op[i] > 0:                                                     # every rank holds the whole calibrated field
                a, b = getattr(pt, k)[act], getattr(par, k)[act]
                assert np.max(np.abs(a - b)) <= 1e-4 * np.max(np.abs(b)), (rank, k, float(np.max(np.abs(a - b))))


@pytest.mark.parametrize("sparse", [False, True])
def test_domain_outputs_of_the_parts_overlay_to_the_single_domain(sparse):
    """setup%save_qsim_domain / save_net_prcp_domain (md_forward_structure.f90:158-194) on a decomposition: every part fills the cells
    it owns in the dense (nrow, ncol, nt) arrays and leaves -99 elsewhere; overlaid, the parts give the single domain's arrays bit
    for bit (forward sweep in storage chunks and pipeline sub-chunks, boundary series through the in-process exchange).
    sparse: the same with setup%sparse_storage -- the parts READ the forcing from the whole grid's (nac, nt) sparse vectors
    (mw_sparse_storage.f90:12-49) and WRITE the (nac, nt) sparse_ output forms, a part's cells at their whole-grid numbers."""
    import torch
    torch.zeros(1, device="cuda")
    import smash_amd
    from smash_amd import tiles
    from smash_amd.solver import Solver
    from test_gpu_parity import _types
    g = _short("gr_b_64x64x720_nse", 96)
    world = 4
    setup, mesh, inp, par, sta, out = _types(g)
    setup.save_qsim_domain = setup.save_net_prcp_domain = True
    out = smash_amd.OutputDT(setup, mesh)
    s0 = Solver(setup, mesh, chunk_steps=32)
    s0.set_forcing(inp.prcp, inp.pet)
    inp._smashx_solver = s0
    smash_amd.forward(setup, mesh, inp, par, par.copy(), sta, sta.copy(), out, np.float32(0))
    ref_q, ref_p = out.qsim_domain.copy(), out.net_prcp_domain.copy()
    pr, pc = tiles.tile_grid(world)
    nrow, ncol = g.mesh.nrow, g.mesh.ncol
    box = {(a, b, k): queue.Queue() for a in range(world) for b in range(world) for k in "fa"}
    res, errs = {}, []
    # the sparse numbering: active cells in the order of mesh%path (all cells are active in this fixture's mask or not: take the mask)
    path = np.asarray(g.mesh.path)
    act = np.asarray(g.mesh.active_cell)
    pr_, pc_ = path[0], path[1]
    keep = (pr_ >= 0) & (pc_ >= 0)
    keep[keep] &= act[pr_[keep], pc_[keep]] == 1
    srow, scol = pr_[keep], pc_[keep]
    nac = int(keep.sum())
    sp_prcp = np.asfortranarray(g.prcp[srow, scol, :]) if sparse else None
    sp_pet = np.asfortranarray(g.pet[srow, scol, :]) if sparse else None

    def run(rank):
        try:
            rect = tiles.tile_rect(rank, nrow, ncol, pr, pc)
            st, ms, loc = _tile_inputs(g, rect, g.mesh.ng)
            sol = Solver(st, ms, chunk_steps=32, pipe_steps=16, group_size=128, tile=rect)
            if sparse:
                sol.set_forcing(sp_prcp, sp_pet, sparse=True)
            else:
                sol.set_forcing(g.prcp, g.pet)
            if loc:
                sol.set_qobs(np.asfortranarray(g.qobs[loc]))
            sol.set_options(st.optimize)
            Loopback(rank, sol, nrow, ncol, pr, pc, box)
            shp = (nac, g.nt) if sparse else (nrow, ncol, g.nt)
            q = np.zeros(shp, np.float32, order="F")
            pn = np.zeros(shp, np.float32, order="F")
            sol.set_domain_outputs(q, pn, sparse=sparse)
            sol.upload(smash_amd.ParametersDT.from_dict(ms, g.params), smash_amd.StatesDT.from_dict(ms, g.states))
            sol.sweep(False, 0.0)
            rows, cols = sol.cell_order()
            res[rank] = (q, pn, rows, cols)
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs and len(res) == world
    if sparse:
        ref_q, ref_p = ref_q[srow, scol, :], ref_p[srow, scol, :]              # the single domain's dense arrays in the sparse numbering
    q_all = np.full_like(ref_q, -99.0)
    p_all = np.full_like(ref_p, -99.0)
    for rank, (q, pn, rows, cols) in res.items():
        own = np.zeros((nrow, ncol), bool)
        own[rows, cols] = True
        if sparse:
            own = own[srow, scol]
        assert np.all(q[~own] == -99.0) and np.all(pn[~own] == -99.0)          # a part writes its own cells only
        q_all[own], p_all[own] = q[own], pn[own]
    assert np.array_equal(q_all, ref_q) and np.array_equal(p_all, ref_p)


@pytest.mark.parametrize("name,world,nt,cut", [("gr_b_64x64x720_nse", 4, 96, None), ("gr_c_32x32x240_d8_ragged", 3, 96, "sub"),
                                              ("gr_b_24x24x120_norm_jreg", 2, 120, None)])
def test_tangent_model_over_tiles_equals_single_domain(name, world, nt, cut):
    """smashx_forward_d (base_forward_d, forward_db.f90:10517-10601) on a decomposition: the boundary series of the value pass and of
    the tangent pass travel through the plans' exchange, a message per pipeline sub-chunk.  Against the single domain along the same
    direction (1 on every parameter field, as mw_adjoint_test's scalar product test): discharge and its tangent at every part's
    gauges bit for bit; cost_d = the parts' criteria terms + the regulariser's term once, to the rounding of that sum."""
    import torch
    torch.zeros(1, device="cuda")
    import smash_amd
    from smash_amd import synth, tiles
    from smash_amd.solver import Solver, _tangent_call
    from test_gpu_parity import _types
    g = gu.load(name)
    keep_opts = dict(g.opts) if "jreg" in name else {}
    if nt < g.nt:
        g = _short(name, nt)
    g.opts = {k: v for k, v in keep_opts.items() if k not in ("params_bgd", "states_bgd")}

    def direction(mesh):
        pd = smash_amd.ParametersDT.from_dict(mesh, g.params)
        sd = smash_amd.StatesDT.from_dict(mesh, g.states)
        for k in synth.PARAM_NAMES:
            getattr(pd, k)[...] = 1.0
        for k in synth.STATE_NAMES:
            getattr(sd, k)[...] = 0.0
        return pd, sd

    setup, mesh, inp, par, sta, out = _types(g)
    s0 = Solver(setup, mesh, chunk_steps=32)
    s0.set_forcing(g.prcp, g.pet)
    s0.set_qobs(g.qobs)
    s0.set_options(setup.optimize)
    pd, sd = direction(mesh)
    out_d = smash_amd.OutputDT(setup, mesh)
    cost0, cost_d0 = _tangent_call(s0, par, pd, par.copy(), sta, sd, sta.copy(), out, out_d)
    jobs_d0, jreg_d0 = s0.tangent_terms()
    ref_q, ref_qd = np.array(out.qsim), np.array(out_d.qsim)

    nrow, ncol = g.mesh.nrow, g.mesh.ncol
    pr, pc = tiles.tile_grid(world)
    owner = tiles.partition_subcatchments(g.mesh, world) if cut == "sub" else None
    box = {(a, b, k): queue.Queue() for a in range(world) for b in range(world) for k in "fa"}
    res, errs = {}, []

    def run(rank):
        try:
            if owner is not None:
                mine = np.asarray(owner) == rank
                st, ms, loc = _tile_inputs(g, None, g.mesh.ng, mine)
                sol = Solver(st, ms, chunk_steps=32, pipe_steps=16, group_size=128, owner_mask=mine)
            else:
                rect = tiles.tile_rect(rank, nrow, ncol, pr, pc)
                st, ms, loc = _tile_inputs(g, rect, g.mesh.ng)
                sol = Solver(st, ms, chunk_steps=32, pipe_steps=16, group_size=128, tile=rect)
            _apply_opts(st, g, loc)
            sol.set_forcing(g.prcp, g.pet)
            if loc:
                sol.set_qobs(np.asfortranarray(g.qobs[loc]))
            sol.set_options(st.optimize)
            ex = Loopback(rank, sol, nrow, ncol, pr, pc, box, owner)
            pt, stt = smash_amd.ParametersDT.from_dict(ms, g.params), smash_amd.StatesDT.from_dict(ms, g.states)
            ptd, std = direction(ms)
            ot, otd = smash_amd.OutputDT(st, ms), smash_amd.OutputDT(st, ms)
            _tangent_call(sol, pt, ptd, pt.copy(), stt, std, stt.copy(), ot, otd)
            res[rank] = (loc, np.array(ot.qsim) if loc else None, np.array(otd.qsim) if loc else None, sol.tangent_terms(), ex.calls)
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs and len(res) == world
    jobs_d, seen = 0.0, 0
    for rank, (loc, q, qd, (jd, jr), calls) in res.items():
        assert calls > 0
        jobs_d += jd
        assert jr == jreg_d0                                   # the regulariser's tangent: the whole grid's, bit for bit, on every part
        if loc:
            seen += len(loc)
            assert np.array_equal(q, ref_q[loc]) and np.array_equal(qd, ref_qd[loc]), rank
    assert seen == g.mesh.ng
    assert abs(jobs_d - jobs_d0) <= 2e-5 * abs(jobs_d0) + 1e-12, (jobs_d, jobs_d0)
    total = jobs_d + float(setup.optimize.wjreg) * jreg_d0
    assert abs(total - cost_d0) <= 2e-5 * abs(cost_d0) + 1e-12, (total, cost_d0)
