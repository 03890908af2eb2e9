#!/usr/bin/env python3
"""One rank of a multi-process tile run (launched by tests/test_gpu_rccl.py under torch.distributed.run): builds its tile of a
golden case, exchanges the boundary series with the other ranks over REAL RCCL -- natively (grouped ncclSend / ncclRecv
posted by libsmashx on its routing stream) or through the torch.distributed callback -- and checks that discharge and every
gradient field are bit-identical to the single-domain run it computes on the same GPU.  On a box with fewer GPUs than ranks
the ranks share a device (tiles.share_one_gpu_env: one host id per rank, socket transport)."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="gr_b_64x64x720_nse")
    ap.add_argument("--nt", type=int, default=96)
    ap.add_argument("--chunk", type=int, default=96)
    ap.add_argument("--pipe", type=int, default=16)
    ap.add_argument("--exchange", default="rccl")
    ap.add_argument("--cut", default="rect")
    ap.add_argument("--opts", action="store_true", help="the fixture's own calibration options (criteria, normalisation, regularisers, "
                    "gauge weights incl. the median over gauges) instead of the plain nse of the short window")
    ap.add_argument("--calibrate", type=int, default=0, help="with --opts: also run N iterations of optimize_lbfgsb over the decomposition "
                    "(tiles.TorchDecomposition over this process group) and compare with the single-domain calibration")
    ap.add_argument("--tangent", action="store_true", help="also run the tangent model (smashx_forward_d) over the decomposition and "
                    "compare discharge tangents and cost_d with the single domain's")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

    def say(msg):                                   # progress on stderr: a hang must be locatable from the log
        print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)
    import torch
    import torch.distributed as dist
    from smash_amd import tiles
    ndev = max(torch.cuda.device_count(), 1)
    if ndev < world:
        tiles.share_one_gpu_env(rank)
    local = int(os.environ.get("LOCAL_RANK", rank)) % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.zeros(1, device=dev)                      # torch's HIP runtime initialises first (tests/conftest.py)
    dist.init_process_group("nccl", device_id=dev)
    dist.barrier()
    say("process group up")

    import golden_util as gu
    import smash_amd
    from smash_amd.solver import Comm, Solver
    from test_gpu_parity import _run_adjoint
    from test_gpu_tiles import _apply_opts, _short, _tile_inputs
    g = gu.load(a.case) if a.opts else _short(a.case, a.nt)
    _, _, ref_out, ref_pb, ref_sb = _run_adjoint(g)
    pr, pc = tiles.tile_grid(world)
    nrow, ncol = g.mesh.nrow, g.mesh.ncol
    owner = None
    if a.cut == "rect":
        rect = tiles.tile_rect(rank, nrow, ncol, pr, pc)
        setup, mesh, loc = _tile_inputs(g, rect, g.mesh.ng)
        sol = Solver(setup, mesh, chunk_steps=a.chunk, pipe_steps=a.pipe, group_size=128, device=local, tile=rect)
    else:
        owner = tiles.partition_subcatchments(g.mesh, world) if a.cut == "sub" else tiles.partition_trunk(g.mesh, world)
        mine = np.asarray(owner) == rank
        setup, mesh, loc = _tile_inputs(g, None, g.mesh.ng, mine)
        sol = Solver(setup, mesh, chunk_steps=a.chunk, pipe_steps=a.pipe, group_size=128, device=local, owner_mask=mine)
    rows, cols = sol.cell_order()
    say(f"plan built: {sol.ncells} cells, edges {sol.halo_counts()}")
    sol.set_forcing(g.prcp, g.pet)
    if loc:
        sol.set_qobs(np.asfortranarray(g.qobs[loc]))
    nslots = 0
    if a.opts:
        nslots, slots = _apply_opts(setup, g, loc)
        if nslots:                                  # the median's slots are summed over the ranks by RCCL on the routing stream (native exchange)
            assert a.exchange == "rccl"
            sol.set_median_slots(nslots, slots)
    sol.set_options(setup.optimize)
    comm = None
    if a.exchange == "rccl":
        uid = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0, device=dev)
        comm = Comm(uid[0], rank, world, local)
        ex = tiles.RcclExchange(sol, comm, nrow, ncol, pr, pc, owner)
    else:
        ex = tiles.TorchDistExchange(sol, nrow, ncol, pr, pc, dev, owner)
    say(f"exchange set ({a.exchange}): out peers {sorted(ex.peers.out_peers)} in peers {sorted(ex.peers.in_peers)}")
    par = smash_amd.ParametersDT.from_dict(mesh, g.params)
    sta = smash_amd.StatesDT.from_dict(mesh, g.states)
    out = smash_amd.OutputDT(setup, mesh)
    pb, sb = par.copy(), sta.copy()
    bad = []
    bgd = ()
    if a.opts and "params_bgd" in g.opts:
        bgd = (smash_amd.ParametersDT.from_dict(mesh, g.opts["params_bgd"]), smash_amd.StatesDT.from_dict(mesh, g.opts["states_bgd"]))
    for rep in range(2):                            # twice: the second sweep reuses every buffer and the communicator
        if a.opts:                                  # (a normalised control comes back denormalised: start from the fixture's fields again)
            par, sta = smash_amd.ParametersDT.from_dict(mesh, g.params), smash_amd.StatesDT.from_dict(mesh, g.states)
        sol.upload(par, sta, *bgd)
        sol.sweep(True, 1.0)
        say(f"sweep {rep} done")
        sol.download(True, par, sta, out, pb, sb)
        for i, gi in enumerate(loc):
            if not np.array_equal(out.qsim[i], ref_out.qsim[gi]):
                bad.append(("qsim", gi, rep))
        for k in gu.STRUCT_PARAMS[g.structure]:
            if not np.array_equal(getattr(pb, k)[rows, cols], getattr(ref_pb, k)[rows, cols]):
                bad.append((k, rep))
        for k in gu.STRUCT_STATES[g.structure]:
            if not np.array_equal(getattr(sb, k)[rows, cols], getattr(ref_sb, k)[rows, cols]):
                bad.append((k, rep))
    # the regulariser is evaluated over the whole grid by every rank (same bits as the single domain) and enters the cost once
    if out.cost_jreg != ref_out.cost_jreg:
        bad.append(("cost_jreg", out.cost_jreg, ref_out.cost_jreg))
    if comm is not None:
        jobs = float(comm.allreduce_sum([out.cost_jobs])[0])
    else:
        ct = torch.tensor([out.cost_jobs], dtype=torch.float64, device=dev)
        dist.all_reduce(ct)
        jobs = float(ct.item())
    cost = tiles.decomposition_cost([jobs], out.cost_jreg, float(g.opts.get("wjreg", 0.0)) if a.opts else 0.0)
    if abs(cost - ref_out.cost) > 1e-6 * abs(ref_out.cost) + 1e-7:
        bad.append(("cost", cost, ref_out.cost))
    if a.calibrate:
        # the distributed calibration (mw_optimize.f90:484-676) over the ranks: rank 0 runs L-BFGS-B, every trial point is one collective
        # forward_b over RCCL; against the same calibration on the single domain (every rank computes that reference itself)
        from test_gpu_parity import _types
        s1, m1, i1, p1, t1, o1 = _types(g)
        s1.optimize.maxiter = a.calibrate
        href = smash_amd.optimize_lbfgsb(s1, m1, i1, p1, t1, o1)
        say(f"single-domain calibration: {href['cost']}")
        setup.optimize.maxiter = a.calibrate
        ip = smash_amd.Input_DataDT(setup, mesh)
        ip.prcp, ip.pet = g.prcp, g.pet
        ip.qobs = np.asfortranarray(g.qobs[loc]) if loc else np.zeros((0, g.nt), np.float32, order="F")
        ip._smashx_solver = sol
        owned = np.zeros((nrow, ncol), bool)
        owned[rows, cols] = True
        o = setup.optimize
        nctl = int(np.count_nonzero(np.asarray(o.optim_parameters) > 0) + np.count_nonzero(np.asarray(o.optim_states) > 0)) * int(np.count_nonzero(np.asarray(g.mesh.active_cell) == 1))
        dec = tiles.TorchDecomposition(owned, nctl, device=dev)
        pt, tt = smash_amd.ParametersDT.from_dict(mesh, g.params), smash_amd.StatesDT.from_dict(mesh, g.states)
        ot = smash_amd.OutputDT(setup, mesh)
        h = smash_amd.optimize_lbfgsb(setup, mesh, ip, pt, tt, ot, decomposition=dec)
        say(f"calibration over {world} ranks: {h['cost'] if rank == 0 else h['final_cost']}")
        if rank == 0 and (len(h["cost"]) != len(href["cost"]) or h["nfg"] != href["nfg"] or not np.allclose(h["cost"], href["cost"], rtol=2e-6, atol=0)):
            bad.append(("calibration", h["cost"], href["cost"], h["nfg"], href["nfg"]))
        if abs(h["final_cost"] - href["final_cost"]) > 2e-6 * abs(href["final_cost"]):
            bad.append(("final cost", h["final_cost"], href["final_cost"]))
        act = np.asarray(g.mesh.active_cell) == 1
        from smash_amd.synth import PARAM_NAMES
        for i, k in enumerate(PARAM_NAMES):
            if o.optim_parameters[i] > 0:
                d = float(np.max(np.abs(getattr(pt, k)[act] - getattr(p1, k)[act])))
                if d > 1e-4 * float(np.max(np.abs(getattr(p1, k)[act]))):
                    bad.append(("field", k, d))
    if a.tangent:
        # base_forward_d over the ranks: value and tangent boundary series through the same exchange; direction = 1 on every parameter
        from smash_amd.solver import _tangent_call
        from smash_amd.synth import PARAM_NAMES, STATE_NAMES
        from test_gpu_parity import _types

        def direction(m_):
            pd_, sd_ = smash_amd.ParametersDT.from_dict(m_, g.params), smash_amd.StatesDT.from_dict(m_, g.states)
            for k in PARAM_NAMES:
                getattr(pd_, k)[...] = 1.0
            for k in STATE_NAMES:
                getattr(sd_, k)[...] = 0.0
            return pd_, sd_
        if not a.opts:
            g.opts = {}
        s1, m1, i1, p1, t1, o1 = _types(g)
        r0 = Solver(s1, m1, chunk_steps=a.chunk, device=local)
        r0.set_forcing(g.prcp, g.pet)
        r0.set_qobs(g.qobs)
        r0.set_options(s1.optimize)
        pd1, sd1 = direction(m1)
        od1 = smash_amd.OutputDT(s1, m1)
        _, cost_d_ref = _tangent_call(r0, p1, pd1, p1.copy(), t1, sd1, t1.copy(), o1, od1)
        jobs_d_ref, jreg_d_ref = r0.tangent_terms()
        r0.close()
        for rep in range(2):
            pt, tt = smash_amd.ParametersDT.from_dict(mesh, g.params), smash_amd.StatesDT.from_dict(mesh, g.states)
            ptd, ttd = direction(mesh)
            ot, otd = smash_amd.OutputDT(setup, mesh), smash_amd.OutputDT(setup, mesh)
            _tangent_call(sol, pt, ptd, pt.copy(), tt, ttd, tt.copy(), ot, otd)
            say(f"tangent sweep {rep} done")
            jd, jr = sol.tangent_terms()
            for i, gi in enumerate(loc):
                if not np.array_equal(ot.qsim[i], o1.qsim[gi]) or not np.array_equal(otd.qsim[i], od1.qsim[gi]):
                    bad.append(("tangent qsim / qsim_d", gi, rep))
            if jr != jreg_d_ref:
                bad.append(("jreg_d", jr, jreg_d_ref))
            ct = torch.tensor([jd], dtype=torch.float64, device=dev)
            dist.all_reduce(ct)
            total = float(ct.item()) + float(setup.optimize.wjreg) * jr
            if abs(total - cost_d_ref) > 2e-5 * abs(cost_d_ref) + 1e-12:
                bad.append(("cost_d", total, cost_d_ref))
    n_out, n_in = sol.halo_counts()
    print(f"rank {rank}/{world} [{a.exchange}, {a.cut}]: cells {sol.ncells}, edges out {n_out} in {n_in}, chunking {sol.chunking()}, "
          f"{'BIT-IDENTICAL' if not bad else 'MISMATCH ' + str(bad[:6])}", flush=True)
    dist.barrier()
    sol.close()
    if comm is not None:
        comm.close()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
