#!/usr/bin/env python3
"""bench.py -- cell-timesteps/s of one forward + adjoint sweep (the work of one reference forward_b call:
cost + gradient w.r.t. all distributed parameters and initial states) on a synthetic catchment with
hourly x 1 yr forcing resident in HBM (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward+adjoint sweep of the hot path over the whole (grid x nt) batch.  At N = 1 the
workload is BASELINE.json configs[2]: 1024 x 1024 grid, 8760 steps, gr-b (4096^2, the grid the metric is
quoted on, needs 1.18 TB of forcing and only fits 8 GPUs).  Prints ONE JSON line on rank 0.

PyTorch is plumbing here (device memory for building the forcing in HBM, streams, torch.distributed);
all solver arithmetic is in libsmashx (HIP) behind the C ABI.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def pmc_traffic(kernel, cellsteps_per_launch):
    """HBM bytes per launch of `kernel` from the PMC counters.  bench.py cannot collect PMC itself: the values
    come from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very command
    (tools/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950), stored per
    cell-step in profiles/r*_pmc_traffic.json and scaled to this run's launch size.  None if never measured."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1])).get(kernel)
        return d["hbm_bytes_per_cellstep_corrected"] * cellsteps_per_launch if d else None
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=1024, help="cells per side of the per-GPU tile (weak scaling: fixed per GPU)")
    ap.add_argument("--tile-rows", type=int, default=0, help="per-GPU tile rows (default --grid); 2048 x 1024 tiles on 8 GPUs = the 4096^2 grid")
    ap.add_argument("--tile-cols", type=int, default=0)
    ap.add_argument("--pipe", type=int, default=0, help="pipeline sub-chunk (default: none on 1 GPU, 1104 on tiles)")
    ap.add_argument("--partition", default="rect", choices=["rect", "sub", "trunk"],
                    help="N > 1: rectangles (default), sub-catchments (tiles.partition_subcatchments) or the depth-2 trunk cut (tiles.partition_trunk)")
    ap.add_argument("--trunk-share", type=float, default=1.0, help="--partition trunk: the trunk part's share of the cells relative to 1/N")
    ap.add_argument("--as-rank", type=int, default=-1, help="diagnostics, one process: build rank R's part of an --of N decomposition and time it "
                    "alone (no-op exchange, zero inflow): what that rank computes, without waiting for its neighbours")
    ap.add_argument("--of", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo only for rehearsing the N>1 path without RCCL")
    ap.add_argument("--nt", type=int, default=8760)
    ap.add_argument("--structure", default="gr-b")
    ap.add_argument("--chunk", type=int, default=0, help="time-chunk length (0 = from free HBM)")
    ap.add_argument("--group", type=int, default=0, help="routing group size (0 = default)")
    ap.add_argument("--ng", type=int, default=8)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--trace-groups", default="", help="diagnostics: write per-round start/end times of the routing groups (JSON) here")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-grid", type=int, default=256)
    ap.add_argument("--cpu-nt", type=int, default=360)
    ap.add_argument("--cpu-cores", type=int, default=0, help="replicas of the CPU baseline (default: the box's CPU share, at most 16)")
    return ap.parse_args()


def _cpu_replica(structure, n, nt, barrier, q):
    """One replica of the CPU baseline (its own process): the reference's forward_b on the bounded sample."""
    from oracle import pyoracle, refbind
    from smash_amd import synth
    m = synth.make_mesh(n, n, ng=4)
    prcp, pet = synth.dense_forcing(m, nt)
    P, S = synth.make_parameters(n, n), synth.make_states(n, n, warm=True)
    Pq = synth.make_parameters(n, n, perturb=0.1)
    kind = "reference" if refbind.available(fast=True) else "port"
    if kind == "reference":
        qobs = refbind.run(structure, m, 3600.0, prcp, pet, np.zeros((4, nt), np.float32), Pq, S, fast=True)["qsim"]
        barrier.wait()
        t0 = time.perf_counter()
        refbind.run(structure, m, 3600.0, prcp, pet, qobs, P, S, adjoint=True, fast=True)
    else:
        qobs = pyoracle.run(structure, m, 3600.0, prcp, pet, np.zeros((4, nt), np.float32), Pq, S)["qsim"]
        barrier.wait()
        t0 = time.perf_counter()
        pyoracle.run(structure, m, 3600.0, prcp, pet, qobs, P, S, adjoint=True)
    q.put((kind, m.nac * nt, time.perf_counter() - t0))


def cpu_baseline(structure, n, nt, cores=0):
    """The reference path timed on the host cores of this box, on a bounded sample of the same synthetic workload:
    forward_b (cost + gradient) on an n x n catchment over nt steps.  forward_b is single-threaded
    (mw_forward.f90:41-68; no OpenMP in it), so the host is filled the way the reference fills it
    (mw_multiple_run.f90:96-107): one independent replica per core, started together; value = all replicas' work /
    the slowest replica's time."""
    import multiprocessing as mp
    # the CPU share of a one-GPU box is 16 cores whatever the affinity mask says; a replica holds ~2 GB (the reference's
    # tape is 40 B per cell-step), so the pool is sized to that share and never to the machine
    cores = cores or max(1, min(len(os.sched_getaffinity(0)), 16))
    ctx = mp.get_context("spawn")                       # no fork of a process that holds a HIP context
    barrier, q = ctx.Barrier(cores), ctx.Queue()
    procs = [ctx.Process(target=_cpu_replica, args=(structure, n, nt, barrier, q)) for _ in range(cores)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join()
    kind = res[0][0]
    work, slow = sum(r[1] for r in res), max(r[2] for r in res)
    what = ("reference Fortran (flang -O3, libsmash_ref_fast.so) forward_b" if kind == "reference"
            else "plain-C oracle (gcc -O2) forward_b")
    return {"value": work / slow, "unit": "cell-timesteps/s", "cores": cores, "kind": kind,
            "per_core_value": float(np.mean([r[1] / r[2] for r in res])),
            "sample": f"{what}, {structure}, {cores} independent replicas (one per core) of a {n}x{n} synthetic catchment x {nt} "
                      f"hourly steps, slowest replica {slow:.1f} s"}


def main():
    a = parse()
    # the CPU baseline starts one process per core: it runs FIRST, before anything in this process touches the GPU
    # (rank 0 at N = 1 only)
    cpu = None
    if not a.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        try:
            cpu = cpu_baseline(a.structure, a.cpu_grid, a.cpu_nt, a.cpu_cores)
        except Exception as e:  # pragma: no cover
            cpu = {"value": None, "unit": "cell-timesteps/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import smash_amd
    from smash_amd import synth
    from smash_amd.solver import Solver

    from smash_amd import tiles
    n, nt = a.grid, a.nt
    trows, tcols = a.tile_rows or n, a.tile_cols or n
    solo = a.as_rank >= 0 and a.of > 1 and world == 1      # diagnostics: one rank of a decomposition, alone
    parts, me = (a.of, a.as_rank) if solo else (world, rank)
    pr, pc = tiles.tile_grid(parts)
    nrow, ncol = pr * trows, pc * tcols            # the whole catchment; every rank owns 1/parts of it (weak scaling)
    t_setup = time.perf_counter()
    m = synth.make_mesh(nrow, ncol, ng=a.ng)
    rect, owner, mine = None, None, None
    if parts > 1 and a.partition == "rect":
        rect = tiles.tile_rect(me, nrow, ncol, pr, pc)
    elif parts > 1:                                # every rank computes the same cut of the river tree
        owner = tiles.partition_trunk(m, parts, a.trunk_share) if a.partition == "trunk" else tiles.partition_subcatchments(m, parts)
        mine = np.asfortranarray((np.asarray(owner) == me).astype(np.int32))
    gp = np.asarray(m.gauge_pos)
    if mine is not None:
        loc = [i for i in range(m.ng) if mine[gp[i, 0], gp[i, 1]]]
    else:
        loc = [i for i in range(m.ng) if rect is None or (rect[0] <= gp[i, 0] < rect[1] and rect[2] <= gp[i, 1] < rect[3])]
    setup = smash_amd.SetupDT(0, len(loc), structure=a.structure, dt=3600.0, ntime_step=nt)
    setup.optimize.jobs_fun, setup.optimize.wjobs_fun = ["nse"], [1.0]
    setup.optimize.wgauge = np.full(len(loc), 1.0 / m.ng, np.float32)      # weights of the global cost (mean over all gauges)
    mesh = smash_amd.MeshDT(setup, nrow, ncol, len(loc))
    mesh.dx, mesh.flwdir, mesh.flwacc, mesh.path, mesh.active_cell = m.dx, m.flwdir, m.flwacc, m.path, m.active_cell
    mesh.gauge_pos = np.asfortranarray(gp[loc].reshape(-1, 2))
    mesh.area = np.asarray(m.area)[loc]
    chunk, pipe = a.chunk, a.pipe
    if parts > 1:
        # every rank must cut time identically (messages are per sub-chunk): fix the lengths instead of sizing from free HBM
        chunk = chunk or ((nt + 15) // 16 * 16 if trows * tcols <= 1100000 else ((nt + 3) // 4 + 15) // 16 * 16)
        pipe = pipe or 1104
    if a.trace_groups:
        os.environ["SMASHX_TRACE_GROUPS"] = "1"
    sol = Solver(setup, mesh, chunk_steps=chunk, pipe_steps=pipe, group_size=a.group, device=local, tile=rect, owner_mask=mine)
    n = None
    rows, cols = sol.cell_order()
    d_rows = torch.from_numpy(rows.astype(np.int64)).to(dev)
    d_cols = torch.from_numpy(cols.astype(np.int64)).to(dev)
    tb = max(1, min(nt, (1 << 26) // max(sol.ncells, 1)))     # ~64 M cell-steps of int64 temporaries per block
    for t0 in range(0, nt, tb):
        t1 = min(nt, t0 + tb)
        prcp, pet = synth.forcing_block(d_rows, d_cols, t0, t1, xp=torch, device=dev)
        torch.cuda.synchronize()
        sol.set_forcing_device_block(t0, t1, prcp.data_ptr(), pet.data_ptr())
        del prcp, pet
    del d_rows, d_cols
    torch.cuda.empty_cache()

    P, S = synth.make_parameters(nrow, ncol), synth.make_states(nrow, ncol, warm=True)
    par, sta = smash_amd.ParametersDT.from_dict(mesh, P), smash_amd.StatesDT.from_dict(mesh, S)
    out = smash_amd.OutputDT(setup, mesh)
    # observations = forward run with parameters perturbed by +10 %
    parq = smash_amd.ParametersDT.from_dict(mesh, synth.make_parameters(nrow, ncol, perturb=0.1))
    sol.set_options(setup.optimize)
    exchange = tiles.TorchDistExchange(sol, nrow, ncol, pr, pc, dev, owner) if world > 1 else None
    if solo:
        exchange = tiles.NoExchange(sol, dev)
    sol.upload(parq, sta)
    sol.sweep(False)
    sol.download(False, parq, sta, out)
    if len(loc):
        sol.set_qobs(out.qsim)
    sol.upload(par, sta)
    if not a.forward_only:
        sol.chunking()                               # allocates the tapes of the reverse sweep now (set-up, ~4 s for 180 GB), not in the first sweep
    t_setup = time.perf_counter() - t_setup

    adjoint = not a.forward_only
    for _ in range(a.warmup):
        sol.sweep(adjoint)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tm_acc = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sol.sweep(adjoint)          # smashx_sweep synchronises its stream before returning
        for k, v in sol.timing().items():
            tm_acc[k] = tm_acc.get(k, 0.0) + v
    barrier()
    secs = time.perf_counter() - t0
    if world > 1:
        cdev = dev if a.backend == "nccl" else "cpu"
        t = torch.tensor([secs], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
        cs = torch.tensor([float(sol.ncells) * nt], dtype=torch.float64, device=cdev)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        cellsteps = float(cs.item())
    else:
        cellsteps = float(sol.ncells) * nt
    par_b, sta_b = par.copy(), sta.copy()
    cost = sol.download(adjoint, par, sta, out, par_b, sta_b)
    if world > 1:                                   # global cost = sum of the per-tile partial costs
        ct = torch.tensor([cost], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        cost = float(ct.item())

    if rank == 0 and a.trace_groups:
        ticks, rnd = sol.group_times()
        rep = {}
        for ps, name in enumerate(("forward", "adjoint")):
            t = ticks[ps].astype(np.float64) / 100e3          # ms at 100 MHz
            if not t[:, 0].any():
                continue
            o = t[:, 0][t[:, 0] > 0].min()
            rep[name] = {"span_ms": float(t[:, 1].max() - o), "rounds": [
                {"round": int(r), "groups": int((rnd == r).sum()),
                 "first_start_ms": float(t[rnd == r, 0].min() - o), "last_start_ms": float(t[rnd == r, 0].max() - o),
                 "first_end_ms": float(t[rnd == r, 1].min() - o), "last_end_ms": float(t[rnd == r, 1].max() - o),
                 "mean_run_ms": float((t[rnd == r, 1] - t[rnd == r, 0]).mean())} for r in np.unique(rnd)]}
        with open(a.trace_groups, "w") as f:
            json.dump(rep, f, indent=1)
    if rank == 0:
        K = a.steps
        ms_per_step = secs * 1e3 / K
        value = cellsteps * K / secs
        tm = {k: v / K for k, v in tm_acc.items()}
        # dominant kernel and its roofline: algorithmic bytes = 8 B per cell-step per vertical pass
        # (prcp + pet read once forward, once in the reverse pass: SURVEY 8d, 16 B per forward+adjoint cell-step)
        kern = {"sx_k_vert_fwd": (tm["vert_fwd_ms"], tm["vert_fwd_launches"]),
                "sx_k_vert_adj": (tm["vert_adj_ms"], tm["vert_adj_launches"]),
                "sx_k_route_fwd": (tm["route_fwd_ms"], tm["route_fwd_launches"]),
                "sx_k_route_adj": (tm["route_adj_ms"], tm["route_adj_launches"])}
        dom = max(kern, key=lambda k: kern[k][0])
        dom_ms, dom_n = kern[dom]
        per_launch_steps = float(sol.ncells) * nt / max(tm["n_chunks"], 1)       # cell-steps one launch processes
        if dom.startswith("sx_k_vert"):
            alg_bytes = 8.0 * per_launch_steps
            n_launch = max(dom_n, 1.0)
            avg_ms = dom_ms / n_launch
        else:
            alg_bytes = 4.0 * per_launch_steps       # routing reads qt once; it is not the streaming kernel
            n_launch = max(tm["n_chunks"], 1)
            avg_ms = dom_ms / n_launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        sweep_bytes = (16.0 if adjoint else 8.0) * float(sol.ncells) * nt
        line = {
            "metric": "cell-timesteps/s, forward+adjoint sweep" if adjoint else "cell-timesteps/s, forward sweep",
            "value": value, "unit": "cell-timesteps/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{nrow}x{ncol} synthetic catchment ({trows}x{tcols} cells per GPU, D8 E/SE/S, all cells active), "
                                   f"hourly x {nt} steps, {a.structure}, nse cost at {a.ng} gauges, one forward+adjoint sweep = cost + "
                                   "gradient of all distributed parameters and initial states (BASELINE.json configs[2]; tiled: configs[4])",
                       "grid": [nrow, ncol], "tile": [trows, tcols], "nt": nt, "structure": a.structure, "active_cells": int(cellsteps / nt),
                       "chunk_steps": int(tm["chunk_steps"]), "n_chunks": int(tm["n_chunks"]),
                       "routing_rounds": int(tm["n_rounds"]), "routing_groups": int(tm["n_groups"]),
                       "parallelism": (f"rank {me} of {parts} alone ({a.partition}), no-op exchange" if solo else
                                       (f"tiles {pr}x{pc}" if a.partition == "rect" else f"{parts} {a.partition} parts of the river tree") +
                                       ", exchange of boundary discharge series (send/recv)" if world > 1 else "single")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, per_launch_steps) if a.structure == "gr-b" else None,
                         "avg_launch_ms": avg_ms, "launches_per_step": n_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "sweep_frac": sweep_bytes / (tm["sweep_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "pointwise transcendental-heavy path: the VALU ceiling binds before HBM (DESIGN.md)"},
            "kernel_ms_per_step": {k: round(v[0], 3) for k, v in kern.items()},
            "device_sweep_ms": tm["sweep_ms"], "cost": cost, "setup_s": t_setup,
            "hbm_plan_gb": tm["device_bytes"] / 1e9,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
