#!/usr/bin/env python3
"""bench.py -- cell-timesteps/s of one forward + adjoint sweep (the work of one reference forward_b call:
cost + gradient w.r.t. all distributed parameters and initial states) on a synthetic catchment with
hourly x 1 yr forcing resident in HBM (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward+adjoint sweep of the hot path over the whole (grid x nt) batch.
  N = 1   the largest single-GPU configuration of BASELINE.json: the 2048 x 2048 grid of configs[3], 8760 steps, gr-b, compact
          forcing resident, adjoint checkpointed in storage chunks.  The same line carries "secondary" (configs[2]: 1024 x 1024,
          store-all adjoint), "tile_solo" (the 2048 x 1024 tile one GPU owns in the 8-GPU decomposition of configs[4], alone:
          the like-for-like base of the N > 1 lines), "exact_libm" (the bit-exact build) and "cpu_baseline".
  N > 1   one process per GPU, every GPU owns a 2048 x 1024 tile: 1x2, 2x2, 2x4 tiles = 2048x2048, 4096x2048 and, at
          N = 8, the 4096 x 4096 grid the metric is quoted on (configs[4]).  Boundary discharge series move between the
          tiles with grouped ncclSend / ncclRecv posted by the library itself on its routing stream (RCCL over xGMI);
          torch.distributed (backend nccl = RCCL) carries the set-up handshake, the barriers and the max-over-ranks time.
          Under a launcher (torchrun: RANK / WORLD_SIZE in the environment) the process is one rank; without one,
          --gpus N starts the N ranks itself as fresh child processes before anything here touches the GPU.
          --tile-rows / --tile-cols (or --grid) choose another per-GPU tile, e.g. --grid 1024 for the 1024^2-per-GPU series.
Prints ONE JSON line on rank 0.

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
N_SIMD, CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs
VALU_CYCLES = 2.0                # a wave64 fp32 instruction on the 32-lane SIMD when another wave is ready (MI355X_MICROARCH.md "Per-instruction
                                 # cycle constants"; one wave alone issues every 4 cycles)
VALU_CYCLES_F64, VALU_CYCLES_TRANS = 4.0, 8.0    # fp64 at half rate; v_rcp / v_rsq / v_sqrt / v_exp / v_log at quarter rate (same table)
KERNELS = ("vert_fwd", "route_fwd", "route_adj", "vert_adj")


def pmc_profile(grid, n_chunks, forward_only=False):
    """Counters of the committed rocprofv3 --pmc passes over this very command (tools/profile_round.sh): bench.py cannot
    collect PMC itself.  Per kernel: HBM bytes per cell-step (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md)
    and VALU wave-instructions per cell-step (SQ_INSTS_VALU).  The newest profiles/r*_pmc_traffic*.json taken on THIS workload
    (same grid, same number of storage chunks: a checkpointed adjoint moves other bytes than a store-all one) wins; files
    without a workload tag are the 1024 x 1024 store-all case of rounds 1-2."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {"grid": [1024, 1024], "n_chunks": 1})
        if (list(w.get("grid", [])) == list(grid) and int(w.get("n_chunks", 0)) == int(n_chunks)
                and bool(w.get("forward_only", False)) == bool(forward_only)):
            return os.path.relpath(f, ROOT), d
    return None, {}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=0, help="square per-GPU tile (default at N = 1: 2048, the largest single-GPU configuration)")
    ap.add_argument("--tile-rows", type=int, default=0, help="per-GPU tile rows (default at N > 1: 2048 x 1024, so 8 GPUs hold the 4096^2 grid)")
    ap.add_argument("--tile-cols", type=int, default=0)
    ap.add_argument("--pipe", type=int, default=0, help="pipeline sub-chunk (default: none on 1 GPU; tiles: 2192 up to 4 ranks, 1104 at 8)")
    ap.add_argument("--partition", default="rect", choices=["rect", "sub", "trunk"],
                    help="N > 1: rectangles (default), sub-catchments (tiles.partition_subcatchments) or the depth-2 trunk cut (tiles.partition_trunk)")
    ap.add_argument("--trunk-share", type=float, default=1.0, help="--partition trunk: the trunk part's share of the cells relative to 1/N")
    ap.add_argument("--as-rank", type=int, default=-1, help="diagnostics, one process: build rank R's part of an --of N decomposition and time it "
                    "alone (no-op exchange, zero inflow): what that rank computes, without waiting for its neighbours")
    ap.add_argument("--of", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo only for rehearsing the N>1 path without RCCL")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"],
                    help="boundary series: rccl = grouped ncclSend/ncclRecv posted by libsmashx on its routing stream (default); "
                         "torch = host callback + torch.distributed point-to-point (always used with --backend gloo)")
    ap.add_argument("--mesh", default="synth", help="catchment: synth = the E/SE/S field of SURVEY 8d on --grid / the default grid (default); d8 = the "
                    "synthetic all-eight-code relief (synth.make_mesh_d8) on the same grid; france:all / france:K = the reference's 1-km D8 raster "
                    "of France, whole or its K largest basins (data fixture tests/golden/mesh/france_d8.npz; N > 1: --partition sub or trunk)")
    ap.add_argument("--no-real-d8", action="store_true", help="N = 1: skip the real river network (france:all) attached to the line")
    ap.add_argument("--nt", type=int, default=8760)
    ap.add_argument("--structure", default="gr-b")
    ap.add_argument("--chunk", type=int, default=0, help="time-chunk length (0 = from HBM size)")
    ap.add_argument("--group", type=int, default=0, help="routing group size (0 = default)")
    ap.add_argument("--ng", type=int, default=8)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--raw-forcing", action="store_true", help="keep the forcing as fp32 (no lossless compaction)")
    ap.add_argument("--trace-groups", default="", help="diagnostics: write per-round start/end times of the routing groups (JSON) here")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the 1024 x 1024 store-all measurement attached to the line")
    ap.add_argument("--no-tile-solo", action="store_true", help="N = 1: skip the 2048 x 1024 tile of the 8-GPU decomposition timed alone")
    ap.add_argument("--no-inclusive", action="store_true", help="N = 1: skip the upload + sweep + download calls (inclusive_ms_per_step)")
    ap.add_argument("--profile", action="store_true", help="the headline case only, nothing attached and no child process: what "
                    "tools/profile_round.sh runs under rocprofv3")
    ap.add_argument("--no-exact", action="store_true", help="N = 1: skip the runs of the exact-libm build attached to the line")
    ap.add_argument("--no-forward-only", action="store_true", help="N = 1: skip the forward-only case (BASELINE.json configs[1]) attached to the line")
    ap.add_argument("--secondary-grid", type=int, default=1024)
    ap.add_argument("--cpu-grid", type=int, default=256)
    ap.add_argument("--cpu-nt", type=int, default=360)
    ap.add_argument("--cpu-cores", type=int, default=0, help="replicas of the CPU baseline (default: the box's CPU share, at most 16)")
    ap.add_argument("--cpu-solo-grid", type=int, default=512, help="the one-replica CPU measurement: grid (its tape leaves every cache level)")
    ap.add_argument("--cpu-solo-nt", type=int, default=160)
    a = ap.parse_args(argv)
    if a.profile:
        a.no_secondary = a.no_tile_solo = a.no_exact = a.no_cpu_baseline = a.no_inclusive = a.no_forward_only = a.no_real_d8 = True
        if a.gpus != 1:
            ap.error("--profile is a single-process run")
    return a


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


def _cpu_replicas(structure, n, nt, cores):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                       # no fork of a process that holds a HIP context
    barrier, q = ctx.Barrier(cores), ctx.Queue()
    procs = [ctx.Process(target=_cpu_replica, args=(structure, n, nt, barrier, q)) for _ in range(cores)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join()
    return res


def cpu_baseline(structure, n, nt, cores=0, solo_n=512, solo_nt=160):
    """The reference path timed on the host cores of this box, on bounded samples of the same synthetic workload:
    forward_b (cost + gradient) on an n x n catchment over nt steps.  forward_b is single-threaded
    (mw_forward.f90:41-68; no OpenMP in it), so SURVEY.md 8d asks for both ways of using the host:
      (ii) filled the way the reference fills it (mw_multiple_run.f90:96-107): one independent replica per core, started
           together; value = all replicas' work / the slowest replica's time (the line's `value`);
      (i)  `solo_core`: ONE replica on an otherwise idle box share, on a window whose tape (40 B per cell-step, adStack.c)
           is far beyond the last-level cache."""
    # the CPU share of a one-GPU box is 16 cores whatever the affinity mask says; a replica holds ~2 GB (the reference's
    # tape is 40 B per cell-step), so the pool is sized to that share and never to the machine
    cores = cores or max(1, min(len(os.sched_getaffinity(0)), 16))
    res = _cpu_replicas(structure, n, nt, cores)
    kind = res[0][0]
    work, slow = sum(r[1] for r in res), max(r[2] for r in res)
    what = ("reference Fortran (flang -O3, libsmash_ref_fast.so) forward_b" if kind == "reference"
            else "plain-C oracle (gcc -O2) forward_b")
    out = {"value": work / slow, "unit": "cell-timesteps/s", "cores": cores, "kind": kind,
           "per_core_value": float(np.mean([r[1] / r[2] for r in res])),
           "sample": f"{what}, {structure}, {cores} independent replicas (one per core) of a {n}x{n} synthetic catchment x {nt} "
                     f"hourly steps, slowest replica {slow:.1f} s"}
    if solo_n > 0 and solo_nt > 0:
        try:
            k1, w1, s1 = _cpu_replicas(structure, solo_n, solo_nt, 1)[0]
            out["solo_core"] = {"value": w1 / s1, "unit": "cell-timesteps/s", "cores": 1, "kind": k1,
                                "sample": f"{what}, {structure}, ONE replica alone on the box share: {solo_n}x{solo_n} cells x {solo_nt} hourly "
                                          f"steps ({w1 * 40e-9:.1f} GB of adStack tape, {w1 * 8e-9:.2f} GB of forcing: nothing stays in cache), {s1:.1f} s"}
        except Exception as e:  # pragma: no cover
            out["solo_core"] = {"value": None, "error": str(e)}
    return out


def self_launch(a, argv):
    """--gpus N without a launcher: start the N ranks as fresh children -- nothing in this parent has imported torch or
    touched the GPU, and no process that has is ever re-executed.  Rank 0's stdout (the JSON line) is ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad and t_fail is None:
            rc, t_fail = bad[0], time.time()
        if t_fail is not None and time.time() - t_fail > 20:      # a rank died: the others would wait for it forever
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = rc or p.returncode
    return rc


def chunk_plan(nt, cells, hbm_bytes, forcing_bytes, structure, forced=0):
    """Storage-chunk length for a tile, from quantities every rank agrees on (the GPU's total HBM, not this rank's free
    memory): the ranks of a decomposition must cut time identically."""
    nt16 = (nt + 15) // 16 * 16
    if forced:
        return forced
    per_step = 4.0 * cells * ({"gr-a": 4, "gr-b": 5, "gr-c": 6, "gr-d": 4, "vic-a": 5}[structure])   # qt, hr_imd + taped levels
    avail = 0.80 * hbm_bytes - forcing_bytes - 2.0e9
    tc = int(max(16, min(nt16, int(avail / per_step) // 16 * 16)))
    nch = (nt + tc - 1) // tc
    return min(tc, ((nt + nch - 1) // nch + 15) // 16 * 16)


class Case:
    """One benchmark workload: plan + forcing resident in HBM + observations, ready to sweep."""

    def __init__(self, a, torch, dev, local, parts, me, world, trows, tcols, solo, raw_forcing, adjoint=None):
        import smash_amd
        from smash_amd import synth, tiles
        from smash_amd.solver import Solver
        self.a, self.torch, self.dev = a, torch, dev
        nt = a.nt
        pr, pc = tiles.tile_grid(parts)
        nrow, ncol = pr * trows, pc * tcols            # the whole catchment; every rank owns 1/parts of it
        self.pr, self.pc, self.nrow, self.ncol, self.parts, self.me = pr, pc, nrow, ncol, parts, me
        t_setup = time.perf_counter()
        mesh_kind = getattr(a, "mesh", "synth")
        if mesh_kind.startswith("france"):
            w = mesh_kind.split(":", 1)[1] if ":" in mesh_kind else "all"
            m = synth.make_mesh_france(w if w == "all" else int(w), ng=a.ng)
            if parts > 1 and a.partition == "rect":
                raise SystemExit("bench.py: a real D8 network cannot be cut into rectangles (cyclic rank graph): --partition sub or trunk")
            nrow, ncol = m.nrow, m.ncol
            self.nrow, self.ncol = nrow, ncol
        elif mesh_kind == "d8":
            m = synth.make_mesh_d8(nrow, ncol, ng=a.ng)
        else:
            m = synth.make_mesh(nrow, ncol, ng=a.ng)
        self.mesh_kind = mesh_kind
        rect, owner, mine = None, None, None
        if parts > 1 and a.partition == "rect":
            rect = tiles.tile_rect(me, nrow, ncol, pr, pc)
        elif parts > 1:                                # every rank computes the same cut of the river tree
            owner = tiles.partition_trunk(m, parts, a.trunk_share) if a.partition == "trunk" else tiles.partition_subcatchments(m, parts)
            mine = np.asfortranarray((np.asarray(owner) == me).astype(np.int32))
        self.owner = owner
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
        self.setup, self.mesh, self.loc = setup, mesh, loc
        chunk, pipe = a.chunk, a.pipe
        compact = not raw_forcing
        if parts > 1:
            # every rank must cut time identically (messages are per sub-chunk): lengths from sizes all ranks share
            cells = -(-nrow * ncol // parts)
            hbm = torch.cuda.get_device_properties(dev).total_memory
            fbytes = cells * nt * (2.0 + 4.0 / 24.0 if compact else 8.0)
            chunk = chunk_plan(nt, cells, hbm, fbytes, a.structure, chunk)
            # sub-chunk length: a pass over n sub-chunks through a rank graph of depth L costs (L - 1 + n) (a / n + b), b = the fill
            # of the chained routing rounds paid per launch: 8 sub-chunks per year at depth 5 (2 x 4 tiles), 4 at
            # depth <= 3 (solo rank, measured: 364 ms against 380)
            pipe = pipe or (2192 if parts <= 4 else 1104)
        if a.trace_groups:
            os.environ["SMASHX_TRACE_GROUPS"] = "1"
        sol = Solver(setup, mesh, chunk_steps=chunk, pipe_steps=pipe, group_size=a.group, device=local, tile=rect, owner_mask=mine)
        self.sol = sol
        if compact:
            daily, w = synth._pet_tables()
            sol.set_forcing_layout(compact=True, prcp_factor=0.1, pet_ratio=w, pet_hour0=0)
        rows, cols = sol.cell_order()
        d_rows = torch.from_numpy(rows.astype(np.int64)).to(dev)
        d_cols = torch.from_numpy(cols.astype(np.int64)).to(dev)
        tb = max(24, (1 << 26) // max(sol.ncells, 1) // 24 * 24)     # whole days; ~64 M cell-steps of int64 temporaries per block
        for t0 in range(0, nt, tb):
            t1 = min(nt, t0 + tb)
            prcp, pet = synth.forcing_block(d_rows, d_cols, t0, t1, xp=torch, device=dev)
            torch.cuda.synchronize()
            sol.set_forcing_device_block(t0, t1, prcp.data_ptr(), pet.data_ptr())
            del prcp, pet
        del d_rows, d_cols
        torch.cuda.empty_cache()
        self.forcing = sol.forcing_info()

        P, S = synth.make_parameters(nrow, ncol), synth.make_states(nrow, ncol, warm=True)
        self.par, self.sta = smash_amd.ParametersDT.from_dict(mesh, P), smash_amd.StatesDT.from_dict(mesh, S)
        self.out = smash_amd.OutputDT(setup, mesh)
        # observations = forward run with parameters perturbed by +10 %
        parq = smash_amd.ParametersDT.from_dict(mesh, synth.make_parameters(nrow, ncol, perturb=0.1))
        sol.set_options(setup.optimize)
        self.comm, self.exchange = None, None
        if world > 1:
            import torch.distributed as dist
            if a.exchange == "rccl" and a.backend == "nccl":
                from smash_amd.solver import Comm
                uid = [Comm.unique_id() if me == 0 else None]
                dist.broadcast_object_list(uid, src=0, device=dev)
                self.comm = Comm(uid[0], me, world, local)
                self.exchange = tiles.RcclExchange(sol, self.comm, nrow, ncol, pr, pc, owner)
            else:
                self.exchange = tiles.TorchDistExchange(sol, nrow, ncol, pr, pc, dev, owner)
        elif solo:
            self.exchange = tiles.NoExchange(sol, dev)
        sol.upload(parq, self.sta)
        sol.sweep(False)
        sol.download(False, parq, self.sta, self.out)
        if len(loc):
            sol.set_qobs(self.out.qsim)
        sol.upload(self.par, self.sta)
        if (not a.forward_only) if adjoint is None else adjoint:
            sol.chunking()                               # allocates the tapes of the reverse sweep now (set-up), not in the first sweep
        self.hbm = sol.hbm()
        self.setup_s = time.perf_counter() - t_setup

    def close(self):
        self.sol.close()
        if self.comm is not None:
            self.comm.close()
        self.torch.cuda.empty_cache()


def timed_sweeps(case, steps, warmup, adjoint, barrier):
    """W untimed sweeps, then exactly K sweeps between two barrier + synchronize brackets.  Returns (seconds, kernel times)."""
    sol = case.sol
    for _ in range(warmup):
        sol.sweep(adjoint)
    tm_acc = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sol.sweep(adjoint)          # smashx_sweep synchronises its streams before returning
        for k, v in sol.timing().items():
            tm_acc[k] = tm_acc.get(k, 0.0) + v
    barrier()
    secs = time.perf_counter() - t0
    return secs, {k: v / steps for k, v in tm_acc.items()}


def inclusive_call_ms(case, adjoint, reps=3):
    """One call through the boundary the way the reference's host makes it: parameters and states uploaded from host arrays,
    the sweep, cost + discharge + all gradient planes downloaded (smashx_forward_b).  Not the metric's value (inputs resident)."""
    sol = case.sol
    par_b, sta_b = case.par.copy(), case.sta.copy()
    case.torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sol.upload(case.par, case.sta)
        sol.sweep(adjoint)
        sol.download(adjoint, case.par, case.sta, case.out, par_b, sta_b)
    case.torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def roofline(tm, adjoint, structure, grid):
    """Dominant kernel against both ceilings: HBM (algorithmic bytes: prcp + pet = 8 B per cell-step per vertical pass,
    SURVEY.md 8d) and VALU issue (wave-instructions per cell-step from the committed PMC pass of THIS workload, priced at
    2 cycles per wave64 fp32 instruction on 1024 SIMDs and -- `weighted` -- at the 2 / 4 / 8-cycle mix of the kernel's loop)."""
    ms = {k: tm[k + "_ms"] for k in KERNELS}
    dom = max(ms, key=ms.get)
    n_launch = max(tm[dom + "_launches"], 1.0)
    avg_ms = ms[dom] / n_launch
    cs_launch = tm[dom + "_cellsteps"] / n_launch                 # cell-steps ONE launch processes (sub-chunked launches are shorter)
    per_cs = 8.0 if dom.startswith("vert") else 4.0               # routing reads qt once: it is not a streaming kernel of the forcing
    alg_bytes = per_cs * cs_launch
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    src, prof = pmc_profile(grid, tm["n_chunks"], not adjoint)
    # (a forward-only sweep runs the untaped forward kernels: their own family in the PMC summary)
    pk = prof.get("sx_k_" + dom + ("_untaped" if not adjoint and dom.endswith("_fwd") else "")) if structure == "gr-b" else None
    traffic = pk["hbm_bytes_per_cellstep_corrected"] * cs_launch if pk and "hbm_bytes_per_cellstep_corrected" in pk else None
    r = {"bound": "hbm", "kernel": "sx_k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
         "traffic_source": (f"{src}: rocprofv3 --pmc passes of this command (FETCH_SIZE x2 + WRITE_SIZE over every launch of the kernel, "
                            "per cell-step), times the cell-steps of one launch" if traffic else None),
         "avg_launch_ms": avg_ms, "launches_per_step": n_launch, "cellsteps_per_launch": cs_launch,
         "algorithmic_bytes_per_launch": alg_bytes,
         "sweep_frac": (16.0 if adjoint else 8.0) * tm["vert_adj_cellsteps" if adjoint else "vert_fwd_cellsteps"] / (tm["sweep_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if pk and "valu_per_cellstep" in pk:
        waves = pk["valu_per_cellstep"] * cs_launch / 64.0
        issue_ms = waves * VALU_CYCLES / (N_SIMD * CLOCK_HZ) * 1e3
        r["valu"] = {"instr_per_cellstep": pk["valu_per_cellstep"], "issue_bound_ms": issue_ms, "frac": issue_ms / avg_ms,
                     "wave_cycles_share": pk.get("wave_cycles_share"),
                     "note": "SQ_INSTS_VALU per cell-step (committed PMC pass) x 2 cycles per wave64 fp32 instruction on 1024 SIMDs at 2.4 GHz "
                             "= the time the vector units need at their fp32 peak rate; frac = that time / the launch time"}
        mix = pk.get("valu_mix_per_cellstep")
        if mix:       # dynamic mix from the SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 / _TRANS_F32 / _TRANS_F64 counters of the same profile
            n64, ntr = mix.get("f64", 0.0), mix.get("trans", 0.0)
            n32 = max(pk["valu_per_cellstep"] - n64 - ntr, 0.0)
            cyc = (n32 * VALU_CYCLES + n64 * VALU_CYCLES_F64 + ntr * VALU_CYCLES_TRANS) / pk["valu_per_cellstep"]
            w_ms = waves * cyc / (N_SIMD * CLOCK_HZ) * 1e3
            r["valu"]["weighted"] = {"issue_bound_ms": w_ms, "frac": w_ms / avg_ms, "cycles_per_instr": cyc,
                                     "mix_per_cellstep": {"fp32_and_int": n32, "fp64": n64, "transcendental": ntr},
                                     "note": "the same count priced at the executed mix (PMC): fp32 / integer 2, fp64 add / mul / fma 4, "
                                             "transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log) 8 cycles per wave64 instruction "
                                             "(MI355X_MICROARCH.md, per-instruction cycle constants): the ceiling this kernel runs against"}
    return r


def attached_case(a, torch, dev, local, barrier, adjoint, grid_rc, parts, me, solo, steps, warmup, what):
    """One more workload measured in this process after the headline case has been closed: returns the object attached to the line."""
    try:
        c = Case(a, torch, dev, local, parts, me, 1, grid_rc[0], grid_rc[1], solo, False, adjoint)
        secs, tm = timed_sweeps(c, steps, warmup, adjoint, barrier)
        o = {"workload": what, "value": float(c.sol.ncells) * a.nt * steps / secs, "unit": "cell-timesteps/s", "steps": steps, "warmup": warmup,
             "ms_per_step": secs * 1e3 / steps, "grid": [c.nrow, c.ncol], "tile": list(grid_rc), "active_cells": int(c.sol.ncells),
             "n_chunks": int(tm["n_chunks"]), "chunk_steps": int(tm["chunk_steps"]), "pipe_steps": int(tm["pipe_steps"]),
             "routing_rounds": int(tm["n_rounds"]), "routing_groups": int(tm["n_groups"]), "deepest_group_stages": int(tm["max_stage"]),
             "chained_groups": int(tm["n_chained_groups"]), "chained_groups_staged": bool(tm["chain_staged"]),
             "chained_launch_ms_per_step": {"forward": round(tm["route_fwd_chained_ms"], 3), "reverse": round(tm["route_adj_chained_ms"], 3),
                                            "launches": int(tm["route_fwd_chained_launches"] + tm["route_adj_chained_launches"])},
             "hbm_plan_gb": tm["device_bytes"] / 1e9,
             "hbm_free_at_plan_gb": c.hbm["free_at_plan"] / 1e9, "forcing": c.forcing,
             "kernel_ms_per_step": {"sx_k_" + k: round(tm[k + "_ms"], 3) for k in KERNELS},
             "kernel_launches_per_step": {"sx_k_" + k: int(tm[k + "_launches"]) for k in KERNELS},
             "roofline": roofline(tm, adjoint, a.structure, [c.nrow, c.ncol]) if parts == 1 else None, "setup_s": c.setup_s}
        c.close()
        return o
    except Exception as e:  # pragma: no cover
        return {"workload": what, "value": None, "error": str(e)}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and a.as_rank < 0:
        sys.exit(self_launch(a, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the CPU baseline starts one process per core: it runs FIRST, before anything in this process touches the GPU
    # (rank 0 at N = 1 only)
    cpu = None
    if not a.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(a.structure, a.cpu_grid, a.cpu_nt, a.cpu_cores, a.cpu_solo_grid, a.cpu_solo_nt)
        except Exception as e:  # pragma: no cover
            cpu = {"value": None, "unit": "cell-timesteps/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    import torch
    import torch.distributed as dist
    from smash_amd import tiles

    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)       # counting devices does not initialise the GPU
    shared = world > 1 and ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if shared and a.backend == "nccl":
        tiles.share_one_gpu_env(rank)              # rehearsal on a box with fewer GPUs than ranks
    local = local % ndev
    if world > 1 and a.backend != "nccl":          # host-side rendezvous first: it needs no device
        dist.init_process_group(a.backend)
        dist.barrier()
        print(f"bench.py: rank {rank}/{world} handshake ok ({a.backend})", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        print("bench.py: no HIP device -- libsmashx has no CPU path, nothing to measure", file=sys.stderr, flush=True)
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()                             # every rank has completed the RCCL handshake past this point
        print(f"bench.py: rank {rank}/{world} handshake ok (nccl)", file=sys.stderr, flush=True)

    solo = a.as_rank >= 0 and a.of > 1 and world == 1      # diagnostics: one rank of a decomposition, alone
    parts, me = (a.of, a.as_rank) if solo else (world, rank)
    if a.tile_rows or a.tile_cols:
        trows, tcols = a.tile_rows or a.tile_cols, a.tile_cols or a.tile_rows
    elif a.grid:
        trows = tcols = a.grid
    else:
        # N = 1: the largest configuration of BASELINE.json that one GPU holds (configs[3]'s 2048^2 grid: compact forcing +
        # checkpointed adjoint, 258 of the 288 GB); N > 1: the 2048 x 1024 tile of configs[4]'s 2 x 4 decomposition
        trows, tcols = (2048, 2048) if parts == 1 else (2048, 1024)
    adjoint = not a.forward_only
    nt = a.nt

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    case = Case(a, torch, dev, local, parts, me, world, trows, tcols, solo, a.raw_forcing)
    sol = case.sol
    secs, tm = timed_sweeps(case, a.steps, a.warmup, adjoint, barrier)
    cdev = dev if a.backend == "nccl" else "cpu"
    if world > 1:
        t = torch.tensor([secs], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
        cs = torch.tensor([float(sol.ncells) * nt], dtype=torch.float64, device=cdev)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        cellsteps = float(cs.item())
    else:
        cellsteps = float(sol.ncells) * nt
    par_b, sta_b = case.par.copy(), case.sta.copy()
    cost = sol.download(adjoint, case.par, case.sta, case.out, par_b, sta_b)
    if world > 1:                                   # global cost = sum of the per-tile partial costs
        if case.comm is not None:
            cost = float(case.comm.allreduce_sum([cost])[0])
        else:
            ct = torch.tensor([cost], dtype=torch.float64, device=cdev)
            dist.all_reduce(ct, op=dist.ReduceOp.SUM)
            cost = float(ct.item())
    inclusive = inclusive_call_ms(case, adjoint) if world == 1 and not solo and not a.no_inclusive else None

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

    line = None
    if rank == 0:
        K = a.steps
        pr, pc, nrow, ncol = case.pr, case.pc, case.nrow, case.ncol
        xch = ("grouped ncclSend/ncclRecv on the routing stream (libsmashx, RCCL)" if case.comm is not None
               else f"host callback + torch.distributed point-to-point ({a.backend})")
        line = {
            "metric": "cell-timesteps/s, forward+adjoint sweep" if adjoint else "cell-timesteps/s, forward sweep",
            "value": cellsteps * K / secs, "unit": "cell-timesteps/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": secs * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{nrow}x{ncol} synthetic catchment ({trows}x{tcols} cells per GPU, D8 E/SE/S, all cells active), "
                                   f"hourly x {nt} steps, {a.structure}, nse cost at {a.ng} gauges, "
                                   + ("one forward+adjoint sweep = cost + gradient of all distributed parameters and initial states "
                                      if adjoint else "one FORWARD sweep = discharge at the gauges + cost (the work of mw_forward::forward) ")
                                   + ("(the grid of BASELINE.json configs[3], the largest configuration one GPU holds: compact forcing resident, "
                                      "adjoint checkpointed in storage chunks)" if world == 1 and (nrow, ncol) == (2048, 2048) else
                                      "(BASELINE.json configs[2])" if world == 1 and (nrow, ncol) == (1024, 1024) else
                                      "" if world == 1 else
                                      "(BASELINE.json configs[4]: the metric's 4096^2 grid)" if (nrow, ncol) == (4096, 4096) else
                                      "(the per-GPU tile of BASELINE.json configs[4], fewer tiles)"),
                       "grid": [nrow, ncol], "tile": [trows, tcols], "nt": nt, "structure": a.structure, "active_cells": int(cellsteps / nt),
                       "chunk_steps": int(tm["chunk_steps"]), "n_chunks": int(tm["n_chunks"]), "pipe_steps": int(tm["pipe_steps"]),
                       "routing_rounds": int(tm["n_rounds"]), "routing_groups": int(tm["n_groups"]), "deepest_group_stages": int(tm["max_stage"]),
                       "chained_groups": int(tm["n_chained_groups"]), "chained_groups_staged": bool(tm["chain_staged"]), "mesh": a.mesh,
                       "forcing": case.forcing,
                       "parallelism": (f"rank {me} of {parts} alone ({a.partition}), no-op exchange" if solo else
                                       ((f"tiles {pr}x{pc}" if a.partition == "rect" else f"{parts} {a.partition} parts of the river tree") +
                                        f", boundary discharge series by {xch}" + (", ranks share GPUs (rehearsal)" if shared else ""))
                                       if world > 1 else "single")},
            "roofline": roofline(tm, adjoint, a.structure, [nrow, ncol]),
            "kernel_ms_per_step": {"sx_k_" + k: round(tm[k + "_ms"], 3) for k in KERNELS},
            "kernel_launches_per_step": {"sx_k_" + k: int(tm[k + "_launches"]) for k in KERNELS},
            "kernel_cellsteps_per_step": {"sx_k_" + k: tm[k + "_cellsteps"] for k in KERNELS},
            "chained_launch_ms_per_step": {"forward": round(tm["route_fwd_chained_ms"], 3), "reverse": round(tm["route_adj_chained_ms"], 3),
                                           "launches": int(tm["route_fwd_chained_launches"] + tm["route_adj_chained_launches"])},
            "device_sweep_ms": tm["sweep_ms"], "cost": cost, "setup_s": case.setup_s,
            "hbm_plan_gb": tm["device_bytes"] / 1e9,
            # free HBM when the plan chose its storage-chunk length (the forcing was resident by then): the chunk count -- and with it the
            # number of forward passes of the checkpointed adjoint -- follows from this figure, not from the card's nominal size
            "hbm_free_at_plan_gb": case.hbm["free_at_plan"] / 1e9, "hbm_total_gb": case.hbm["total"] / 1e9,
        }
        if a.profile:
            # what tools/pmc_summary.py divides the counters by: every sweep this process ran, and the cell-steps each kernel
            # family processed in them (a storage chunk but the last is swept forward twice: untaped, then taped for the reverse sweep)
            C_, Tc_ = int(tm["n_chunks"]), int(tm["chunk_steps"])
            ncell = cellsteps / nt
            untaped = ncell * (C_ - 1) * Tc_ if adjoint else cellsteps
            line["profile_accounting"] = {
                "adjoint_sweeps": (a.warmup + K) if adjoint else 0, "forward_sweeps": 1 + (0 if adjoint else a.warmup + K),
                "cellsteps": cellsteps, "grid": [nrow, ncol], "n_chunks": C_, "chunk_steps": Tc_,
                "per_adjoint_sweep": {"taped_forward": cellsteps, "untaped_forward": untaped, "reverse": cellsteps},
                "per_forward_sweep": {"untaped_forward": cellsteps}}
        if inclusive is not None:
            line["inclusive_ms_per_step"] = inclusive
            line["inclusive_note"] = ("one smashx_forward_b-style call: host parameter/state planes uploaded, sweep, cost + discharge + every "
                                      "gradient plane downloaded; `value` is the resident-input rate (forcing upload is one-off set-up)")
        if cpu is not None:
            line["cpu_baseline"] = cpu
    # N = 1, default workload: two more cases measured on the same GPU and attached to the line
    default_case = (world == 1 and not solo and not a.raw_forcing and (trows, tcols) == (2048, 2048) and nt == 8760 and line is not None
                    and a.mesh == "synth")
    if default_case and not (a.no_secondary and a.no_tile_solo and a.no_forward_only and a.no_real_d8):
        case.close()
        del case, sol
        if not a.no_secondary and a.secondary_grid > 0:
            g = a.secondary_grid
            line["secondary"] = attached_case(
                a, torch, dev, local, barrier, adjoint, (g, g), 1, 0, False, min(a.steps, 5), 1,
                f"{g}x{g} synthetic catchment (BASELINE.json configs[2]), hourly x {nt} steps, {a.structure}, one forward+adjoint sweep; "
                "store-all adjoint (one storage chunk, nothing recomputed)")
        if not a.no_tile_solo:
            o = attached_case(
                a, torch, dev, local, barrier, adjoint, (2048, 1024), 8, 0, True, min(a.steps, 5), 1,
                "rank 0's 2048x1024 tile of the 2x4 decomposition of the 4096x4096 grid (BASELINE.json configs[4]), alone on this GPU: "
                "no-op exchange, zero inflow, the chunking and sub-chunk pipeline of the 8-rank run -- the per-GPU work of every "
                "`--gpus N` line, so value(N) / (N x this value) is the price of the decomposition's pipeline and exchange")
            o["per_gpu_value"] = o.get("value")
            line["tile_solo"] = o
        if not a.no_real_d8 and adjoint:
            # a REAL river network (all eight D8 codes) beside the E/SE/S field: the reference's 1-km raster of France, every cell with a
            # direction active (957 k cells, a forest of basins), and the synthetic all-eight-code relief on 1024^2
            ra = argparse.Namespace(**vars(a))
            ra.mesh = "france:all"
            o = attached_case(
                ra, torch, dev, local, barrier, adjoint, (1125, 1200), 1, 0, False, min(a.steps, 5), 1,
                f"the reference's 1-km D8 flow-direction raster of France (smash/dataset/France_flwdir.tif as data fixture tests/golden/mesh/france_d8.npz; all "
                f"eight codes, every cell with a direction active, closed loops masked), hourly x {nt} steps, {a.structure}, one forward+adjoint sweep; "
                "store-all adjoint")
            ra.mesh = "d8"
            o["synthetic_d8_1024"] = attached_case(
                ra, torch, dev, local, barrier, adjoint, (1024, 1024), 1, 0, False, min(a.steps, 3), 1,
                "synthetic all-eight-code relief (synth.make_mesh_d8: one basin, interior outlet) on 1024x1024 cells, same forcing and parameters")
            line["real_d8"] = o
        if not a.no_forward_only and adjoint:
            g = a.secondary_grid or 1024
            o = attached_case(
                a, torch, dev, local, barrier, False, (g, g), 1, 0, False, min(a.steps, 5), 1,
                f"{g}x{g} synthetic catchment, hourly x {nt} steps, {a.structure}, ONE FORWARD sweep = discharge at the gauges + cost "
                "(BASELINE.json configs[1]; the work of the reference's mw_forward::forward, mw_forward.f90:18-39); compact forcing resident, "
                "no tape; roofline: 8 algorithmic bytes per cell-step (prcp + pet once, SURVEY.md 8d)")
            o["metric"] = "cell-timesteps/s, forward sweep"
            line["forward_only"] = o
    # N = 1: the same workload on the exact-libm build (libsmashx_exact.so: bit-identical to the reference on every golden vector),
    # in a child process -- the library is chosen when smash_amd is imported
    if (default_case and not a.no_exact
            and os.environ.get("SMASHX_EXACT_LIBM", "0") in ("", "0") and os.path.exists(os.path.join(ROOT, "smash_amd", "libsmashx_exact.so"))):
        import subprocess
        if "case" in locals():                       # the child needs the HBM this process still holds
            case.close()
            del case, sol
        torch.cuda.empty_cache()
        def exact_child(grid, steps):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "1", "--profile", "--grid", str(grid),
                                "--structure", a.structure], env=dict(os.environ, SMASHX_EXACT_LIBM="1"), capture_output=True,
                               text=True, timeout=600)
            e = json.loads(r.stdout.strip().splitlines()[-1])
            return {"grid": [grid, grid], "value": e["value"], "ms_per_step": e["ms_per_step"], "steps": steps, "warmup": 1, "cost": e["cost"],
                    "n_chunks": e["config"]["n_chunks"], "kernel_ms_per_step": e["kernel_ms_per_step"]}
        try:
            e = exact_child(1024, 2)
            line["exact_libm"] = dict(e, what="libsmashx_exact.so (-DSX_EXACT_LIBM=1: glibc 2.35 expf/logf/powf/tanhf restated, IEEE divisions): "
                                              "bit-identical to the reference Fortran on all 318 golden outputs (profiles/r3_parity_exact.md); the "
                                              "default build differs from it by libm rounding only (profiles/r3_parity_default.md).  Top level: the "
                                              "1024x1024 workload of `secondary`; `headline`: the 2048x2048 workload of this line's `value`")
            h = exact_child(2048, max(3, min(a.steps, 5)))
            h["slowdown_vs_default"] = h["ms_per_step"] / line["ms_per_step"]
            line["exact_libm"]["headline"] = h
            if isinstance(line.get("secondary"), dict) and line["secondary"].get("ms_per_step"):
                line["exact_libm"]["slowdown_vs_default"] = e["ms_per_step"] / line["secondary"]["ms_per_step"]
        except Exception as ex:  # pragma: no cover
            line.setdefault("exact_libm", {"value": None})["error"] = str(ex)
    if world > 1 and rank == 0 and line is not None:
        if case.comm is not None:
            # what the communicator itself says: ranks it spans (ncclCommCount) and RCCL's version code (ncclGetVersion)
            line["rccl"] = case.comm.info()
        if not a.no_tile_solo and a.partition == "rect":
            # rank 0's own tile once more, ALONE (no-op exchange, zero inflow), after the timed region: the like-for-like base of this
            # line measured on this very GPU, so the run reports its own efficiency; the other ranks wait at the final barrier
            case.sol.close()
            torch.cuda.empty_cache()
            o = attached_case(a, torch, dev, local, torch.cuda.synchronize, adjoint, (trows, tcols), world, 0, True, min(a.steps, 5), 1,
                              f"rank 0's {trows}x{tcols} tile of this decomposition alone on its GPU (no-op exchange, zero inflow, same chunking and "
                              "sub-chunks), measured after the timed region")
            line["tile_solo"] = o
            if o.get("value"):
                line["efficiency_vs_solo_tile"] = line["value"] / (world * o["value"])
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        if 'case' in locals():
            case.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
