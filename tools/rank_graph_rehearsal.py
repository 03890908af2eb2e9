#!/usr/bin/env python3
"""The real 2 x 4 rank graph of the metric's decomposition (BASELINE.json configs[4]) rehearsed in ONE process on ONE GPU at a reduced
tile: eight plans, one thread per tile, the boundary series handed over through in-process queues (as tests/test_gpu_tiles.py does)
instead of RCCL.  Records, per tile, phase (forward receive / send, reverse receive / send) and sub-chunk, when the hook was entered
and how long the tile waited for its neighbour -- the skeleton of the pipeline the N-GPU sweep will have (who waits for whom, in which
order the sub-chunks become available), with one GPU's time shared by the eight tiles.

    python tools/rank_graph_rehearsal.py [--rows 512 --cols 256 --nt 2208 --chunk 1104 --pipe 368] --out profiles/r3_rank_graph_2x4.json
"""
import argparse
import json
import os
import queue
import sys
import threading
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=512)
    ap.add_argument("--cols", type=int, default=256)
    ap.add_argument("--nt", type=int, default=2208)
    ap.add_argument("--chunk", type=int, default=1104)
    ap.add_argument("--pipe", type=int, default=368)
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    import smash_amd
    from smash_amd import synth, tiles
    from smash_amd.solver import Solver
    torch.zeros(1, device="cuda")
    world, pr, pc = 8, 2, 4
    nrow, ncol = pr * a.rows, pc * a.cols
    m = synth.make_mesh(nrow, ncol, ng=8)
    prcp, pet = synth.dense_forcing(m, a.nt)
    P, S = synth.make_parameters(nrow, ncol), synth.make_states(nrow, ncol, warm=True)
    box = {(s, d, k): queue.Queue() for s in range(world) for d in range(world) for k in "fa"}
    t_origin = [0.0]
    log = {r: [] for r in range(world)}
    res = {}

    class Timed:
        def __init__(self, rank, sol):
            self.rank = rank
            self.peers = tiles.PeerLists(sol, nrow, ncol, pr, pc)
            _, self.tp = sol.chunking()
            self.out_buf = torch.zeros(max(self.peers.n_out, 1) * self.tp, dtype=torch.float32, device="cuda")
            self.in_buf = torch.zeros(max(self.peers.n_in, 1) * self.tp, dtype=torch.float32, device="cuda")
            sol.set_halo(self.out_buf.data_ptr(), self.in_buf.data_ptr(), self)

        def __call__(self, phase, t0, nsteps):
            w = 4 * ((nsteps + 3) // 4)
            use_out = phase in (1, 2)
            buf = self.out_buf if use_out else self.in_buf
            n = self.peers.n_out if use_out else self.peers.n_in
            idx = self.peers.out_peers if use_out else self.peers.in_peers
            view = buf[: n * w].view(n, w)
            kind = "f" if phase < 2 else "a"
            t_in = time.perf_counter()
            waited = 0.0
            for p, ix in idx.items():
                ixt = torch.from_numpy(ix).cuda()
                if phase in (1, 3):
                    box[(self.rank, p, kind)].put((t0, view[ixt].clone()))
                else:
                    tq = time.perf_counter()
                    t0_, data = box[(p, self.rank, kind)].get(timeout=300)
                    waited += time.perf_counter() - tq
                    view[ixt] = data
            torch.cuda.current_stream().synchronize()
            log[self.rank].append({"phase": ("fwd_recv", "fwd_send", "adj_recv", "adj_send")[phase], "t0": int(t0), "steps": int(nsteps),
                                   "entered_ms": (t_in - t_origin[0]) * 1e3, "waited_ms": waited * 1e3, "left_ms": (time.perf_counter() - t_origin[0]) * 1e3})
            return 0

    bar = threading.Barrier(world)
    errs = []

    def run(rank):
        try:
            rect = tiles.tile_rect(rank, nrow, ncol, pr, pc)
            gp = np.asarray(m.gauge_pos)
            loc = [i for i in range(m.ng) if rect[0] <= gp[i, 0] < rect[1] and rect[2] <= gp[i, 1] < rect[3]]
            setup = smash_amd.SetupDT(0, len(loc), structure="gr-b", dt=3600.0, ntime_step=a.nt)
            setup.optimize.jobs_fun, setup.optimize.wjobs_fun = ["nse"], [1.0]
            setup.optimize.wgauge = np.full(len(loc), 1.0 / m.ng, np.float32)
            mesh = smash_amd.MeshDT(setup, nrow, ncol, len(loc))
            mesh.dx, mesh.flwdir, mesh.flwacc, mesh.path, mesh.active_cell = m.dx, m.flwdir, m.flwacc, m.path, m.active_cell
            mesh.gauge_pos = np.asfortranarray(gp[loc].reshape(-1, 2)) if loc else np.zeros((0, 2), np.int32, order="F")
            mesh.area = np.asarray(m.area)[loc] if loc else np.zeros(0, np.float32)
            sol = Solver(setup, mesh, chunk_steps=a.chunk, pipe_steps=a.pipe, tile=rect)
            sol.set_forcing(prcp, pet)
            if loc:
                sol.set_qobs(np.zeros((len(loc), a.nt), np.float32, order="F"))
            sol.set_options(setup.optimize)
            Timed(rank, sol)
            par, sta = smash_amd.ParametersDT.from_dict(mesh, P), smash_amd.StatesDT.from_dict(mesh, S)
            sol.upload(par, sta)
            times = []
            for s in range(a.sweeps):
                bar.wait(timeout=600)
                if rank == 0:
                    t_origin[0] = time.perf_counter()
                bar.wait(timeout=600)
                if s == a.sweeps - 1:
                    log[rank].clear()
                ts = time.perf_counter()
                sol.sweep(True, 1.0)
                times.append((time.perf_counter() - ts) * 1e3)
            res[rank] = {"rect": [int(v) for v in rect], "sweep_ms": times, "kernels_last_sweep": {k: v for k, v in sol.timing().items() if k.endswith("_ms")},
                         "edges_out_in": list(sol.halo_counts())}
            bar.wait(timeout=600)
            sol.close()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)
            bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=1200)
    if errs:
        sys.exit(1)
    rep = {"what": __doc__.split("\n\n")[0], "grid": [nrow, ncol], "tiles": [pr, pc], "tile": [a.rows, a.cols], "nt": a.nt, "chunk_steps": a.chunk,
           "pipe_steps": a.pipe, "rank_graph": {str(r): sorted({int(p) for p in tiles.PeerLists.__new__(tiles.PeerLists).__dict__}) for r in range(0)},
           "tiles_report": res,
           "waits": {str(r): log[r] for r in range(world)},
           "summary": {str(r): {"sweep_ms_last": res[r]["sweep_ms"][-1], "waited_ms_total": float(sum(e["waited_ms"] for e in log[r])),
                                "longest_wait": max(log[r], key=lambda e: e["waited_ms"]) if log[r] else None} for r in range(world)}}
    del rep["rank_graph"]
    txt = json.dumps(rep, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)
    print(json.dumps(rep["summary"], indent=1))


if __name__ == "__main__":
    main()
