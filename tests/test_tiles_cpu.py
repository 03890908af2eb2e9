"""CPU: host logic of the multi-GPU tile decomposition -- no kernel runs here.

* the per-tile routing schedules partition the active cells, and every series that leaves a tile is expected,
  in the same order, by exactly one other tile (checked in-process for 1x2, 2x2 and 2x4 tilings);
* the same agreement is checked by TWO PROCESSES over torch.distributed (gloo, world_size 2): each rank probes
  its own tile and the ranks compare their edge lists through send/recv, the way the RCCL exchange pairs them.
"""
import ctypes as C
import os

import numpy as np
import pytest

from smash_amd import _lib, synth, tiles
from smash_amd.solver import make_config
import smash_amd


def probe(m, rect, group=128, owner_mask=None):
    setup = smash_amd.SetupDT(0, 0, structure="gr-b", ntime_step=16)
    mesh = smash_amd.MeshDT.from_synth(setup, m)
    mesh.ng = 0
    cfg = make_config(setup, mesh, group_size=group, tile=rect)
    cfg.ng = 0
    keep = [np.asfortranarray(m.flwdir, np.int32), np.asfortranarray(m.flwacc, np.int32), np.asfortranarray(m.active_cell, np.int32)]
    cm = _lib.Mesh(keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, None, None, None)
    if owner_mask is not None:
        keep.append(np.asfortranarray(owner_mask, np.int32))
        cm.owner_mask = keep[-1].ctypes.data
    info = (C.c_int * 8)()
    cap = 4 * (m.nrow + m.ncol) + 16
    arr = [np.zeros(cap, np.int32) for _ in range(4)]
    _lib.check(_lib.lib().smashx_tile_probe(C.byref(cfg), C.byref(cm), info, *[a.ctypes.data_as(C.c_void_p) for a in arr], cap))
    no, ni = info[6], info[7]
    return dict(cells=info[0], rounds=info[1], groups=info[2], n_out=no, n_in=ni,
                out_src=arr[0][:no].copy(), out_dst=arr[1][:no].copy(), in_src=arr[2][:ni].copy(), in_dst=arr[3][:ni].copy())


@pytest.mark.parametrize("world,mask", [(2, False), (4, False), (8, False), (4, True)])
def test_tiles_partition_cells_and_agree_on_boundary_series(world, mask):
    pr, pc = tiles.tile_grid(world)
    m = synth.make_mesh(48, 64, ng=1, mask_corner=mask)
    rects = [tiles.tile_rect(r, m.nrow, m.ncol, pr, pc) for r in range(world)]
    P = [probe(m, rect) for rect in rects]
    assert sum(p["cells"] for p in P) == m.nac
    whole = probe(m, None)
    assert whole["cells"] == m.nac and whole["n_out"] == 0 and whole["n_in"] == 0
    for a in range(world):
        own = tiles.owner_of(P[a]["out_dst"], m.nrow, m.ncol, pr, pc) if P[a]["n_out"] else np.zeros(0, int)
        assert not np.any(own == a)
        for b in np.unique(own):
            sel = own == b
            src_b = tiles.owner_of(P[b]["in_src"], m.nrow, m.ncol, pr, pc) == a
            assert np.array_equal(P[a]["out_src"][sel], P[b]["in_src"][src_b])       # same edges, same order
            assert np.array_equal(P[a]["out_dst"][sel], P[b]["in_dst"][src_b])
    assert sum(p["n_out"] for p in P) == sum(p["n_in"] for p in P) > 0
    # E/SE/S drainage: series only ever go to a tile with a larger (row, col) block index -> acyclic tile graph
    for a in range(world):
        if P[a]["n_out"]:
            assert np.all(tiles.owner_of(P[a]["out_dst"], m.nrow, m.ncol, pr, pc) > a)


@pytest.mark.parametrize("kind,world,cut", [("d8", 3, "sub"), ("d8", 5, "sub"), ("ese", 4, "sub"), ("d8", 8, "sub"),
                                             ("d8", 4, "trunk"), ("ese", 8, "trunk")])
def test_subcatchment_partition_is_balanced_acyclic_and_agrees_on_boundary_series(kind, world, cut):
    """Arbitrary partitions (real catchments drain in all 8 directions, rectangles would give a cyclic rank graph):
    partition_subcatchments cuts the river tree; each part is a plan with an owner mask."""
    m = synth.make_mesh_d8(40, 56, ng=1, seed=3) if kind == "d8" else synth.make_mesh(48, 64, ng=1, mask_corner=True)
    owner = tiles.partition_subcatchments(m, world) if cut == "sub" else tiles.partition_trunk(m, world)
    act = np.asarray(m.active_cell) == 1
    assert np.all(owner[act] >= 0) and np.all(owner[~act] == -1)
    sizes = np.bincount(owner[act], minlength=world)
    assert sizes.sum() == m.nac and sizes.max() - sizes.min() <= 1                  # balanced to the cell
    ds, _ = synth.downstream_index(m.flwdir, m.active_cell)
    oc = np.asarray(owner).reshape(-1)
    src = np.flatnonzero(ds >= 0)
    assert np.all(oc[ds[src]] >= oc[src])                                           # part ids = a topological order
    of = np.asarray(owner).reshape(-1, order="F")                                   # flat = row + col * nrow
    P = [probe(m, None, owner_mask=(owner == r)) for r in range(world)]
    assert [p["cells"] for p in P] == sizes.tolist()
    n_cross = int(np.count_nonzero(oc[ds[src]] != oc[src]))
    if cut == "trunk":                                                              # two levels: leaves -> trunk
        assert np.all(oc[ds[src]][oc[ds[src]] != oc[src]] == world - 1)
    assert sum(p["n_out"] for p in P) == sum(p["n_in"] for p in P) == n_cross > 0
    for a in range(world):
        own = of[P[a]["out_dst"]] if P[a]["n_out"] else np.zeros(0, int)
        assert np.all(own > a)
        if cut == "trunk":
            assert (P[a]["n_in"] == 0) == (a < world - 1) and np.all(own == world - 1)
        assert np.all(of[P[a]["out_src"]] == a) and np.all(of[P[a]["in_dst"]] == a)
        for b in np.unique(own):
            sel = own == b
            src_b = of[P[b]["in_src"]] == a
            assert np.array_equal(P[a]["out_src"][sel], P[b]["in_src"][src_b])       # same edges, same order
            assert np.array_equal(P[a]["out_dst"][sel], P[b]["in_dst"][src_b])


def _gloo_worker(rank, world, port, q, cut="rect"):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pr, pc = tiles.tile_grid(world)
        if cut == "rect":
            m = synth.make_mesh(40, 56, ng=1)
            me = probe(m, tiles.tile_rect(rank, m.nrow, m.ncol, pr, pc))
        else:            # full-D8 flow field: every rank cuts the river tree the same way and owns one part
            m = synth.make_mesh_d8(40, 56, ng=1, seed=3)
            owner = tiles.partition_trunk(m, world)
            me = probe(m, None, owner_mask=(owner == rank))
        if rank == 0:    # upstream tile: tell the peer which series it will send, in its own order
            assert me["n_in"] == 0 and me["n_out"] > 0
            dist.send(torch.tensor([me["n_out"]]), 1)
            dist.send(torch.from_numpy(np.stack([me["out_src"], me["out_dst"]]).astype(np.int64)), 1)
            ok = torch.zeros(1, dtype=torch.int64)
            dist.recv(ok, 1)
            q.put((rank, int(ok.item())))
        else:
            n = torch.zeros(1, dtype=torch.int64)
            dist.recv(n, 0)
            got = torch.zeros((2, int(n.item())), dtype=torch.int64)
            dist.recv(got, 0)
            same = int(int(n.item()) == me["n_in"] and np.array_equal(got[0].numpy(), me["in_src"]) and np.array_equal(got[1].numpy(), me["in_dst"]))
            dist.send(torch.tensor([same]), 0)
            q.put((rank, same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cut", ["rect", "trunk"])
def test_two_ranks_agree_over_gloo(cut):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if cut == "trunk" else 0)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q, cut)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: 1, 1: 1}


def _calib_problem(n):
    rng = np.random.default_rng(11)
    tgt, w = rng.uniform(-0.2, 1.2, n), 10.0 ** rng.uniform(-1, 1, n)

    def part(x, sel):              # cost and gradient of the variables in sel only
        d = x[sel] - tgt[sel]
        g = np.zeros(n)
        g[sel] = 2 * w[sel] * d
        return float(np.sum(w[sel] * d * d)), g
    return part


def _gloo_calibration_worker(rank, world, port, q):
    """The protocol of optimize_lbfgsb(decomposition=...) with the sweep replaced by a separable function: rank 0 runs the library's
    L-BFGS-B, the others evaluate what it asks for until it says the search is over."""
    import torch.distributed as dist
    from smash_amd.optimize import _Held, _lbfgsb_native
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 60
        part = _calib_problem(n)
        mine = np.arange(n) % world == rank
        dec = tiles.TorchDecomposition(mine.reshape(n, 1), n)
        evals = []

        def fg(xc):
            x = dec.bcast_point(xc)
            f, g = part(x, mine)
            v = np.array([f])
            dec.allreduce(v)
            dec.allreduce(g)
            evals.append(float(v[0]))
            return float(v[0]), g
        if rank == 0:
            x, f, info = _lbfgsb_native(fg, np.full(n, 0.5), 10, 10.0, 1e-12, 30, 400, None)
            x = dec.bcast_point(x, done=True)
        else:
            while True:
                xw = dec.bcast_point(None)
                if xw is None:
                    break
                fg(_Held(xw))
            x = dec.final_point
        q.put((rank, evals, x.tolist()))
    finally:
        dist.destroy_process_group()


def test_calibration_protocol_over_gloo():
    """tiles.TorchDecomposition (gloo, world_size 2) under the library's L-BFGS-B: both ranks see the same evaluations and end on
    the same point, which is the one a single process finds."""
    import torch.multiprocessing as mp
    from smash_amd.optimize import _lbfgsb_native
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 23
    procs = [ctx.Process(target=_gloo_calibration_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (e, x) for r, e, x in (q.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and len(res[0][0]) > 5
    n = 60
    part = _calib_problem(n)
    xs, fs, info = _lbfgsb_native(lambda x: part(x, np.ones(n, bool)), np.full(n, 0.5), 10, 10.0, 1e-12, 30, 400, None)
    assert info["funcalls"] == len(res[0][0]) and np.allclose(xs, res[0][1], rtol=0, atol=1e-12)
