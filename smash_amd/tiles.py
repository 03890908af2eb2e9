"""Domain decomposition of the catchment grid across GPUs.

The grid is cut into Pr x Pc rectangular tiles, one rank (process, GPU) per tile.  Every rank builds the routing
schedule of its own cells from the *global* flow directions (smashx plan with cfg.tile); a cell whose D8 receiver
lies in another tile publishes its discharge series, the receiver's tile consumes it as an inlet -- the same
exchange-series mechanism that links routing groups inside one GPU.  Per pipeline sub-chunk and per tile border
one message of (edges x steps) floats moves downstream in the forward sweep and upstream (adjoint of the
boundary discharge) in the reverse sweep: point-to-point send/recv over RCCL (torch.distributed backend "nccl"),
no collective on the data path.  Gradients stay tile-local; the cost is the sum of per-tile partial costs.

The flow network must make the rank graph acyclic (a river may not leave a part and come back): true for
rectangles on the synthetic E/SE/S catchments; real catchments (all eight D8 codes) are cut along the river tree
instead -- partition_subcatchments() -- and a part is handed to the plan as an owner mask (Solver(owner_mask=...)).
"""
from __future__ import annotations

import numpy as np


def tile_grid(world: int):
    """Pr x Pc for 1, 2, 4, 8 ranks (2 x 4 at 8: BASELINE.json configs[4])."""
    return {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}.get(world) or (1, world)


def tile_rect(rank: int, nrow: int, ncol: int, pr: int, pc: int):
    i, j = divmod(rank, pc)
    r = [nrow * q // pr for q in range(pr + 1)]
    c = [ncol * q // pc for q in range(pc + 1)]
    return (r[i], r[i + 1], c[j], c[j + 1])


def owner_of(flat, nrow: int, ncol: int, pr: int, pc: int):
    """rank owning each flat (row + col*nrow) cell index."""
    flat = np.asarray(flat, np.int64)
    row, col = flat % nrow, flat // nrow
    rb = np.array([nrow * q // pr for q in range(pr + 1)])
    cb = np.array([ncol * q // pc for q in range(pc + 1)])
    i = np.searchsorted(rb, row, side="right") - 1
    j = np.searchsorted(cb, col, side="right") - 1
    return (i * pc + j).astype(np.int64)


def partition_subcatchments(mesh, nparts: int):
    """Cut the river tree of `mesh` into nparts sub-catchment sets of nac / nparts cells each (what replaces
    rectangles when the flow field is not E/SE/S).  Part p is filled, largest first, with not-yet-assigned subtrees that
    still fit; whatever a detached subtree drains into is assigned later, so part ids are a topological order of the rank
    graph (acyclic by construction) and every part is a union of whole sub-catchments minus the ones cut off before.
    Returns owner (nrow, ncol) int32: part id of every active cell, -1 elsewhere."""
    from . import synth
    nrow, ncol = mesh.nrow, mesh.ncol
    act = (np.asarray(mesh.active_cell) == 1).reshape(-1)
    ds, _ = synth.downstream_index(np.asarray(mesh.flwdir), np.asarray(mesh.active_cell))      # C-order flat, -1 = outlet
    n = nrow * ncol
    rem = np.where(act, np.asarray(mesh.flwacc).reshape(-1).astype(np.int64), 0)                # size of the unassigned subtree
    owner = np.full(n, -1, np.int64)
    up = [[] for _ in range(n)]
    for c in np.flatnonzero(ds >= 0):
        up[ds[c]].append(int(c))
    nac = int(np.count_nonzero(act))
    for part in range(nparts - 1):
        cap = (nac * (part + 1)) // nparts - (nac * part) // nparts
        while cap > 0:
            fits = np.where((owner < 0) & act & (rem <= cap), rem, 0)
            root = int(np.argmax(fits))
            size = int(fits[root])
            if size == 0:
                break
            stack = [root]
            while stack:                                 # the unassigned upstream closure of root
                c = stack.pop()
                owner[c] = part
                stack.extend(u for u in up[c] if owner[u] < 0)
            d = ds[root]
            while d >= 0:                                # its ancestors lose that many cells
                rem[d] -= size
                d = ds[d]
            cap -= size
    owner[(owner < 0) & act] = nparts - 1
    return np.asfortranarray(owner.reshape(nrow, ncol).astype(np.int32))


def median_slots(wgauge_global, local_gauges):
    """Arguments of Solver.set_median_slots for one tile: the negative-weight gauges of the whole decomposition are numbered in
    global gauge order (the order the reference's compute_jobs collects them in, mwd_cost.f90:139-150); local_gauges = the global
    indices of this tile's gauges, in the tile's order.  Returns (nslots, slot_of_gauge)."""
    w = np.asarray(wgauge_global, np.float32)
    neg = np.flatnonzero(w < 0)
    slot = {int(g): i for i, g in enumerate(neg)}
    return len(neg), np.array([slot.get(int(g), -1) for g in local_gauges], np.int32)


def decomposition_cost(cost_jobs_per_rank, cost_jreg, wjreg):
    """The cost of a decomposition from the ranks' smashx_costs: every rank evaluates the regulariser over the whole grid (same
    cost_jreg everywhere), so it enters once: sum(cost_jobs) + wjreg * cost_jreg (mwd_cost.f90:300)."""
    return float(np.sum(np.asarray(cost_jobs_per_rank, np.float64)) + float(wjreg) * float(cost_jreg))


class Decomposition:
    """What smash_amd.optimize_lbfgsb(decomposition=...) needs from the ranks of a tile decomposition (the calibration over several
    GPUs): rank, the cells this rank owns, a sum over the ranks and rank 0's trial points on every rank.  Subclasses supply the
    transport: TorchDecomposition (torch.distributed: RCCL on GPUs, gloo in rehearsals), ThreadDecomposition (plans of one process,
    tests)."""

    def __init__(self, rank, world, owned):
        self.rank, self.world = int(rank), int(world)
        self.owned = np.asarray(owned, bool)             # (nrow, ncol): active cells of this rank's part
        self.final_point = None

    def allreduce(self, v):                              # in place, float64
        raise NotImplementedError

    def _bcast(self, head, x):                           # rank 0: sends (head, x); others: return (head, x)
        raise NotImplementedError

    def bcast_point(self, x, done=False):
        """rank 0: hand the trial point x (done: the final iterate) to the other ranks, returns x.  Other ranks: called with None,
        return the next point to evaluate, or None once rank 0 is done (the final iterate is then in final_point)."""
        if hasattr(x, "x"):                              # a point this rank has already received
            return x.x
        if self.rank == 0:
            x = np.ascontiguousarray(x, np.float64)
            self._bcast(0 if done else 1, x)
            return x
        head, xr = self._bcast(None, None)
        if head == 0:
            self.final_point = xr
            return None
        return xr


class TorchDecomposition(Decomposition):
    """The ranks of a torch.distributed process group (backend nccl = RCCL: device tensors; gloo: host tensors)."""

    def __init__(self, owned, n_control, device=None, group=None):
        import torch
        import torch.distributed as dist
        super().__init__(dist.get_rank(group), dist.get_world_size(group), owned)
        self.torch, self.dist, self.group = torch, dist, group
        self.dev = device if dist.get_backend(group) == "nccl" else "cpu"
        self.n = int(n_control)

    def allreduce(self, v):
        t = self.torch.from_numpy(v).to(self.dev)
        self.dist.all_reduce(t, group=self.group)
        v[:] = t.cpu().numpy()

    def _bcast(self, head, x):
        buf = self.torch.zeros(self.n + 1, dtype=self.torch.float64, device=self.dev)
        if self.rank == 0:
            buf[0] = float(head)
            buf[1:] = self.torch.from_numpy(x).to(self.dev)
        self.dist.broadcast(buf, 0, group=self.group)
        if self.rank == 0:
            return head, x
        b = buf.cpu().numpy()
        return int(b[0]), b[1:].copy()


class ThreadDecomposition(Decomposition):
    """Plans of ONE process, one thread per rank (tests/test_gpu_tiles.py): make(world) returns the shared state, then one
    ThreadDecomposition(shared, rank, owned) per thread."""

    @staticmethod
    def make(world):
        import threading
        return {"world": world, "bar": threading.Barrier(world), "lock": threading.Lock(), "acc": None, "msg": None}

    def __init__(self, shared, rank, owned):
        super().__init__(rank, shared["world"], owned)
        self.sh = shared

    def allreduce(self, v):
        sh = self.sh
        with sh["lock"]:
            sh["acc"] = v.copy() if sh["acc"] is None else sh["acc"] + v
        sh["bar"].wait(timeout=300)
        v[:] = sh["acc"]
        if sh["bar"].wait(timeout=300) == 0:
            sh["acc"] = None
        sh["bar"].wait(timeout=300)

    def _bcast(self, head, x):
        sh = self.sh
        if self.rank == 0:
            sh["msg"] = (head, x.copy())
        sh["bar"].wait(timeout=300)
        head, x = sh["msg"]
        sh["bar"].wait(timeout=300)
        return head, x.copy()


class PeerLists:
    """For one tile: which rows of the out / in message buffers go to / come from which peer rank.
    owner: optional (nrow, ncol) part id per cell (partition_subcatchments) instead of the pr x pc rectangles."""

    def __init__(self, solver, nrow, ncol, pr, pc, owner=None):
        out_src, out_dst, in_src, in_dst = solver.halo_edges()
        self.n_out, self.n_in = len(out_src), len(in_src)
        if owner is not None:
            of = np.asarray(owner).reshape(-1, order="F").astype(np.int64)        # flat = row + col * nrow
            o_owner = of[np.asarray(out_dst, np.int64)] if self.n_out else np.zeros(0, np.int64)
            i_owner = of[np.asarray(in_src, np.int64)] if self.n_in else np.zeros(0, np.int64)
        else:
            o_owner = owner_of(out_dst, nrow, ncol, pr, pc) if self.n_out else np.zeros(0, np.int64)
            i_owner = owner_of(in_src, nrow, ncol, pr, pc) if self.n_in else np.zeros(0, np.int64)
        self.out_owner, self.in_owner = o_owner.astype(np.int32), i_owner.astype(np.int32)
        self.out_peers = {int(p): np.flatnonzero(o_owner == p) for p in np.unique(o_owner)}
        self.in_peers = {int(p): np.flatnonzero(i_owner == p) for p in np.unique(i_owner)}
        self.out_src, self.out_dst, self.in_src, self.in_dst = out_src, out_dst, in_src, in_dst


class RcclExchange:
    """Native exchange: the plan posts grouped ncclSend / ncclRecv on its routing stream itself (smashx_set_exchange); nothing
    runs on the host per sub-chunk.  comm: a smash_amd.solver.Comm spanning the ranks of the decomposition."""

    def __init__(self, solver, comm, nrow, ncol, pr, pc, owner=None):
        self.peers = PeerLists(solver, nrow, ncol, pr, pc, owner)
        solver.set_exchange(comm, self.peers.out_owner, self.peers.in_owner)


def share_one_gpu_env(rank: int):
    """Rehearsals on a one-GPU box: RCCL refuses two ranks on one device unless they claim different hosts; with a host id
    per rank the duplicate-GPU check passes and the socket transport (loopback) carries the messages.  Call before the first
    RCCL communicator is created.  Production runs (one GPU per rank) never call this."""
    import os
    os.environ["NCCL_HOSTID"] = f"smashx-rank-{rank}"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    for k in ("NCCL_IB_DISABLE", "NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE"):
        os.environ.setdefault(k, "1")


class TorchDistExchange:
    """Halo exchange over torch.distributed point-to-point ops (RCCL on GPUs)."""

    def __init__(self, solver, nrow, ncol, pr, pc, device, owner=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.dev = torch, dist, device
        self.host_staged = dist.get_backend() != "nccl"     # gloo (rehearsals without RCCL) moves host tensors
        self.peers = PeerLists(solver, nrow, ncol, pr, pc, owner)
        _, self.tp = solver.chunking()
        self.out_buf = torch.zeros(max(self.peers.n_out, 1) * self.tp, dtype=torch.float32, device=device)
        self.in_buf = torch.zeros(max(self.peers.n_in, 1) * self.tp, dtype=torch.float32, device=device)
        self.idx_out = {p: torch.from_numpy(ix).to(device) for p, ix in self.peers.out_peers.items()}
        self.idx_in = {p: torch.from_numpy(ix).to(device) for p, ix in self.peers.in_peers.items()}
        solver.set_halo(self.out_buf.data_ptr(), self.in_buf.data_ptr(), self)
        # one handshake per neighbour now: RCCL creates its point-to-point channels on first use, which belongs to the set-up
        # and not to the first sweep
        # (both directions with every neighbour in ONE batch: a connection first used inside a sweep would be set up while the two
        # ranks sit in different groups of their dependency chain -- see smashx_set_exchange)
        mdev = "cpu" if self.host_staged else device
        nbrs = sorted(set(self.idx_out) | set(self.idx_in))
        hello = {p: torch.zeros(1, dtype=torch.float32, device=mdev) for p in nbrs}
        ops = [dist.P2POp(dist.isend, torch.ones(1, dtype=torch.float32, device=mdev), p) for p in nbrs] + \
              [dist.P2POp(dist.irecv, hello[p], p) for p in nbrs]
        if ops:
            if self.host_staged:
                for r in [op.op(op.tensor, op.peer) for op in ops]:
                    r.wait()
            else:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
                torch.cuda.current_stream().synchronize()

    def __call__(self, phase, t0, nsteps):
        torch, dist = self.torch, self.dist
        w = 4 * ((nsteps + 3) // 4)                       # floats per edge in this sub-chunk
        # forward: recv on the in edges (0), send on the out edges (1); adjoint: recv on out (2), send on in (3)
        use_out = phase in (1, 2)
        buf = self.out_buf if use_out else self.in_buf
        n = self.peers.n_out if use_out else self.peers.n_in
        idx = self.idx_out if use_out else self.idx_in
        view = buf[: n * w].view(n, w)
        ops, tmps = [], {}
        mdev = "cpu" if self.host_staged else self.dev
        for p, ix in idx.items():
            if phase in (1, 3):
                tmps[p] = view[ix].contiguous().to(mdev)
                ops.append(dist.P2POp(dist.isend, tmps[p], p))
            else:
                tmps[p] = torch.empty((len(ix), w), dtype=torch.float32, device=mdev)
                ops.append(dist.P2POp(dist.irecv, tmps[p], p))
        if ops:
            if self.host_staged:
                for r in [op.op(op.tensor, op.peer) for op in ops]:
                    r.wait()
            else:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
        if phase in (0, 2):
            for p, ix in idx.items():
                view[ix] = tmps[p].to(self.dev)
        torch.cuda.current_stream().synchronize()
        return 0


class NoExchange:
    """Diagnostics: the halo hooks of one part run alone -- nothing moves, inflow stays zero."""

    def __init__(self, solver, device):
        import torch
        n_out, n_in = solver.halo_counts()
        _, tp = solver.chunking()
        self.out_buf = torch.zeros(max(n_out, 1) * tp, dtype=torch.float32, device=device)
        self.in_buf = torch.zeros(max(n_in, 1) * tp, dtype=torch.float32, device=device)
        solver.set_halo(self.out_buf.data_ptr(), self.in_buf.data_ptr(), self)

    def __call__(self, phase, t0, nsteps):
        return 0


def partition_trunk(mesh, nparts: int, trunk_share: float = 1.0):
    """Depth-2 cut of the river tree: parts 0 .. nparts-2 are unions of WHOLE sub-catchments (subtrees no earlier cut has
    touched), the last part is the trunk everything else drains through.  No leaf part receives anything, every leaf part
    sends only to the trunk, so the rank graph has two levels whatever nparts is (partition_subcatchments tends to a chain
    of nparts levels; rectangles on the E/SE/S catchment have Pr + Pc - 1).  Parts are balanced to the cell;
    trunk_share scales the trunk's share of the cells (its routing is the deep, latency-bound part of the network).
    Returns owner (nrow, ncol) int32, -1 on inactive cells."""
    from . import synth
    nrow, ncol = mesh.nrow, mesh.ncol
    n = nrow * ncol
    act = (np.asarray(mesh.active_cell) == 1).reshape(-1)
    ds, _ = synth.downstream_index(np.asarray(mesh.flwdir), np.asarray(mesh.active_cell))     # C-order flat, -1 = outlet
    acc = np.where(act, np.asarray(mesh.flwacc).reshape(-1).astype(np.int64), 0)              # subtree sizes
    src = np.flatnonzero(ds >= 0)
    order = src[np.argsort(ds[src], kind="stable")]                                           # children, grouped by parent
    ptr = np.concatenate(([0], np.cumsum(np.bincount(ds[src], minlength=n))))

    def subtree(root):
        out, f = [np.array([root], np.int64)], np.array([root], np.int64)
        while True:
            c0 = ptr[f]
            k = ptr[f + 1] - c0
            tot = int(k.sum())
            if tot == 0:
                return np.concatenate(out)
            first = np.cumsum(k) - k
            f = order[np.repeat(c0 - first, k) + np.arange(tot)]
            out.append(f)

    nac = int(np.count_nonzero(act))
    trunk = int(round(nac / nparts * trunk_share)) if nparts > 1 else nac
    leaf_total = nac - trunk
    owner = np.full(n, -1, np.int64)
    avail = act.copy()                                   # may still become the root of a whole, untouched sub-catchment
    for part in range(nparts - 1):
        cap = (leaf_total * (part + 1)) // (nparts - 1) - (leaf_total * part) // (nparts - 1)
        while cap > 0:
            fits = np.where(avail & (acc <= cap), acc, 0)
            root = int(np.argmax(fits))
            size = int(fits[root])
            if size == 0:
                break
            cells = subtree(root)
            owner[cells] = part
            avail[cells] = False
            d = ds[root]
            while d >= 0 and avail[d]:                   # everything downstream of a cut is trunk material from now on
                avail[d] = False
                d = ds[d]
            cap -= size
    owner[act & (owner < 0)] = nparts - 1
    return np.asfortranarray(owner.reshape(nrow, ncol).astype(np.int32))
