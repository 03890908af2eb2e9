"""Deterministic synthetic catchments, forcing and parameters.

Bench/test input generator -- not part of the solver.  Everything that has to be reproduced
bit-for-bit on two devices (numpy on the host for the CPU baseline, torch on the GPU for the
resident HBM forcing) is written in pure *integer* arithmetic on an ``xp`` array module followed
by ONE float32 multiply, so numpy and torch give identical forcing for any (cell, time) window.

D8 convention (reference smash/mesh/mw_meshing.f90:163-164): code k = 1..8 = N, NE, E, SE, S, SW,
W, NW flows to (row + DROW[k-1], col + DCOL[k-1]) with row increasing southwards.
"""
from __future__ import annotations

import numpy as np

DROW = np.array([-1, -1, 0, 1, 1, 1, 0, -1], dtype=np.int64)
DCOL = np.array([0, 1, 1, 1, 0, -1, -1, -1], dtype=np.int64)

SEED = 20240501
_M32 = 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------
# integer hash usable with numpy int64 arrays and torch int64 tensors alike
# ----------------------------------------------------------------------------------------------
def _mix32(h):
    """murmur3 finaliser on the low 32 bits of int64 lanes (wrap-around keeps the low bits)."""
    h = h & _M32
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & _M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & _M32
    h = h ^ (h >> 16)
    return h


def _hash3(seed, a, b, c):
    h = _mix32(a + 0x9E3779B9 * (seed & 0xFFFF))
    h = _mix32(h ^ ((b * 0x27D4EB2F) & _M32))
    h = _mix32(h ^ ((c * 0x165667B1) & _M32))
    return h


# ----------------------------------------------------------------------------------------------
# mesh
# ----------------------------------------------------------------------------------------------
def make_flwdir(nrow: int, ncol: int, seed: int = SEED) -> np.ndarray:
    """Acyclic, fully connected D8 field draining to the south-east corner.

    Directions are drawn from {E(3), SE(4), S(5)} by hash(seed,row,col) mod 3; the last row is
    forced E and the last column forced S, so every cell reaches (nrow-1, ncol-1).
    """
    r = np.arange(nrow, dtype=np.int64)[:, None]
    c = np.arange(ncol, dtype=np.int64)[None, :]
    h = _hash3(seed, r, c, np.int64(7))
    fd = (3 + (h % 3)).astype(np.int32)
    fd[nrow - 1, :] = 3
    fd[:, ncol - 1] = 5
    return np.asfortranarray(fd)


def downstream_index(flwdir: np.ndarray, active: np.ndarray | None = None):
    """Flat (C-order) index of the cell each cell drains to, -1 for outlets / nodata."""
    nrow, ncol = flwdir.shape
    fd = np.asarray(flwdir).astype(np.int64)
    ok = (fd >= 1) & (fd <= 8)
    if active is not None:
        ok &= np.asarray(active) == 1
    k = np.where(ok, fd - 1, 0)
    r = np.arange(nrow, dtype=np.int64)[:, None] + DROW[k]
    c = np.arange(ncol, dtype=np.int64)[None, :] + DCOL[k]
    inside = ok & (r >= 0) & (r < nrow) & (c >= 0) & (c < ncol)
    rr = np.where(inside, r, 0)
    cc = np.where(inside, c, 0)
    # a pit pair (two cells pointing at each other) stops the accumulation (mw_meshing.f90:183)
    pit = inside & (np.abs(fd - fd[rr, cc]) == 4)
    inside &= ~pit
    if active is not None:
        inside &= np.asarray(active)[rr, cc] == 1
    ds = np.where(inside, rr * ncol + cc, -1)
    return ds.reshape(-1), ok.reshape(-1)


def flow_accumulation(flwdir: np.ndarray, active: np.ndarray | None = None) -> np.ndarray:
    """Number of cells draining through each cell, itself included (mw_meshing.f90:204-233 semantics),
    computed by a vectorised Kahn sweep (one numpy pass per topological level)."""
    nrow, ncol = flwdir.shape
    ds, ok = downstream_index(flwdir, active)
    n = nrow * ncol
    acc = np.where(ok, 1, 0).astype(np.int64)
    indeg = np.zeros(n, dtype=np.int64)
    np.add.at(indeg, ds[ds >= 0], 1)
    front = np.flatnonzero(ok & (indeg == 0))
    while front.size:
        d = ds[front]
        m = d >= 0
        src, d = front[m], d[m]
        np.add.at(acc, d, acc[src])
        np.subtract.at(indeg, d, 1)
        cand = np.unique(d)
        front = cand[indeg[cand] == 0]
    out = acc.reshape(nrow, ncol).astype(np.int32)
    if active is not None:
        out = np.where(np.asarray(active) == 1, out, -99).astype(np.int32)
    return np.asfortranarray(out)


def make_path(flwacc: np.ndarray) -> np.ndarray:
    """Cell visiting order = ascending flow accumulation (reference meshing.py:216-224); 0-based
    (2, nrow*ncol) like the Python side of the reference sees it."""
    a = np.ascontiguousarray(flwacc)
    idx = np.argsort(a, axis=None, kind="stable")
    r, c = np.unravel_index(idx, a.shape)
    path = np.zeros((2, a.size), dtype=np.int32, order="F")
    path[0, :] = r
    path[1, :] = c
    return path


def pick_gauges(flwacc: np.ndarray, ng: int):
    """Outlet + (ng-1) interior cells with the largest flow accumulation, kept apart by 1/8 of the grid."""
    nrow, ncol = flwacc.shape
    a = np.ascontiguousarray(flwacc).reshape(-1).astype(np.int64)
    order = np.argsort(-a, kind="stable")
    r, c = order // ncol, order % ncol
    ok = a[order] > 0
    sep = max(1, max(nrow, ncol) // 8)
    pos = []
    while len(pos) < ng and ok.any():
        i = int(np.argmax(ok))                    # best remaining candidate
        pos.append((int(r[i]), int(c[i])))
        ok &= (np.abs(r - r[i]) + np.abs(c - c[i])) >= sep
    k = 0
    while len(pos) < ng and k < order.size:       # tiny grids: relax the separation rule
        p = (int(r[k]), int(c[k]))
        if p not in pos and a[order[k]] > 0:
            pos.append(p)
        k += 1
    return np.asfortranarray(np.array(pos, dtype=np.int32).reshape(ng, 2))


class Mesh:
    """Plain container with the MeshDT fields the path reads (mwd_mesh.f90:45-72); 0-based indices."""

    def __init__(self, nrow, ncol, dx, flwdir, flwacc, path, active_cell, gauge_pos, area):
        self.nrow, self.ncol, self.dx = int(nrow), int(ncol), float(dx)
        self.flwdir = np.asfortranarray(flwdir, dtype=np.int32)
        self.flwacc = np.asfortranarray(flwacc, dtype=np.int32)
        self.path = np.asfortranarray(path, dtype=np.int32)
        self.active_cell = np.asfortranarray(active_cell, dtype=np.int32)
        self.gauge_pos = np.asfortranarray(np.asarray(gauge_pos, dtype=np.int32).reshape(-1, 2))
        self.area = np.ascontiguousarray(area, dtype=np.float32)
        self.ng = int(self.gauge_pos.shape[0])
        self.nac = int(np.count_nonzero(self.active_cell == 1))


def make_mesh(nrow: int, ncol: int, ng: int = 8, dx: float = 1000.0, seed: int = SEED,
              mask_corner: bool = False) -> Mesh:
    flwdir = make_flwdir(nrow, ncol, seed)
    active = np.ones((nrow, ncol), dtype=np.int32, order="F")
    if mask_corner:
        # knock out the whole upstream area of one interior cell's neighbours (north-west block):
        # the mask stays upstream-closed because flow only goes E / SE / S.
        active[: nrow // 3, : ncol // 3] = 0
        flwdir = np.asfortranarray(np.where(active == 1, flwdir, -99).astype(np.int32))
    flwacc = flow_accumulation(flwdir, active if mask_corner else None)
    path = make_path(np.where(active == 1, flwacc, -99))
    gauge_pos = pick_gauges(np.where(active == 1, flwacc, -1), ng)
    area = np.array([float(flwacc[r, c]) * dx * dx for r, c in gauge_pos], dtype=np.float32)
    return Mesh(nrow, ncol, dx, flwdir, flwacc, path, active, gauge_pos, area)


def make_mesh_d8(nrow: int, ncol: int, ng: int = 3, dx: float = 1000.0, seed: int = SEED, radius: float = 0.0) -> Mesh:
    """A catchment that uses all eight D8 codes: every cell drains towards an interior outlet along the steepest
    descent of  distance-to-outlet + 0.45 * noise  (a king move always shortens the distance by > 0.9, so the
    field is acyclic and has one basin).  The outlet flows south into a masked cell; radius > 0 also masks
    everything farther than radius * min(nrow, ncol) from the outlet (ragged, upstream-closed mask)."""
    r = np.arange(nrow, dtype=np.int64)[:, None]
    c = np.arange(ncol, dtype=np.int64)[None, :]
    r0, c0 = (3 * nrow) // 5, (11 * ncol) // 20
    u = (_hash3(seed, r, c, np.int64(11)) & 0xFFFF).astype(np.float64) / 65536.0
    phi = np.hypot(r - r0, c - c0) + 0.45 * u
    active = np.ones((nrow, ncol), dtype=np.int32)
    active[r0 + 1, c0] = 0
    if radius > 0:
        active[np.hypot(r - r0, c - c0) > radius * min(nrow, ncol)] = 0
        active[r0 + 1, c0] = 0
    big = np.float64(1e30)
    pad = np.full((nrow + 2, ncol + 2), big)
    pad[1:-1, 1:-1] = np.where(active == 1, phi, big)
    best = np.full((nrow, ncol), big)
    fd = np.zeros((nrow, ncol), dtype=np.int32)
    for k in range(8):
        nb = pad[1 + DROW[k]: 1 + DROW[k] + nrow, 1 + DCOL[k]: 1 + DCOL[k] + ncol]
        take = nb < best
        best = np.where(take, nb, best)
        fd = np.where(take, k + 1, fd)
    assert np.all((best < phi) | (active != 1) | ((r == r0) & (c == c0))), "local minimum in the synthetic relief"
    fd[r0, c0] = 5
    flwdir = np.asfortranarray(np.where(active == 1, fd, -99).astype(np.int32))
    active = np.asfortranarray(active)
    flwacc = flow_accumulation(flwdir, active)
    path = make_path(np.where(active == 1, flwacc, -99))
    gauge_pos = pick_gauges(np.where(active == 1, flwacc, -1), ng)
    area = np.array([float(flwacc[a, b]) * dx * dx for a, b in gauge_pos], dtype=np.float32)
    return Mesh(nrow, ncol, dx, flwdir, flwacc, path, active, gauge_pos, area)


def make_mesh_france(which="all", ng: int = 8, fixture: str | None = None) -> Mesh:
    """A REAL river network: the reference's 1-km D8 raster of France (all eight codes, 957 k cells with a direction, hundreds of
    coastal and border outlets), from the data fixture tests/golden/mesh/france_d8.npz (tests/golden/make_france_d8.py).
    which = "all": every cell of the raster that has a direction is active (a forest of basins); an integer K: the K largest basins
    (by cells draining through their outlets), cropped to their bounding box -- K = 1 is the Loire.  Mesh fields by the rules of the
    reference's generator (smash/mesh/meshing.py:216-297, mw_meshing.f90:204-233: flow accumulation, path = ascending accumulation);
    gauges: the ng cells with the largest accumulation, kept apart (pick_gauges)."""
    import os
    if fixture is None:
        fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mesh", "france_d8.npz")
    z = np.load(fixture)
    fd = z["flwdir"].astype(np.int32)
    dx = float(z["dx"])
    nrow, ncol = fd.shape
    act = fd > 0
    n = nrow * ncol

    def outlets(a):
        ds, _ = downstream_index(np.where(a, fd, -99), a.astype(np.int32))
        root = np.where(ds >= 0, ds, np.arange(n))
        for _ in range(24):                              # pointer doubling: every cell learns where its water ends up
            root = root[root]
        return ds, root
    # The raster holds a few closed loops (pairs of cells that point at each other in flat or endorheic spots: mw_meshing.f90:183 stops
    # the accumulation there).  A catchment delineated from an outlet never contains one; on the whole raster they are taken out of the
    # mask, so that what drained into them ends in an inactive cell like water that reaches the sea.
    r = np.arange(nrow, dtype=np.int64)[:, None] + DROW[np.where(act, fd - 1, 0)]
    c = np.arange(ncol, dtype=np.int64)[None, :] + DCOL[np.where(act, fd - 1, 0)]
    inside = act & (r >= 0) & (r < nrow) & (c >= 0) & (c < ncol)
    raw = np.where(inside, np.where(inside, r, 0) * ncol + np.where(inside, c, 0), -1).reshape(-1)
    raw = np.where((raw >= 0) & act.reshape(-1)[np.maximum(raw, 0)], raw, -1)
    jump = np.where(raw >= 0, raw, np.arange(n))
    for _ in range(24):
        jump = jump[jump]
    img = np.unique(jump[act.reshape(-1)])
    loop = img[raw[img] >= 0]                            # images of f^(2^24) that still have a downstream cell: the cells on loops
    act.reshape(-1)[loop] = False
    if which != "all":
        ds, root = outlets(act)
        flat = act.reshape(-1)
        sizes = np.bincount(root[flat], minlength=n)
        keep = np.argsort(-sizes, kind="stable")[: int(which)]
        act = (np.isin(root, keep) & flat).reshape(nrow, ncol)
        rr, cc = np.flatnonzero(act.any(axis=1)), np.flatnonzero(act.any(axis=0))
        fd, act = fd[rr[0]: rr[-1] + 1, cc[0]: cc[-1] + 1], act[rr[0]: rr[-1] + 1, cc[0]: cc[-1] + 1]
        nrow, ncol = fd.shape
    active = np.asfortranarray(act.astype(np.int32))
    flwdir = np.asfortranarray(np.where(act, fd, -99).astype(np.int32))
    flwacc = flow_accumulation(flwdir, active)
    path = make_path(np.where(active == 1, flwacc, -99))
    gauge_pos = pick_gauges(np.where(active == 1, flwacc, -1), ng)
    area = np.array([float(flwacc[a, b]) * dx * dx for a, b in gauge_pos], dtype=np.float32)
    return Mesh(nrow, ncol, dx, flwdir, flwacc, path, active, gauge_pos, area)


# ----------------------------------------------------------------------------------------------
# forcing (integer construction; identical under numpy and torch)
# ----------------------------------------------------------------------------------------------
_BS = 64      # spatial correlation length in cells
_TB = 4       # storm time block in steps


def _pet_tables():
    """daily PET (mm/day, 1..5 over the year) and a diurnal weight (sums to 1, zero at night)."""
    day = np.arange(366, dtype=np.float64)
    daily = (1.0 + 4.0 * np.sin(np.pi * day / 365.0) ** 2).astype(np.float32)
    hour = np.arange(24, dtype=np.float64)
    w = np.where((hour >= 6) & (hour <= 19), np.sin(np.pi * (hour - 5.5) / 14.0) ** 2, 0.0)
    w = (w / w.sum()).astype(np.float32)
    return daily, w


def prcp_counts(xp, rows, cols, t, seed: int = SEED, gap_per_million: int = 1000):
    """Integer rain depth in 0.1 mm for cells (rows, cols) at time steps t (broadcast together),
    and the gap mask.  ``xp`` is numpy or torch; all operands int64."""
    R, x = rows // _BS, rows % _BS
    C, y = cols // _BS, cols % _BS
    T = t // _TB

    def node(dr, dc):
        h = _hash3(seed, T * 4099 + 1, R + dr, C + dc)
        wet = (h & 0xFF) < 56                     # ~22 % of storm nodes are wet
        amp = (h >> 8) & 0xFFF                    # 0..4095
        return xp.where(wet, amp, amp * 0)

    v = (node(0, 0) * (_BS - x) * (_BS - y) + node(1, 0) * x * (_BS - y)
         + node(0, 1) * (_BS - x) * y + node(1, 1) * x * y) >> 12          # 0..4095
    ht = _hash3(seed + 1, t, R * 0 + 3, C * 0 + 5)
    m = ht & 0xF                                                          # hourly modulation 0..15
    m = xp.where(m < 8, m * 0, m - 7)                                     # half of the hours dry
    n = (v * v * m) >> 19                                                 # 0..~255 (x 0.1 mm)
    hg = _hash3(seed + 2, t, rows, cols)
    gap = (hg % 1000000) < gap_per_million
    return n, gap


def forcing_block(rows, cols, t0: int, t1: int, seed: int = SEED, gap_per_million: int = 1000,
                  xp=np, device=None):
    """prcp, pet float32 arrays of shape (t1-t0, ncells) for the cells (rows[k], cols[k]).

    prcp: intermittent, spatially correlated storms quantised to 0.1 mm, with -99 gap markers
    (reference md_forward_structure.f90:106 treats negative forcing as a gap); pet: daily value x
    diurnal weight, spatially uniform."""
    daily, w = _pet_tables()
    if xp is np:
        r = np.asarray(rows, dtype=np.int64)[None, :]
        c = np.asarray(cols, dtype=np.int64)[None, :]
        t = np.arange(t0, t1, dtype=np.int64)[:, None]
        n, gap = prcp_counts(np, r, c, t, seed, gap_per_million)
        prcp = n.astype(np.float32) * np.float32(0.1)
        prcp = np.where(gap, np.float32(-99.0), prcp).astype(np.float32)
        tt = np.arange(t0, t1)
        pet1 = daily[(tt // 24) % 366] * w[tt % 24]
        pet = np.broadcast_to(pet1[:, None].astype(np.float32), prcp.shape).copy()
        return prcp, pet
    import torch
    r = rows.to(torch.int64)[None, :]
    c = cols.to(torch.int64)[None, :]
    t = torch.arange(t0, t1, dtype=torch.int64, device=device)[:, None]
    n, gap = prcp_counts(torch, r, c, t, seed, gap_per_million)
    prcp = n.to(torch.float32) * 0.1
    prcp = torch.where(gap, torch.full_like(prcp, -99.0), prcp)
    tt = np.arange(t0, t1)
    pet1 = torch.from_numpy((daily[(tt // 24) % 366] * w[tt % 24]).astype(np.float32)).to(device)
    pet = pet1[:, None].expand(prcp.shape).contiguous()
    return prcp, pet


def dense_forcing(mesh: Mesh, nt: int, seed: int = SEED, gap_per_million: int = 1000):
    """(nrow, ncol, nt) Fortran-ordered prcp/pet as Input_DataDT holds them (mwd_input_data.f90:32-50)."""
    r, c = np.meshgrid(np.arange(mesh.nrow), np.arange(mesh.ncol), indexing="ij")
    # memory order of a Fortran (nrow, ncol) plane = col-major: cell index = row + col*nrow
    rr = r.reshape(-1, order="F")
    cc = c.reshape(-1, order="F")
    prcp, pet = forcing_block(rr, cc, 0, nt, seed, gap_per_million)
    shp = (mesh.nrow, mesh.ncol, nt)
    return (np.asfortranarray(prcp.T.reshape(shp, order="F")),
            np.asfortranarray(pet.T.reshape(shp, order="F")))


# ----------------------------------------------------------------------------------------------
# parameters / states
# ----------------------------------------------------------------------------------------------
PARAM_NAMES = ("ci", "cp", "beta", "cft", "cst", "alpha", "exc", "b", "cusl1", "cusl2", "clsl",
               "ks", "ds", "dsm", "ws", "lr")          # md_constant.f90:37-57
STATE_NAMES = ("hi", "hp", "hft", "hst", "husl1", "husl2", "hlsl", "hlr")   # md_constant.f90:59-71

PARAM_DEFAULTS = dict(ci=1e-6, cp=200.0, beta=1000.0, cft=500.0, cst=500.0, alpha=0.9, exc=0.0,
                      b=0.3, cusl1=100.0, cusl2=500.0, clsl=2000.0, ks=20.0, ds=0.02, dsm=0.33,
                      ws=0.8, lr=5.0)                  # mwd_parameters.f90:150-167
STATE_DEFAULTS = dict(hi=0.01, hp=0.01, hft=0.01, hst=0.01, husl1=0.01, husl2=0.01, hlsl=0.01,
                      hlr=1e-6)                        # mwd_states.f90:117-126


def _smooth(nrow, ncol, lo, hi, phase):
    r = np.arange(nrow, dtype=np.float64)[:, None] / max(nrow, 2)
    c = np.arange(ncol, dtype=np.float64)[None, :] / max(ncol, 2)
    s = 0.5 + 0.25 * np.sin(2 * np.pi * (1.7 * r + 0.9 * c) + phase) \
        + 0.25 * np.cos(2 * np.pi * (0.6 * r - 2.3 * c) + 2.0 * phase)
    return np.asfortranarray((lo + (hi - lo) * s).astype(np.float32))


def make_parameters(nrow: int, ncol: int, perturb: float = 0.0) -> dict:
    """Spatially smooth distributed parameters; ``perturb`` scales them (used for qobs)."""
    p = {k: np.full((nrow, ncol), v, dtype=np.float32, order="F") for k, v in PARAM_DEFAULTS.items()}
    p["cp"] = _smooth(nrow, ncol, 50.0, 400.0, 0.3)
    p["cft"] = _smooth(nrow, ncol, 100.0, 800.0, 1.1)
    p["cst"] = _smooth(nrow, ncol, 100.0, 800.0, 2.3)
    p["exc"] = _smooth(nrow, ncol, -5.0, 1.0, 0.7)
    p["lr"] = _smooth(nrow, ncol, 2.0, 30.0, 1.9)
    p["ci"][:] = 1.0
    # vic-a (md_vic_operator.f90): soil-layer capacities, infiltration shape, conductivity, baseflow curve
    p["b"] = _smooth(nrow, ncol, 0.1, 0.6, 2.9)
    p["cusl1"] = _smooth(nrow, ncol, 50.0, 200.0, 3.7)
    p["cusl2"] = _smooth(nrow, ncol, 200.0, 800.0, 4.3)
    p["clsl"] = _smooth(nrow, ncol, 1000.0, 3000.0, 5.1)
    p["ks"] = _smooth(nrow, ncol, 5.0, 40.0, 5.9)
    p["ds"] = _smooth(nrow, ncol, 0.01, 0.1, 6.7)
    p["dsm"] = _smooth(nrow, ncol, 0.1, 1.0, 7.3)
    p["ws"] = _smooth(nrow, ncol, 0.5, 0.9, 8.1)
    if perturb:
        for k in ("cp", "cft", "cst", "exc", "lr", "b", "cusl1", "cusl2", "clsl", "ks", "dsm"):
            p[k] = np.asfortranarray((p[k] * np.float32(1.0 + perturb)).astype(np.float32))
    return p


STATE_WARM = dict(hi=0.2, hp=0.45, hft=0.35, hst=0.3, husl1=0.3, husl2=0.3, hlsl=0.3, hlr=0.8)


def make_states(nrow: int, ncol: int, warm: bool = False) -> dict:
    """Initial states: the reference defaults (cold start) or spun-up levels (warm).  A cold start keeps
    the transfer store near empty, where discharge is the difference of nearly equal numbers and the
    reference itself is only reproducible to ~1e-4 between its own builds (tests/golden noise_* keys)."""
    src = STATE_WARM if warm else STATE_DEFAULTS
    return {k: np.full((nrow, ncol), v, dtype=np.float32, order="F") for k, v in src.items()}
