"""CPU: the real river network of the bench (bench.py --mesh france:*): the reference's 1-km D8 raster of France as a data fixture
(tests/golden/mesh/france_d8.npz, tests/golden/make_france_d8.py) -> synth.make_mesh_france -> the routing schedule (smashx_tile_probe,
host only)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _probe(m, group=512):
    import smash_amd
    from smash_amd import _lib
    from smash_amd.solver import make_config
    setup = smash_amd.SetupDT(0, 0, structure="gr-b", ntime_step=16)
    mesh = smash_amd.MeshDT.from_synth(setup, m)
    mesh.ng = 0
    cfg = make_config(setup, mesh, group_size=group)
    cfg.ng = 0
    keep = [np.asfortranarray(m.flwdir, np.int32), np.asfortranarray(m.flwacc, np.int32), np.asfortranarray(m.active_cell, np.int32)]
    cm = _lib.Mesh(keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, None, None, None)
    info = (C.c_int * 8)()
    _lib.check(_lib.lib().smashx_tile_probe(C.byref(cfg), C.byref(cm), info, None, None, None, None, 0))
    return dict(zip("cells rounds groups slots xseries deepest n_out n_in".split(), list(info)))


def test_france_raster_gives_an_acyclic_forest_the_schedule_accepts():
    from smash_amd import synth
    m = synth.make_mesh_france("all")
    assert (m.nrow, m.ncol) == (1125, 1200) and m.nac == 956614          # 956 958 cells with a direction, 344 of them on closed loops
    fd, act = np.asarray(m.flwdir), np.asarray(m.active_cell)
    assert set(np.unique(fd[act == 1])) == set(range(1, 9))              # all eight D8 codes
    assert int(np.asarray(m.flwacc).max()) == 139742                     # the largest basin
    p = _probe(m)
    assert p["cells"] == m.nac and p["rounds"] == 6 and p["deepest"] == 226 and p["n_out"] == 0 and p["n_in"] == 0
    # every gauge sits on an active cell, the first one on the largest outlet
    gp = np.asarray(m.gauge_pos)
    assert all(act[r, c] == 1 for r, c in gp) and int(np.asarray(m.flwacc)[gp[0, 0], gp[0, 1]]) == 139742


def test_largest_basins_are_upstream_closed():
    from smash_amd import synth
    m1, m3 = synth.make_mesh_france(1), synth.make_mesh_france(3)
    assert m1.nac == 139742 and m3.nac == 139742 + 117137 + 97016
    ds, ok = synth.downstream_index(m1.flwdir, m1.active_cell)
    assert int((ok & (ds < 0)).sum()) == 1                               # one basin, one outlet
    assert _probe(m1)["cells"] == m1.nac
