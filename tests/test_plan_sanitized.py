"""CPU: the host half of the product that is plain C++ -- the routing-schedule builder smash_amd/csrc/sx_plan.cpp -- compiled
with AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool; the CPU build is where they can
run) and driven over meshes of every kind the tests use: E/SE/S synthetic catchments, full-D8 fields with masks and nodata, ragged
borders, tiles of a decomposition, group sizes from 64 to 512, and a flow-direction cycle (must be refused, not crash).  The
harness tests/csrc/sx_plan_check.cpp also verifies the schedule's invariants."""
import os
import subprocess

import numpy as np
import pytest

from smash_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "csrc", "sx_plan_check")


@pytest.fixture(scope="module")
def exe():
    src = [os.path.join(HERE, "csrc", "sx_plan_check.cpp"), os.path.join(HERE, "..", "smash_amd", "csrc", "sx_plan.cpp")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(s) for s in src):
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", EXE] + src,
                           capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return EXE


def _run(exe, m, group, rect=None, flwdir=None, sublevels=4):
    fd = np.asarray(m.flwdir if flwdir is None else flwdir, np.int32).reshape(-1, order="F")
    act = np.asarray(m.active_cell, np.int32).reshape(-1, order="F")
    head = f"{m.nrow} {m.ncol} {group} {int(rect is not None)} " + " ".join(str(v) for v in (rect or (0, 0, 0, 0)))
    txt = head + "\n" + " ".join(map(str, fd)) + "\n" + " ".join(map(str, act)) + f"\n{sublevels}\n"
    r = subprocess.run([exe], input=txt, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "VIOLATION" not in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (r.stdout, r.stderr[-1500:])
    return r.stdout.strip()


@pytest.mark.parametrize("group", [64, 128, 512])
def test_schedule_builder_under_sanitizers(exe, group):
    out = _run(exe, synth.make_mesh(96, 80, ng=2), group)
    assert out.startswith("ok cells 7680")
    for seed in (3, 11):
        m = synth.make_mesh_d8(60, 72, ng=2, seed=seed)            # all eight D8 codes, ragged active mask
        assert _run(exe, m, group).startswith(f"ok cells {m.nac}")
    m = synth.make_mesh(64, 64, ng=1, mask_corner=True)
    assert _run(exe, m, group).startswith(f"ok cells {m.nac}")


@pytest.mark.parametrize("group", [64, 256, 512])
def test_components_shorten_the_stages(exe, group):
    """Sub-levels (sx_plan.h "components"): the invariants hold for 1 (the old one-level-per-stage schedule), 2, 4 and 8 levels per
    super-step, and the groups' depths -- the super-steps a routing launch spends filling -- shrink accordingly."""
    m = synth.make_mesh(128, 128, ng=2)
    depth = {}
    for u in (1, 2, 4, 8):
        out = _run(exe, m, group, sublevels=u)
        assert out.startswith("ok cells 16384")
        depth[u] = (int(out.split(" deepest ")[1].split()[0]), int(out.split(" stagesum ")[1]))
    assert depth[2][0] < depth[1][0] and depth[4][0] < depth[2][0] and depth[8][0] <= depth[4][0]
    assert depth[4][1] < 0.5 * depth[1][1]
    md = synth.make_mesh_d8(60, 72, ng=2, seed=5)
    for u in (1, 3, 4):
        assert _run(exe, md, group, sublevels=u).startswith(f"ok cells {md.nac}")


def test_tiles_and_bad_meshes_under_sanitizers(exe):
    m = synth.make_mesh(64, 96, ng=1)
    tot = 0
    for rect in ((0, 32, 0, 48), (0, 32, 48, 96), (32, 64, 0, 48), (32, 64, 48, 96)):
        out = _run(exe, m, 128, rect)
        assert out.startswith("ok cells 1536")
        tot += int(out.split(" out ")[1].split()[0])
    assert tot > 0
    # a two-cell cycle in the flow directions: refused with SMASHX_E_MESH (-5), no crash
    fd = np.asarray(m.flwdir).copy(order="F")
    fd[10, 10], fd[10, 11] = 3, 7                                    # E <-> W
    assert _run(exe, m, 128, flwdir=fd).startswith(("rc -5", "ok"))
