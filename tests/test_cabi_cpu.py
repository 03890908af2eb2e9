"""CPU: the C-ABI library loads and exports every symbol include/smashx.h declares; host-side logic
that needs no GPU (argument checks, the routing schedule builder via plan creation failing cleanly)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from smash_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "smashx.h")).read()
    declared = set(re.findall(r"\b(smashx_[a-z_]+)\s*\(", hdr))
    L = _lib.lib()
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the product must refuse to run (no CPU path)."""
    import smash_amd
    from smash_amd import _lib, synth
    if _lib.lib().smashx_device_count() > 0:
        pytest.skip("a HIP device is present")
    m = synth.make_mesh(8, 8, ng=1)
    setup = smash_amd.SetupDT(0, 1, structure="gr-a", ntime_step=4)
    mesh = smash_amd.MeshDT.from_synth(setup, m)
    with pytest.raises(smash_amd.SmashxError) as e:
        smash_amd.Solver(setup, mesh)
    assert e.value.code == _lib.E_NODEVICE


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under smash_amd/ may reference it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "smash_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
