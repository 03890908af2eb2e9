"""GPU: the exact-libm build (libsmashx_exact.so, -DSX_EXACT_LIBM=1).  The default build computes the model's powers,
exponentials and logarithms correctly rounded, which glibc's float functions are not in ~6e-4 of calls (1.7 % for logf); that
rounding difference is the whole distance between the HIP path and the reference.  This build proves it: with glibc 2.35's own
algorithms restated (smash_amd/csrc/sx_libm.h, bit-identical to the C library on ~10^9 arguments: tests/test_sx_math.py) and
IEEE divisions, every forward output must be BIT-IDENTICAL to the reference Fortran's golden vectors and every gradient field
within the STRICT 1e-6 of BASELINE.json (tools/parity_table.py --assert-exact).  A separate process: the library is chosen at
import time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_exact_build_reproduces_the_reference():
    env = dict(os.environ, SMASHX_EXACT_LIBM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_table.py"), "--assert-exact"], env=env,
                       capture_output=True, text=True, timeout=1500)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_exact_build_is_bit_identical_on_the_edge_cases():
    """tests/test_gpu_parity.py::test_edge_cases_vs_oracle (ragged sizes, one long data gap, a single cell, no gauge) has no golden
    vector to take a noise-aware bar from and asserts 2e-5 in the default build; under the exact-libm build the same test asserts
    BIT-IDENTITY with the oracle (which is bit-identical to the reference) on every output and gradient field."""
    env = dict(os.environ, SMASHX_EXACT_LIBM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_edge_cases_vs_oracle or test_real_river_network_vs_oracle", "-p", "no:cacheprovider"], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    sys.stdout.write(r.stdout[-3000:])
    # (five edge cases + the largest basin of the reference's D8 raster of France -- a real river network of 139 742 cells -- on plain and on staging rows)
    assert r.returncode == 0 and "7 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
