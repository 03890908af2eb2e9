"""CPU: the library's L-BFGS-B (smash_amd/csrc/sx_lbfgsb.cpp: host C++ with its own threads, windowed breakpoint heaps, lazily
allocated history) under AddressSanitizer + UndefinedBehaviorSanitizer and, separately, ThreadSanitizer -- GPU sanitizers are not
available on the pool, and this part of the product never touches the GPU.  The harness tests/csrc/sx_lbfgsb_check.cpp drives it
through the C entry points of include/smashx.h and checks the invariants (box, monotone decrease, convergence)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build(kind, flags):
    exe = os.path.join(HERE, "csrc", "sx_lbfgsb_check_" + kind)
    src = [os.path.join(HERE, "csrc", "sx_lbfgsb_check.cpp"), os.path.join(ROOT, "smash_amd", "csrc", "sx_lbfgsb.cpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(s) for s in src):
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include")] + flags + ["-o", exe] + src,
                           capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return exe


def _run(exe, n, m, seed, maxiter):
    r = subprocess.run([exe, str(n), str(m), str(seed), str(maxiter)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "VIOLATION" not in r.stdout and "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, (r.stdout, r.stderr[-1500:])
    return r.stdout.strip()


@pytest.mark.parametrize("n,m,maxiter", [(1, 5, 100), (2, 1, 200), (7, 3, 300), (400, 10, 2000), (5000, 17, 2000)])
def test_lbfgsb_under_address_and_ub_sanitizers(n, m, maxiter):
    exe = _build("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    for seed in (1, 2):
        out = _run(exe, n, m, seed, maxiter)
        assert out.startswith(f"ok n {n} ") and "CONVERGENCE" in out, out


def test_lbfgsb_threaded_sizes_under_sanitizers():
    """Above 2^18 variables the n-vector passes run on several threads: address/UB and thread sanitizers over a few iterations that
    cross breakpoints (windowed heaps), store and recycle history columns (m = 3 < iterations)."""
    n = (1 << 18) + 12345
    out = _run(_build("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]), n, 3, 5, 8)
    assert out.startswith(f"ok n {n} it 8")
    out = _run(_build("tsan", ["-fsanitize=thread"]), n, 3, 5, 6)
    assert out.startswith(f"ok n {n} it 6")
