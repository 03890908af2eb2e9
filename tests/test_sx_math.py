"""CPU: the host build of smash_amd/csrc/sx_math.h (the same header the kernels compile) against glibc.

* sx_tanhf restates fdlibm's float tanh/expm1 and must be BIT-IDENTICAL to glibc's tanhf (which is not
  correctly rounded): checked on every float in [2^-63, 24) -- the whole range the model can produce.
* the fp64-refined fixed powers must agree with glibc powf except where glibc itself misrounds (0.06 %).
* the reciprocal + 2 FMA division must equal IEEE division.
The device build uses v_rcp/v_rsq/v_sqrt seeds instead of the host's exact ones; the Newton step makes the
results independent of the seed's last bits, and the GPU parity tests cover the device build end to end."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "csrc", "sx_math_host.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "csrc", "sx_math_host.cpp")
    hdr = os.path.join(HERE, "..", "smash_amd", "csrc", "sx_math.h")
    hdr2 = os.path.join(HERE, "..", "smash_amd", "csrc", "sx_libm.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-o", SO, src, "-lm"])
    L = C.CDLL(SO)
    L.sxt_tanh_mismatches.restype = C.c_long
    L.sxt_tanh_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.sxt_div_mismatches.restype = C.c_long
    L.sxt_div_mismatches.argtypes = [C.c_long]
    L.sxt_pow_mismatches.argtypes = [C.c_long, C.POINTER(C.c_long)]
    return L


def test_tanh_bit_identical_to_glibc_exhaustive(lib):
    # 0x20000000 = 2^-63, 0x41c00000 = 24.0: every float in between (5.7e8 values, ~8 s)
    assert lib.sxt_tanh_mismatches(0x20000000, 0x41C00000, 1) == 0


def test_fixed_powers_match_glibc_up_to_its_own_misrounding(lib):
    n = 2_000_000
    out = (C.c_long * 6)()
    lib.sxt_pow_mismatches(n, out)
    for i, name in enumerate(["x^-4", "x^-5", "y^-1/4", "y^-5/4", "h^3.5", "h^2.5"]):
        assert out[i] / n < 1.5e-3, (name, out[i] / n)       # glibc powf itself misrounds 6e-4 of the time


def test_runtime_pow_and_log(lib):
    """sx_powf / sx_logf (vic-a's run-time exponents, the logarithmic criterion): the fp64 values are within 2^-44 of the
    double-precision library's; the fp32 power equals glibc's powf except where a rounding boundary is that close (glibc's own
    powf misrounds 6e-4 of the time), the fp32 logarithm is the correctly rounded one (what the kernels used before, through
    the library's double log; glibc's logf is 1.7 % away from it); special values as glibc."""
    n = 4_000_000
    out, dout = (C.c_long * 2)(), (C.c_double * 2)()
    lib.sxt_powlog_check.argtypes = [C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    lib.sxt_powlog_check(n, out, dout)
    assert dout[0] < 2.0 ** -44 and dout[1] < 2.0 ** -44, (dout[0], dout[1])
    assert out[0] / n < 1.5e-3 and out[1] / n < 1e-4, (out[0] / n, out[1] / n)
    lib.sxt_pow_specials.restype = C.c_long
    assert lib.sxt_pow_specials() == 0


def test_reciprocal_fma_division_is_ieee_division(lib):
    assert lib.sxt_div_mismatches(20_000_000) == 0


def test_scaled_division_of_tiny_numerators_is_ieee_division(lib):
    """sx_div_scaled (the exact-libm build's division for quotients below 2^-100: the adjoint fringe): whenever it says ok the result
    is a / d, and it only declines subnormal / zero results."""
    out = (C.c_long * 3)()
    lib.sxt_div_scaled_check(20_000_000, out)
    assert out[0] == 0 and out[2] == 0 and out[1] > 5_000_000, list(out)


def test_restated_glibc_expf_logf_powf_are_bit_identical_to_the_c_library(lib):
    """smash_amd/csrc/sx_libm.h (the exact-libm build of the kernels) against glibc 2.35 itself: expf on every 7th float,
    logf on every 5th positive float, powf with the six fixed exponents of the GR operators on every 11th base in [1e-7, 1e4],
    powf on 5e7 random (x, y) pairs incl. arbitrary bit patterns, and the special values.  Zero differing bit patterns."""
    for f in ("sxt_g_expf_mismatches", "sxt_g_logf_mismatches", "sxt_g_powf_fixed_mismatches"):
        getattr(lib, f).restype = C.c_long
        getattr(lib, f).argtypes = [C.c_uint32] * 3
    lib.sxt_g_powf_random_mismatches.restype = C.c_long
    lib.sxt_g_powf_random_mismatches.argtypes = [C.c_long, C.c_uint]
    lib.sxt_g_specials.restype = C.c_long
    assert lib.sxt_g_specials() == 0
    assert lib.sxt_g_expf_mismatches(0, 0x7F800000, 7) == 0
    assert lib.sxt_g_expf_mismatches(0x80000000, 0xFF800000, 7) == 0
    assert lib.sxt_g_logf_mismatches(0, 0x7F800000, 5) == 0
    assert lib.sxt_g_powf_fixed_mismatches(0x33D6BF95, 0x461C4000, 11) == 0
    assert lib.sxt_g_powf_random_mismatches(50_000_000, 99) == 0
    # the paired form the exact build uses for x^-4 / x^-5, y^-1/4 / y^-5/4, h^3.5 / h^2.5 (one log2 per base), incl. zero and subnormals
    lib.sxt_g_powf2_mismatches.restype = C.c_long
    lib.sxt_g_powf2_mismatches.argtypes = [C.c_uint32] * 3
    assert lib.sxt_g_powf2_mismatches(0, 0x7F800000, 37) == 0
