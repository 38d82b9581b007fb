"""poselib_amd/csrc/pl_libm.h against the host's libm (glibc 2.35 in this image): cbrt, acos, cos (and sin) on the device
are glibc's algorithms (sysdeps/ieee754/dbl-64/s_cbrt.c, e_asin.c, s_sin.c) restated as explicit IEEE operation sequences,
so that the cubics of the P3P / 7-point solvers (PoseLib/misc/univariate.cc:82, 86, 107-124) and the quaternion
exponential of the refiners round like the reference's host library.  The very header hipcc compiles is compiled for the
host (tests/hostmath) and compared BIT FOR BIT with libm: cbrt on 1.2e7 arguments (every bit pattern class, 120 binades,
subnormals), acos on 2e7 (all of (-1, 1), the neighbourhoods of +-1 and of 0, the edges of its piecewise expansion),
cos and sin on 1.5e7 each.  acos / cos / sin follow the variant glibc selects on hosts with FMA (as this image's and the
GPU box's hosts are): on a host without FMA libm itself takes another code path and the last bit may differ."""
import ctypes as C
import math

import hostmath_lib as HM


def test_cbrt_is_bit_identical_to_glibc():
    L = HM.lib()
    L.hm_cbrt_mismatches.restype = C.c_uint64
    L.hm_cbrt_mismatches.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.hm_cbrt.restype = C.c_double
    L.hm_cbrt.argtypes = [C.c_double]
    for mode, count in ((0, 3_000_000), (1, 4_000_000), (2, 4_000_000), (3, 1_000_000)):
        bad_x = C.c_double(0.0)
        bad = L.hm_cbrt_mismatches(count, 11 + mode, mode, C.byref(bad_x))
        assert bad == 0, (mode, bad, bad_x.value.hex())
    libm = C.CDLL("libm.so.6")
    libm.cbrt.restype = C.c_double
    libm.cbrt.argtypes = [C.c_double]
    for x in (27.0, -8.0, 1.0, 0.5, 1e-300, -1e300, 5e-324):  # (glibc: cbrt(27) = 3.0000000000000004 - faithful, not exact)
        assert L.hm_cbrt(x) == libm.cbrt(x), x
    assert math.copysign(1.0, L.hm_cbrt(-0.0)) == -1.0 and L.hm_cbrt(0.0) == 0.0
    assert L.hm_cbrt(math.inf) == math.inf and L.hm_cbrt(-math.inf) == -math.inf and math.isnan(L.hm_cbrt(math.nan))


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return True


def test_acos_cos_sin_are_bit_identical_to_glibc():
    import pytest

    if not _has_fma():
        pytest.skip("host without FMA: glibc runs its sse2 variants of acos / cos / sin here")
    L = HM.lib()
    L.hm_libm_mismatches.restype = C.c_uint64
    L.hm_libm_mismatches.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]
    for which, name, count in ((0, "acos", 20_000_000), (1, "cos", 15_000_000), (2, "sin", 15_000_000)):
        bad_x = C.c_double(0.0)
        bad = L.hm_libm_mismatches(which, count, 3 + which, C.byref(bad_x))
        assert bad == 0, (name, bad, bad_x.value.hex())


def test_sincos_is_bit_identical_to_glibc():
    """pl_sincos = what quat_exp's cos / sin pair IS in the reference's gcc build (one sincos() call; libm's sincos has no FMA
    variant and differs from sin() / cos() of the same host in the last bit for ~0.1 % of the arguments): 2*10^7 arguments in
    [-6, 6], [0, 0.9], [-2.4, 2.4] and down to 2^-40, every bit of both results"""
    L = HM.lib()
    L.hm_sincos_mismatches.restype = C.c_uint64
    L.hm_sincos_mismatches.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]
    bad_x = C.c_double(0.0)
    bad = L.hm_sincos_mismatches(20_000_000, 11, C.byref(bad_x))
    assert bad == 0, (bad, bad_x.value.hex())


def test_device_form_of_the_nielsen_cube_against_glibc_pow():
    """pl_refine.h lm_cube_fma - what the DEVICE computes for std::pow(2 rho - 1, 3) of the LM's Nielsen update
    (lm_impl.h:124); the host test build calls pow itself.  Two-product FMA form, nearly correctly rounded: it must equal
    glibc's pow(x, 3) for >= 99.9 % of the arguments the update can see (|2 rho - 1| <= a few) and never be off by more than
    one ulp (the form used to be exercised on the GPU only)."""
    import numpy as np

    L = HM.lib()
    rs = np.random.RandomState(12)
    x = np.r_[rs.uniform(-1.0, 1.0, 1_000_000), rs.uniform(-4.0, 4.0, 500_000), 10.0 ** rs.uniform(-8, 3, 200_000) * rs.choice([-1, 1], 200_000),
              [0.0, 1.0, -1.0, 0.5, 1 / 3, np.inf, -np.inf, 1e200, 1e-200]]
    out = np.zeros_like(x)
    L.hm_lm_cube(x.ctypes.data_as(C.c_void_p), C.c_uint64(x.size), out.ctypes.data_as(C.c_void_p))
    want = np.array([math.pow(v, 3) if abs(v) < 1e100 else v * v * v for v in x])
    same = out == want
    assert same.mean() > 0.999, same.mean()
    bad = ~same & np.isfinite(want)
    assert (np.abs(out[bad] - want[bad]) <= np.spacing(np.abs(want[bad]))).all()
