"""poselib_amd/csrc/pl_libm.h against the host's libm (glibc 2.35 in this image): the device's cbrt is glibc's algorithm
(sysdeps/ieee754/dbl-64/s_cbrt.c) restated as plain IEEE operations, so that the cubic of the P3P / 7-point solvers
(PoseLib/misc/univariate.cc:82, 86, 107) rounds like the reference's host library.  The very header hipcc compiles is
compiled for the host (tests/hostmath) and compared bit for bit on 1.2e7 arguments: every bit pattern class, the
range the solvers use, 120 binades, subnormals."""
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
