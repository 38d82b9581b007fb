// poselib_amd — libm functions whose LAST BIT matters for parity, restated so that the device rounds like the
// reference's host library instead of like ocml.
//
// The reference calls std::cbrt / std::acos / std::cos in the cubic solvers of P3P and the 7-point solver
// (PoseLib/misc/univariate.cc:82, 86, 107-124).  Those come from the C library of the machine the reference runs on -
// glibc 2.35 (libm.so.6, sysdeps/ieee754/dbl-64) in this image - a third-party dependency that is not vendored in
// /root/reference.  glibc's results are faithfully rounded, not correctly rounded, so "the same function from another
// vendor" (ocml) differs in the last bit for a sizeable share of the arguments; the models of a minimal sample then
// differ in their last bits, which decides ties between models of the same sample drawn twice (small N).
//
//   pl_cbrt : glibc's algorithm (s_cbrt.c, unchanged since glibc 2.0): frexp, a degree-6 polynomial for the
//             mantissa's cube root, one Halley step, 2^(e mod 3 / 3) from a table, ldexp.  Plain IEEE operations in
//             a fixed order, no FMA (the x86-64 build of s_cbrt.c has no FMA variant) => bit-identical to the host's
//             cbrt for every argument; tests/test_libm_vs_glibc.py checks 2e7 arguments against the host libm.
//   acos / cos : glibc's IBM Accurate Mathematical Library routines are table driven (asincos.tbl, sincostab) and the
//             variant selected on AVX2 hosts (__acos_fma, __cos_fma) is compiled with FMA contraction; they are NOT
//             restated here - the device keeps ocml's (DESIGN.md 5 lists this as the remaining last-bit difference).
#pragma once
#include "pl_math.h"

namespace pl {

PL_HD double pl_cbrt(double x) {
    // frexp of |x|: xm in [0.5, 1), |x| = xm 2^xe  (glibc's frexp sets xe = 0 for 0, inf, NaN)
    uint64_t bits;
    const double ax = fabs(x);
    __builtin_memcpy(&bits, &ax, 8);
    int ex = (int)(bits >> 52);
    if (ex == 0x7ff || ax == 0.0)
        return x + x; // s_cbrt.c: "if (xe == 0 && fpclassify (x) <= FP_ZERO) return x + x"
    int xe;
    if (ex == 0) { // subnormal: normalise first (frexp multiplies by 2^54)
        const double sc = ax * 18014398509481984.0;
        __builtin_memcpy(&bits, &sc, 8);
        ex = (int)(bits >> 52);
        xe = ex - 1022 - 54;
    } else {
        xe = ex - 1022;
    }
    bits = (bits & 0x000fffffffffffffull) | 0x3fe0000000000000ull;
    double xm;
    __builtin_memcpy(&xm, &bits, 8);

    const double u = (0.354895765043919860 +
                      ((1.50819193781584896 -
                        ((2.11499494167371287 -
                          ((2.44693122563534430 -
                            ((1.83469277483613086 - (0.784932344976639262 - 0.145263899385486377 * xm) * xm) * xm)) *
                           xm)) *
                         xm)) *
                       xm));
    const double t2 = u * u * u;
    // factor[2 + xe % 3] with C's truncating remainder: {1 / 2^(2/3), 1 / 2^(1/3), 1, 2^(1/3), 2^(2/3)}
    const int rem = xe % 3;
    const double cbrt2 = 1.2599210498948731648, sqr_cbrt2 = 1.5874010519681994748;
    const double f = rem == -2 ? 1.0 / sqr_cbrt2 : rem == -1 ? 1.0 / cbrt2 : rem == 0 ? 1.0 : rem == 1 ? cbrt2 : sqr_cbrt2;
    const double ym = u * (t2 + 2.0 * xm) / (2.0 * t2 + xm) * f;
    // ldexp(+-ym, xe / 3): ym in [0.39, 1.6), the result of a finite argument never leaves the normal range
    const int q = xe / 3;
    double r = x > 0.0 ? ym : -ym;
    uint64_t rb;
    __builtin_memcpy(&rb, &r, 8);
    rb += (uint64_t)(int64_t)q << 52;
    __builtin_memcpy(&r, &rb, 8);
    return r;
}

} // namespace pl
