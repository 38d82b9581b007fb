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
//   pl_cos  : glibc's IBM Accurate Mathematical Library routine (s_sin.c: __cos, do_cos, do_sin, reduce_sincos): the
//             argument is split at multiples of 1/128, sin / cos of the grid point come from a double-double table
//             (regenerated from first principles by scripts/gen_libm_tables.py), the remainder from short polynomials.
//   pl_acos : e_asin.c __ieee754_acos: a polynomial below 1/8, piecewise expansions of asin around the midpoints of
//             intervals of width 2^-8 up to 31/32 (asincos.tbl), and 2 asin(sqrt((1 - |x|) / 2)) with an inline
//             double-double square root above.
//             Both are restated as the operation sequence of the variant glibc selects on hosts with FMA
//             (__cos_fma / __acos_fma: the same source compiled with -mfma, i.e. with GCC's contractions) - every
//             fused multiply-add below is one in that binary, everything else is a separately rounded operation.
//             tests/test_libm_vs_glibc.py checks both bit for bit against the host's libm (2e7 arguments each).
//             A host without FMA would run glibc's sse2 variant, whose results differ in the last bit for a small
//             share of the arguments - the oracle then differs from this header exactly as it differs from itself
//             on another machine.
#pragma once
#include "pl_defs.h"

#define PL_TABLE static constexpr
#include "pl_libm_tables.h"
#undef PL_TABLE

namespace pl {

PL_HD double pl_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PL_HD uint32_t pl_hi(double v) {
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    return (uint32_t)(b >> 32);
}
PL_HD uint32_t pl_lo(double v) {
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    return (uint32_t)b;
}

// ---- cos (s_sin.c) ---------------------------------------------------------------------------------------------------
namespace libm_detail {
constexpr double kSn3 = -1.66666666666664880952546298448555E-01, kSn5 = 8.33333214285722277379541354343671E-03;
constexpr double kCs2 = 4.99999999999999999999950396842453E-01, kCs4 = -4.16666666666664434524222570944589E-02,
                 kCs6 = 1.38888874007937613028114285595617E-03;
constexpr double kS1 = -0x1.5555555555555p-3, kS2 = 0x1.1111111110ECEp-7, kS3 = -0x1.A01A019DB08B8p-13,
                 kS4 = 0x1.71DE27B9A7ED9p-19, kS5 = -0x1.ADDFFC2FCDF59p-26;
constexpr double kBig = 0x1.8p45;                                        // 1.5 * 2^45: rounds to multiples of 2^-7
constexpr double kHp0 = 0x1.921FB54442D18p0, kHp1 = 0x1.1A62633145C07p-54; // pi / 2 in two parts

// FMA = true: the operation sequence of glibc's -mfma variants (__sin_fma / __cos_fma: every mad below is ONE fused operation in
// that binary); FMA = false: the same source without contractions (the sse2 build - what `sincos` is on x86-64: it has no FMA variant)
template <bool FMA> PL_HD double mad(double a, double b, double c) {
    if constexpr (FMA)
        return pl_fma(a, b, c);
    else
        return a * b + c;
}
template <bool FMA = true> PL_HD double do_cos(double x, double dx) {
    if (x < 0)
        dx = -dx;
    const double u = kBig + fabs(x);
    const int k = (int)pl_lo(u);
    x = fabs(x) - (u - kBig) + dx;
    const double xx = x * x;
    const double s = mad<FMA>(x * xx, mad<FMA>(xx, kSn5, kSn3), x);
    const double c = xx * mad<FMA>(xx, mad<FMA>(xx, kCs6, kCs4), kCs2);
    const double sn = kSinCosTab[k][0], ssn = kSinCosTab[k][1], cs = kSinCosTab[k][2], ccs = kSinCosTab[k][3];
    const double cor = mad<FMA>(-s, sn, mad<FMA>(-c, cs, mad<FMA>(-s, ssn, ccs)));
    return cs + cor;
}
template <bool FMA = true> PL_HD double do_sin(double x, double dx) {
    const double xold = x;
    if (fabs(x) < 0.126) { // TAYLOR_SIN
        const double xx = x * x;
        const double p = mad<FMA>(xx, mad<FMA>(xx, mad<FMA>(xx, mad<FMA>(xx, kS5, kS4), kS3), kS2), kS1);
        const double t = mad<FMA>(xx, mad<FMA>(p, x, -(0.5 * dx)), dx);
        return x + t;
    }
    if (x <= 0)
        dx = -dx;
    const double u = kBig + fabs(x);
    const int k = (int)pl_lo(u);
    x = fabs(x) - (u - kBig);
    const double xx = x * x;
    const double s = x + mad<FMA>(x * xx, mad<FMA>(xx, kSn5, kSn3), dx);
    const double c = mad<FMA>(dx, x, xx * mad<FMA>(xx, mad<FMA>(xx, kCs6, kCs4), kCs2));
    const double sn = kSinCosTab[k][0], ssn = kSinCosTab[k][1], cs = kSinCosTab[k][2], ccs = kSinCosTab[k][3];
    const double cor = mad<FMA>(s, cs, mad<FMA>(-c, sn, mad<FMA>(s, ccs, ssn)));
    return copysign(sn + cor, xold);
}
} // namespace libm_detail

// |x| < 105414350 (beyond that glibc switches to another reduction; the solvers' arguments are below 4.2)
PL_HD double pl_cos(double x) {
    using namespace libm_detail;
    const uint32_t k = pl_hi(x) & 0x7fffffffu;
    if (k < 0x3e400000u) // |x| < 2^-27
        return 1.0;
    if (k < 0x3feb6000u) // |x| < 0.855469
        return do_cos(x, 0.0);
    if (k < 0x400368fdu) { // |x| < 2.426265
        const double y = kHp0 - fabs(x);
        const double a = y + kHp1;
        const double da = (y - a) + kHp1;
        return do_sin(a, da);
    }
    if (k < 0x419921FBu) { // reduce_sincos: x - n pi / 2 with pi / 2 in four parts
        constexpr double hpinv = 0x1.45F306DC9C883p-1, toint = 0x1.8p52;
        constexpr double mp1 = 0x1.921FB58000000p0, mp2 = -0x1.DDE973C000000p-27, pp3 = -0x1.CB3B398000000p-55,
                         pp4 = -0x1.d747f23e32ed7p-83;
        const double t = pl_fma(x, hpinv, toint);
        const double xn = t - toint;
        const double y = pl_fma(-xn, mp2, pl_fma(-xn, mp1, x));
        int n = (int)(pl_lo(t) & 3u);
        double t1 = xn * pp3;
        const double t2 = y - t1;
        double db = (y - t2) - t1;
        t1 = xn * pp4;
        const double b = t2 - t1;
        db += (t2 - b) - t1;
        n = n + 1;
        const double r = (n & 1) ? do_cos(b, db) : do_sin(b, db);
        return (n & 2) ? -r : r;
    }
    return cos(x); // not reached by the solvers
}

// sin for the arguments quat_exp produces (|x| < 2.426265; s_sin.c __sin)
PL_HD double pl_sin(double x) {
    using namespace libm_detail;
    const uint32_t k = pl_hi(x) & 0x7fffffffu;
    if (k < 0x3e500000u) // |x| < 2^-26
        return x;
    if (k < 0x3feb6000u)
        return do_sin(x, 0.0);
    if (k < 0x400368fdu) {
        const double t = kHp0 - fabs(x);
        return copysign(do_cos(t, kHp1), x);
    }
    return sin(x); // not reached
}

// sin and cos of the SAME argument: s_sincos.c __sincos, the sse2 build (libm exports one `sincos`, without an FMA variant).
// quat_exp (misc/quaternion.h:73-96) calls std::cos and std::sin on theta / 2; GCC merges the two calls into one sincos()
// at -O1 and above, so the reference's Release build - and the oracle - step their quaternions with THESE roundings, which
// differ from sin() / cos() of the same machine in the last bit for 0.1 % (sin) and 0.03 % (cos) of the arguments below 0.86
// (measured on 2*10^7 arguments; found by scripts/soak_intrinsics.py as a one-ulp difference of an LM step).
// tests/test_libm_vs_glibc.py checks it bit for bit against the host's sincos().
PL_HD void pl_sincos(double x, double &sn, double &cs) {
    using namespace libm_detail;
    const uint32_t k = pl_hi(x) & 0x7fffffffu;
    if (k < 0x400368fdu) {
        if (k < 0x3e400000u) { // |x| < 2^-27
            sn = x;
            cs = 1.0;
            return;
        }
        if (k < 0x3feb6000u) { // |x| < 0.855469
            sn = do_sin<false>(x, 0.0);
            cs = do_cos<false>(x, 0.0);
            return;
        }
        const double y = kHp0 - fabs(x); // |x| < 2.426265
        const double a = y + kHp1;
        const double da = (y - a) + kHp1;
        sn = copysign(do_cos<false>(a, da), x);
        cs = do_sin<false>(a, da);
        return;
    }
    if (k < 0x419921FBu) { // reduce_sincos + do_sincos(n), do_sincos(n + 1)
        constexpr double hpinv = 0x1.45F306DC9C883p-1, toint = 0x1.8p52;
        constexpr double mp1 = 0x1.921FB58000000p0, mp2 = -0x1.DDE973C000000p-27, pp3 = -0x1.CB3B398000000p-55,
                         pp4 = -0x1.d747f23e32ed7p-83;
        const double t = x * hpinv + toint;
        const double xn = t - toint;
        const double y = (x - xn * mp1) - xn * mp2;
        const int n = (int)(pl_lo(t) & 3u);
        double t1 = xn * pp3;
        const double t2 = y - t1;
        double db = (y - t2) - t1;
        t1 = xn * pp4;
        const double b = t2 - t1;
        db += (t2 - b) - t1;
        auto pick = [&](int m) {
            const double r = (m & 1) ? do_cos<false>(b, db) : do_sin<false>(b, db);
            return (m & 2) ? -r : r;
        };
        sn = pick(n);
        cs = pick(n + 1);
        return;
    }
    sn = sin(x); // (rotation increments of an LM step never get here)
    cs = cos(x);
}

// ---- acos (e_asin.c __ieee754_acos, FMA variant) -----------------------------------------------------------------------
PL_HD double pl_acos(double x) {
    using namespace libm_detail;
    // (uasncs.h: the odd polynomial of asin near 0, also used for asin(sqrt(z)) near |x| = 1)
    constexpr double f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4, f3 = 0x1.6db6dae42c0e4p-5,
                     f4 = 0x1.f1c7e04f4ad99p-6, f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6;
    const uint32_t hi = pl_hi(x);
    const int32_t m = (int32_t)hi;
    const uint32_t k = hi & 0x7fffffffu;
    if (k < 0x3c880000u) // |x| < 2^-55
        return kHp0;
    if (k < 0x3fc00000u) { // |x| < 1/8
        const double x2 = x * x;
        const double p = pl_fma(x2, pl_fma(x2, pl_fma(x2, pl_fma(x2, pl_fma(x2, f6, f5), f4), f3), f2), f1);
        const double r = kHp0 - x;
        const double e = ((kHp0 - r) - x) + kHp1;
        return r + pl_fma(-p, x * x2, e);
    }
    if (k < 0x3fef0000u) { // 1/8 <= |x| < 31/32: expansion of asin around the midpoint of the argument's interval
        int n, nc; // row start, number of coefficients
        if (k < 0x3fd00000u)
            n = 11 * (int)((k >> 15) & 0x1fu), nc = 6;
        else if (k < 0x3fe00000u)
            n = 11 * (int)((k >> 14) & 0x3fu) + 352, nc = 6;
        else if (k < 0x3fe80000u)
            n = 12 * (int)((k >> 13) & 0x7fu) + 1056, nc = 7;
        else if (k < 0x3fed8000u)
            n = 13 * (int)((k >> 13) & 0x7fu) + 992, nc = 8;
        else if (k < 0x3fee8000u)
            n = 14 * (int)((k >> 13) & 0x7fu) + 884, nc = 9;
        else
            n = 15 * (int)((k >> 13) & 0x7fu) + 768, nc = 10;
        const double *T = kAsinRows + n;
        const double ax = (m > 0) ? x : -x;
        const double xx = ax - T[0];
        double p = T[nc];
        for (int j = nc - 1; j >= 2; --j)
            p = pl_fma(xx, p, T[j]);
        p = pl_fma(xx * xx, p, T[nc + 1]); // + asin(x0) low part
        const double t = pl_fma(xx, T[1], p);
        const double y = T[nc + 2]; // asin(x0) high part
        if (m > 0)
            return (kHp1 - t) + (kHp0 - y);
        return (t + kHp1) + (y + kHp0);
    }
    if (k < 0x3ff00000u) { // 31/32 <= |x| < 1: 2 asin(sqrt(z)), z = (1 - |x|) / 2, sqrt(z) as y + cc
        constexpr double rt0 = 0x1.fffffffecc1ddp-1, rt1 = 0x1.fffffff757304p-2, rt2 = 0x1.800496769c91ap-2,
                         rt3 = 0x1.4006318d1dab9p-2, t27 = 134217728.0; // (uasncs.h: inverse square root refinement)
        const double z = ((m > 0) ? (1.0 - x) : (x + 1.0)) * 0.5;
        const uint32_t zh = pl_hi(z);
        const int e = (int)(zh >> 21); // (sign bit clear)
        uint64_t pw = (uint64_t)(1023 + (511 - e)) << 52; // powtwo[511 - e] = 2^(511 - e)
        double two;
        __builtin_memcpy(&two, &pw, 8);
        double t = kInvRoot[(zh >> 14) & 0x7fu] * two;
        const double r = pl_fma(-(t * t), z, 1.0);
        t = pl_fma(r, pl_fma(r, pl_fma(r, rt3, rt2), rt1), rt0) * t;
        const double c = z * t;
        const double h = pl_fma(-(t * 0.5), c, 1.5);
        const double w = pl_fma(c, t27, c);
        const double y = pl_fma(-t27, c, w);
        const double den = pl_fma(h, c, y);
        const double cc = pl_fma(-y, y, z) / den;
        const double p = pl_fma(z, pl_fma(z, pl_fma(z, pl_fma(z, pl_fma(z, f6, f5), f4), f3), f2), f1) * z;
        const double cor = p * (y + cc);
        if (m > 0) {
            const double s = (cc + cor) + y;
            return s + s;
        }
        const double s = ((kHp1 - cc) - cor) + (kHp0 - y);
        return s + s;
    }
    if (k == 0x3ff00000u && pl_lo(x) == 0u) // |x| = 1
        return (m > 0) ? 0.0 : 0x1.921FB54442D18p1;
    if (k > 0x7ff00000u || (k == 0x7ff00000u && pl_lo(x) != 0u))
        return x + x; // NaN
    return (x - x) / (x - x); // |x| > 1
}

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
