// poselib_amd — register-resident P3P for one lane (one RANSAC iteration per lane).
//
// Algorithm: Ding et al. CVPR'23 as used by the reference — one real root of a cubic
// (PoseLib/misc/univariate.cc:74-92), the degenerate conic it selects is split into two lines
// (PoseLib/solvers/p3p_common.h:31-71), each line gives a quadratic in the depth ratio
// (PoseLib/solvers/p3p.cc:129-195), depths are polished by <= 5 Newton steps
// (p3p_common.h:74-94) and R = Y * X^-1 (p3p.cc:121-122,162-164).  All branches are kept so the
// set and ORDER of returned solutions equals the reference's; cbrt / acos / cos are glibc's algorithms (pl_libm.h,
// bit-identical to the host's libm), so the device models equal the CPU path's to the bit.
// The coefficient block of p3p() below follows PoseLib/solvers/p3p.cc:77-101 operation for operation (BSD-3 source):
// the arithmetic ORDER is what bit-parity with the reference requires, so that part is a transliteration by design.
#pragma once
#include "pl_math.h"

namespace pl {

// univariate.cc:74-92.  Returns true when the cubic has exactly one real root.
PL_HD bool cubic_one_real_root(double c2, double c1, double c0, double &root) {
    const double a = c1 - c2 * c2 / 3.0;
    double b = (2.0 * c2 * c2 * c2 - 9.0 * c2 * c1) / 27.0 + c0;
    double c = b * b / 4.0 + a * a * a / 27.0;
    if (c != 0) {
        if (c > 0) {
            c = sqrt(c);
            b *= -0.5;
            root = pl_cbrt(b + c) + pl_cbrt(b - c) - c2 / 3.0;
            return true;
        }
        c = 3.0 * b / (2.0 * a) * sqrt(-3.0 / a);
        root = 2.0 * sqrt(-a / 3.0) * pl_cos(pl_acos(c) / 3.0) - c2 / 3.0;
        return false;
    }
    root = -c2 / 3.0 + (a != 0 ? (3.0 * b / a) : 0);
    return false;
}

// p3p_common.h:7-29
PL_HD bool monic_quadratic_roots(double b, double c, double &r1, double &r2) {
    const double thr = -1.0e-12;
    const double v = b * b - 4.0 * c;
    if (v < thr) {
        r1 = r2 = -0.5 * b;
        return v >= 0;
    }
    if (v > thr && v < 0.0) {
        r1 = -0.5 * b;
        r2 = -2;
        return true;
    }
    const double y = sqrt(v);
    if (b < 0) {
        r1 = 0.5 * (-b + y);
        r2 = 0.5 * (-b - y);
    } else {
        r1 = 2.0 * c / (-b + y);
        r2 = 2.0 * c / (-b - y);
    }
    return true;
}

// p3p_common.h:74-94
PL_HD void polish_depths(double &l1, double &l2, double &l3, double a12, double a13, double a23, double b12,
                         double b13, double b23) {
    for (int it = 0; it < 5; ++it) {
        const double r1 = (l1 * l1 - 2.0 * l1 * l2 * b12 + l2 * l2 - a12);
        const double r2 = (l1 * l1 - 2.0 * l1 * l3 * b13 + l3 * l3 - a13);
        const double r3 = (l2 * l2 - 2.0 * l2 * l3 * b23 + l3 * l3 - a23);
        if (fabs(r1) + fabs(r2) + fabs(r3) < 1e-10)
            return;
        const double x11 = l1 - l2 * b12, x12 = l2 - l1 * b12;
        const double x21 = l1 - l3 * b13, x23 = l3 - l1 * b13;
        const double x32 = l2 - l3 * b23, x33 = l3 - l2 * b23;
        const double dj = 0.5 / (x11 * x23 * x32 + x12 * x21 * x33);
        l1 += (-x23 * x32 * r1 - x12 * x33 * r2 + x12 * x23 * r3) * dj;
        l2 += (-x21 * x33 * r1 + x11 * x33 * r2 - x11 * x23 * r3) * dj;
        l3 += (x21 * x32 * r1 - x11 * x32 * r2 - x12 * x21 * r3) * dj;
    }
}

struct P3PSolution {
    Mat3 R; // as produced by the solver (before the R -> q -> R round trip)
    Vec3 t;
};

// The solver in two halves, so that the generator can run the second one on full wavefronts of CANDIDATES (kernels.hip):
//   p3p_front  everything up to the candidate depth triples (d0, d1, d2) of the <= 4 solutions, in the reference's order
//              (both lines of the degenerate conic, both roots of each line's quadratic, sign tests; p3p.cc:129-160), and
//              what the second half needs of the sample: the relabelled bearings, X0, inverse(X01, X02, X01 x X02),
//              the squared side lengths and the cosines;
//   p3p_back   one candidate -> Newton polish of the depths (p3p_common.h:74-94), R = Y X^-1, t (p3p.cc:121-122, 162-164).
// No candidate is dropped by the second half, so the first half's count is the number of solutions.
struct P3PFront {
    Vec3 x0, x1, x2, X0;
    Mat3 XX;
    double a01, a02, a12, m01, m02, m12;
};

PL_HD int p3p_front(Vec3 x0, Vec3 x1, Vec3 x2, Vec3 X0, Vec3 X1, Vec3 X2, P3PFront &f, double (*cand)[3]) {
    Vec3 X01 = X0 - X1, X02 = X0 - X2, X12 = X1 - X2;
    double a01 = dot(X01, X01), a02 = dot(X02, X02), a12 = dot(X12, X12);

    // relabel so that |X1 - X2| is the longest side (p3p.cc:58-73)
    if (a01 > a02) {
        if (a01 > a12) {
            Vec3 tv = x0;
            x0 = x2, x2 = tv;
            tv = X0;
            X0 = X2, X2 = tv;
            const double ts = a01;
            a01 = a12, a12 = ts;
            X01 = -X12;
            X02 = -X02;
        }
    } else if (a02 > a12) {
        Vec3 tv = x0;
        x0 = x1, x1 = tv;
        tv = X0;
        X0 = X1, X1 = tv;
        const double ts = a02;
        a02 = a12, a12 = ts;
        X01 = -X01;
        X02 = X12;
    }

    const double a12d = 1.0 / a12;
    const double a = a01 * a12d, b = a02 * a12d;
    const double m01 = dot(x0, x1), m02 = dot(x0, x2), m12 = dot(x1, x2);

    const double m12sq = -m12 * m12 + 1.0;
    const double m02sq = -1.0 + m02 * m02;
    const double m01sq = -1.0 + m01 * m01;
    const double ab = a * b, bsq = b * b, asq = a * a;
    const double m013 = -2.0 + 2.0 * m01 * m02 * m12;
    const double bsqm12sq = bsq * m12sq;
    const double asqm12sq = asq * m12sq;
    const double abm12sq = 2.0 * ab * m12sq;

    const double k3_inv = 1.0 / (bsqm12sq + b * m02sq);
    const double k2 = k3_inv * ((-1.0 + a) * m02sq + abm12sq + bsqm12sq + b * m013);
    const double k1 = k3_inv * (asqm12sq + abm12sq + a * m013 + (-1.0 + b) * m01sq);
    const double k0 = k3_inv * (asqm12sq + a * m01sq);

    double s;
    const bool single_root = cubic_one_real_root(k2, k1, k0, s);

    // the degenerate conic C (symmetric) ...
    double c00 = -a + s * (1 - b);
    double c01 = -m02 * s;
    double c02 = a * m12 + b * m12 * s;
    double c11 = s + 1;
    double c12 = -m01;
    double c22 = -a - b * s + 1;
    // ... split into two lines (p3p_common.h:31-71).  A = negated adjugate of C (symmetric);
    // C is symmetric at this point so C(1,0)=c01 etc.
    const double A00 = c12 * c12 - c11 * c22;
    const double A11 = c02 * c02 - c00 * c22;
    const double A22 = c01 * c01 - c00 * c11;
    const double A01 = c01 * c22 - c02 * c12;
    const double A02 = c02 * c11 - c01 * c12;
    const double A12 = c00 * c12 - c02 * c01;
    Vec3 v;
    if (A00 > A11) {
        if (A00 > A22)
            v = v3(A00, A01, A02) / sqrt(A00);
        else
            v = v3(A02, A12, A22) / sqrt(A22);
    } else if (A11 > A22) {
        v = v3(A01, A11, A12) / sqrt(A11);
    } else {
        v = v3(A02, A12, A22) / sqrt(A22);
    }
    // de-symmetrised matrix: first column (p) and first row (q)
    const Vec3 pl0 = v3(c00, c01 + v.z, c02 - v.y); // column 0: (C00, C10 + v2, C20 - v1)
    const Vec3 pl1 = v3(c00, c01 - v.z, c02 + v.y); // row 0:    (C00, C01 - v2, C02 + v1)

    Mat3 XX;
    set_col(XX, 0, X01);
    set_col(XX, 1, X02);
    set_col(XX, 2, cross(X01, X02));
    f.XX = inverse3(XX);
    f.x0 = x0, f.x1 = x1, f.x2 = x2, f.X0 = X0;
    f.a01 = a01, f.a02 = a02, f.a12 = a12, f.m01 = m01, f.m02 = m02, f.m12 = m12;

    int n = 0;
    for (int i = 0; i < 2; ++i) {
        const Vec3 line = (i == 0) ? pl0 : pl1;
        const double p0 = line.x, p1 = line.y, p2 = line.z;
        const bool elim_d0 = fabs(p0) <= fabs(p1);
        double w0, w1, cb, cc;
        if (elim_d0) {
            w0 = -p0 / p1;
            w1 = -p2 / p1;
            const double ca = 1.0 / (w1 * w1 - b);
            cb = 2.0 * (b * m12 - m02 * w1 + w0 * w1) * ca;
            cc = (w0 * w0 - 2 * m02 * w0 - b + 1.0) * ca;
        } else {
            w0 = -p1 / p0;
            w1 = -p2 / p0;
            const double ca = 1.0 / (-a * w1 * w1 + 2 * a * m12 * w1 - a + 1);
            cb = 2 * (a * m12 * w0 - m01 - a * w0 * w1) * ca;
            cc = (1 - a * w0 * w0) * ca;
        }
        double tau0, tau1;
        if (monic_quadratic_roots(cb, cc, tau0, tau1)) {
            for (int r = 0; r < 2; ++r) {
                const double tau = (r == 0) ? tau0 : tau1;
                if (tau <= 0)
                    continue;
                double d0, d1, d2;
                if (elim_d0) {
                    d2 = sqrt(a12 / (tau * (tau - 2.0 * m12) + 1.0));
                    d1 = tau * d2;
                    d0 = (w0 * d2 + w1 * d1);
                    if (d0 < 0)
                        continue;
                } else {
                    d0 = sqrt(a01 / (tau * (tau - 2.0 * m01) + 1.0));
                    d1 = tau * d0;
                    d2 = w0 * d0 + w1 * d1;
                    if (d2 < 0)
                        continue;
                }
                PL_UNROLL
                for (int k = 0; k < 4; ++k) // (cand[n] with a run-time n would put the array into scratch memory on the device)
                    if (k == n)
                        cand[k][0] = d0, cand[k][1] = d1, cand[k][2] = d2;
                ++n;
            }
        }
        if (n > 0 && single_root)
            break;
    }
    return n;
}

PL_HD void p3p_back(const P3PFront &f, double d0, double d1, double d2, Mat3 &R, Vec3 &t) {
    polish_depths(d0, d1, d2, f.a01, f.a02, f.a12, f.m01, f.m02, f.m12);
    const Vec3 v1 = d0 * f.x0 - d1 * f.x1;
    const Vec3 v2 = d0 * f.x0 - d2 * f.x2;
    Mat3 YY;
    set_col(YY, 0, v1);
    set_col(YY, 1, v2);
    set_col(YY, 2, cross(v1, v2));
    R = mul(YY, f.XX);
    t = d0 * f.x0 - mul(R, f.X0);
}

// x: unit bearings, X: 3-D points.  Returns the number of solutions (<= 4), each handed to emit(m, R, t) in the reference's order.
template <typename Emit> PL_HD int p3p_emit(Vec3 x0, Vec3 x1, Vec3 x2, Vec3 X0, Vec3 X1, Vec3 X2, Emit &&emit) {
    P3PFront f;
    double cand[4][3];
    const int n = p3p_front(x0, x1, x2, X0, X1, X2, f, cand);
    PL_UNROLL
    for (int m = 0; m < 4; ++m)
        if (m < n) {
            Mat3 R;
            Vec3 t;
            p3p_back(f, cand[m][0], cand[m][1], cand[m][2], R, t);
            emit(m, R, t);
        }
    return n;
}
PL_HD int p3p(Vec3 x0, Vec3 x1, Vec3 x2, Vec3 X0, Vec3 X1, Vec3 X2, P3PSolution *out) {
    return p3p_emit(x0, x1, x2, X0, X1, X2, [&](int m, const Mat3 &R, const Vec3 &t) {
        out[m].R = R;
        out[m].t = t;
    });
}

} // namespace pl
