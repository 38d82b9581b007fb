// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the univariate root finders and the P3P minimal solver.
//   closed-form roots : PoseLib/misc/univariate.cc:48-61 (real quadratic), :74-92 (one real cubic root),
//                       :94-126 (all real cubic roots + one Newton polish)
//   P3P               : PoseLib/solvers/p3p.cc:39-202 with helpers PoseLib/solvers/p3p_common.h:7-94
// libm calls (cbrt / acos / cos / sqrt) are the same ones the reference makes.
#include "solvers.h"

#include <cmath>
#include <utility>

namespace orc {

int quadratic_real_roots(double a, double b, double c, double r[2]) { // univariate.cc:48-61
    const double disc = b * b - 4 * a * c;
    if (disc < 0)
        return 0;
    const double s = std::sqrt(disc);
    r[0] = (b > 0) ? (2 * c) / (-b - s) : (2 * c) / (-b + s);
    r[1] = c / (a * r[0]);
    return 2;
}

bool cubic_one_real_root(double c2, double c1, double c0, double &root) { // univariate.cc:74-92
    const double a = c1 - c2 * c2 / 3.0;
    double b = (2.0 * c2 * c2 * c2 - 9.0 * c2 * c1) / 27.0 + c0;
    double c = b * b / 4.0 + a * a * a / 27.0;
    if (c != 0) {
        if (c > 0) {
            c = std::sqrt(c);
            b *= -0.5;
            root = std::cbrt(b + c) + std::cbrt(b - c) - c2 / 3.0;
            return true; // exactly one real root
        }
        c = 3.0 * b / (2.0 * a) * std::sqrt(-3.0 / a);
        root = 2.0 * std::sqrt(-a / 3.0) * std::cos(std::acos(c) / 3.0) - c2 / 3.0;
        return false;
    }
    root = -c2 / 3.0 + (a != 0 ? (3.0 * b / a) : 0);
    return false;
}

int cubic_real_roots(double c2, double c1, double c0, double r[3]) { // univariate.cc:94-126
    const double a = c1 - c2 * c2 / 3.0;
    double b = (2.0 * c2 * c2 * c2 - 9.0 * c2 * c1) / 27.0 + c0;
    double c = b * b / 4.0 + a * a * a / 27.0;
    int n;
    if (a == 0.0 && b == 0.0) {
        r[0] = r[1] = r[2] = -c2 / 3.0;
        n = 3;
    } else if (c > 0) {
        c = std::sqrt(c);
        b *= -0.5;
        r[0] = std::cbrt(b + c) + std::cbrt(b - c) - c2 / 3.0;
        n = 1;
    } else {
        c = 3.0 * b / (2.0 * a) * std::sqrt(-3.0 / a);
        const double d = 2.0 * std::sqrt(-a / 3.0);
        r[0] = d * std::cos(std::acos(c) / 3.0) - c2 / 3.0;
        r[1] = d * std::cos(std::acos(c) / 3.0 - 2.09439510239319526263557236234192) - c2 / 3.0;
        r[2] = d * std::cos(std::acos(c) / 3.0 - 4.18879020478639052527114472468384) - c2 / 3.0;
        n = 3;
    }
    for (int i = 0; i < n; ++i) { // one Newton step on the monic cubic
        const double x = r[i];
        const double x2 = x * x;
        const double x3 = x * x2;
        const double dx = -(x3 + c2 * x2 + c1 * x + c0) / (3 * x2 + 2 * c2 * x + c1);
        r[i] += dx;
    }
    return n;
}

namespace {

// p3p_common.h:7-29
bool monic_quadratic_roots(double b, double c, double &r1, double &r2) {
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
    const double y = std::sqrt(v);
    if (b < 0) {
        r1 = 0.5 * (-b + y);
        r2 = 0.5 * (-b - y);
    } else {
        r1 = 2.0 * c / (-b + y);
        r2 = 2.0 * c / (-b - y);
    }
    return true;
}

// Split the rank-2 conic C into its two lines; returns (column 0, row 0) of the
// de-symmetrised matrix.  p3p_common.h:31-71
void split_degenerate_conic(M3 C, V3 &p, V3 &q) {
    auto c = [&](int i, int j) -> double & { return C.m[i][j]; };
    M3 A; // negated adjugate (symmetric)
    A.m[0][0] = c(1, 2) * c(2, 1) - c(1, 1) * c(2, 2);
    A.m[1][1] = c(0, 2) * c(2, 0) - c(0, 0) * c(2, 2);
    A.m[2][2] = c(0, 1) * c(1, 0) - c(0, 0) * c(1, 1);
    A.m[0][1] = c(0, 1) * c(2, 2) - c(0, 2) * c(2, 1);
    A.m[0][2] = c(0, 2) * c(1, 1) - c(0, 1) * c(1, 2);
    A.m[1][0] = A.m[0][1];
    A.m[1][2] = c(0, 0) * c(1, 2) - c(0, 2) * c(1, 0);
    A.m[2][0] = A.m[0][2];
    A.m[2][1] = A.m[1][2];

    int pick;
    if (A.m[0][0] > A.m[1][1])
        pick = (A.m[0][0] > A.m[2][2]) ? 0 : 2;
    else
        pick = (A.m[1][1] > A.m[2][2]) ? 1 : 2;
    const V3 v = A.col(pick) / std::sqrt(A.m[pick][pick]);

    c(0, 1) -= v.z;
    c(0, 2) += v.y;
    c(1, 2) -= v.x;
    c(1, 0) += v.z;
    c(2, 0) -= v.y;
    c(2, 1) += v.x;
    p = C.col(0);
    q = C.row(0);
}

// Newton polish of the three depths.  p3p_common.h:74-94
void polish_depths(double &l1, double &l2, double &l3, double a12, double a13, double a23, double b12, double b13,
                   double b23) {
    for (int it = 0; it < 5; ++it) {
        const double r1 = (l1 * l1 - 2.0 * l1 * l2 * b12 + l2 * l2 - a12);
        const double r2 = (l1 * l1 - 2.0 * l1 * l3 * b13 + l3 * l3 - a13);
        const double r3 = (l2 * l2 - 2.0 * l2 * l3 * b23 + l3 * l3 - a23);
        if (std::abs(r1) + std::abs(r2) + std::abs(r3) < 1e-10)
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

} // namespace

int p3p(const V3 xin[3], const V3 Xin[3], Pose out[4]) { // p3p.cc:39-202
    V3 X[3] = {Xin[0], Xin[1], Xin[2]};
    V3 x[3] = {xin[0], xin[1], xin[2]};
    V3 X01 = X[0] - X[1], X02 = X[0] - X[2], X12 = X[1] - X[2];
    double a01 = sqnorm(X01), a02 = sqnorm(X02), a12 = sqnorm(X12);

    // relabel so that |X1-X2| is the longest side (p3p.cc:58-73)
    if (a01 > a02) {
        if (a01 > a12) {
            std::swap(x[0], x[2]);
            std::swap(X[0], X[2]);
            std::swap(a01, a12);
            X01 = -X12;
            X02 = -X02;
        }
    } else if (a02 > a12) {
        std::swap(x[0], x[1]);
        std::swap(X[0], X[1]);
        std::swap(a02, a12);
        X01 = -X01;
        X02 = X12;
    }

    const double a12d = 1.0 / a12;
    const double a = a01 * a12d, b = a02 * a12d;
    const double m01 = dot(x[0], x[1]), m02 = dot(x[0], x[2]), m12 = dot(x[1], x[2]);

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

    M3 C;
    C.m[0][0] = -a + s * (1 - b);
    C.m[0][1] = -m02 * s;
    C.m[0][2] = a * m12 + b * m12 * s;
    C.m[1][0] = C.m[0][1];
    C.m[1][1] = s + 1;
    C.m[1][2] = -m01;
    C.m[2][0] = C.m[0][2];
    C.m[2][1] = C.m[1][2];
    C.m[2][2] = -a - b * s + 1;

    V3 lines[2];
    split_degenerate_conic(C, lines[0], lines[1]);

    M3 XX;
    XX.set_col(0, X01);
    XX.set_col(1, X02);
    XX.set_col(2, cross(X01, X02));
    XX = inverse(XX);

    int n = 0;
    auto emit = [&](double d0, double d1, double d2) {
        polish_depths(d0, d1, d2, a01, a02, a12, m01, m02, m12);
        const V3 v1 = d0 * x[0] - d1 * x[1];
        const V3 v2 = d0 * x[0] - d2 * x[2];
        M3 YY;
        YY.set_col(0, v1);
        YY.set_col(1, v2);
        YY.set_col(2, cross(v1, v2));
        const M3 R = YY * XX;
        out[n++] = Pose(R, d0 * x[0] - R * X[0]);
    };

    for (int i = 0; i < 2; ++i) {
        const double p0 = lines[i].x, p1 = lines[i].y, p2 = lines[i].z;
        if (std::abs(p0) <= std::abs(p1)) { // eliminate d0
            const double w0 = -p0 / p1;
            const double w1 = -p2 / p1;
            const double ca = 1.0 / (w1 * w1 - b);
            const double cb = 2.0 * (b * m12 - m02 * w1 + w0 * w1) * ca;
            const double cc = (w0 * w0 - 2 * m02 * w0 - b + 1.0) * ca;
            double tau[2];
            if (!monic_quadratic_roots(cb, cc, tau[0], tau[1]))
                continue;
            for (double t : tau) {
                if (t <= 0)
                    continue;
                const double d2 = std::sqrt(a12 / (t * (t - 2.0 * m12) + 1.0));
                const double d1 = t * d2;
                const double d0 = (w0 * d2 + w1 * d1);
                if (d0 < 0)
                    continue;
                emit(d0, d1, d2);
            }
        } else {
            const double w0 = -p1 / p0;
            const double w1 = -p2 / p0;
            const double ca = 1.0 / (-a * w1 * w1 + 2 * a * m12 * w1 - a + 1);
            const double cb = 2 * (a * m12 * w0 - m01 - a * w0 * w1) * ca;
            const double cc = (1 - a * w0 * w0) * ca;
            double tau[2];
            if (!monic_quadratic_roots(cb, cc, tau[0], tau[1]))
                continue;
            for (double t : tau) {
                if (t <= 0)
                    continue;
                const double d0 = std::sqrt(a01 / (t * (t - 2.0 * m01) + 1.0));
                const double d1 = t * d0;
                const double d2 = w0 * d0 + w1 * d1;
                if (d2 < 0)
                    continue;
                emit(d0, d1, d2);
            }
        }
        if (n > 0 && single_root)
            break;
    }
    return n;
}

} // namespace orc
