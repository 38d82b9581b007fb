// Host-side 3x3 SVD of the fundamental-matrix refinement's entry (driver.cc lm_params_from_record); a header of
// its own so that tests/hostmath can compile it on the CPU and compare it bit for bit with the oracle's.
#pragma once
#include "pl_math.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <utility>

namespace pl {

// SVD of a 3x3 (row-major), A = U diag(s) V^T, s descending - used once per fundamental-matrix refinement to
// enter the Bartoli-Sturm factorisation (PoseLib/robust/optim/optim_utils.h:57-72: Eigen::JacobiSVD, then U resp.
// V negated when its determinant is negative).  A fundamental matrix has rank 2, so det(U)·det(V) is the sign of a
// rounding-level third singular value BEFORE it is made positive: the sign of the refined F that the caller gets
// back is a function of the SVD's exact operation order.  This therefore follows Eigen 3.4's JacobiSVD for a real
// square matrix step by step (two-sided Jacobi: sweeps over (p, q) = (1,0), (2,0), (2,1); the 2x2 block is made
// symmetric by a left rotation, then diagonalised; negative diagonal entries negate the column of U; selection
// sort swaps columns of U and V together), plain IEEE operations, no contraction (the TU is built with
// -ffp-contract=off).
struct PlaneRot {
    double c, s;
    bool is_identity() const { return c == 1.0 && s == 0.0; }
};
// (x, y) <- (c x + s y, -s x + c y) over two strided triples of a row-major 3x3
inline void rotate_pair(double *x, double *y, int stride, PlaneRot r) {
    if (r.is_identity())
        return;
    for (int i = 0; i < 3; ++i, x += stride, y += stride) {
        const double xi = *x, yi = *y;
        *x = r.c * xi + r.s * yi;
        *y = -r.s * xi + r.c * yi;
    }
}
inline PlaneRot symmetric_jacobi(double x, double y, double z) { // diagonalises [x y; y z]
    const double deno = 2.0 * std::fabs(y);
    if (deno < DBL_MIN)
        return PlaneRot{1.0, 0.0};
    const double tau = (x - z) / deno;
    const double w = std::sqrt(tau * tau + 1.0);
    const double t = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0.0 ? 1.0 : -1.0;
    const double n = 1.0 / std::sqrt(t * t + 1.0);
    return PlaneRot{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}
inline void svd3(const Mat3 &A, Mat3 &U, double s[3], Mat3 &V) {
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) {
        const double a = std::fabs(A.m[i]);
        if (!(a <= scale))
            scale = a;
        U.m[i] = V.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    if (!std::isfinite(scale)) {
        s[0] = s[1] = s[2] = std::numeric_limits<double>::quiet_NaN();
        return;
    }
    if (scale == 0.0)
        scale = 1.0;
    Mat3 W;
    for (int i = 0; i < 9; ++i)
        W.m[i] = A.m[i] / scale;
    double max_diag = std::max(std::fabs(W(0, 0)), std::max(std::fabs(W(1, 1)), std::fabs(W(2, 2))));
    for (bool finished = false; !finished;) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double threshold = std::max(DBL_MIN, 2.0 * DBL_EPSILON * max_diag);
                if (!(std::fabs(W(p, q)) > threshold || std::fabs(W(q, p)) > threshold))
                    continue;
                finished = false;
                // the 2x2 block [W_pp W_pq; W_qp W_qq]: symmetrise from the left, then diagonalise
                double blk[4] = {W(p, p), W(p, q), W(q, p), W(q, q)};
                PlaneRot sym{1.0, 0.0};
                const double tr = blk[0] + blk[3], df = blk[2] - blk[1];
                if (!(std::fabs(df) < DBL_MIN)) {
                    const double u = tr / df, h = std::sqrt(1.0 + u * u);
                    sym.s = 1.0 / h;
                    sym.c = u / h;
                }
                if (!sym.is_identity()) {
                    const double x0 = blk[0], y0 = blk[2], x1 = blk[1], y1 = blk[3];
                    blk[0] = sym.c * x0 + sym.s * y0;
                    blk[1] = sym.c * x1 + sym.s * y1;
                    blk[3] = -sym.s * x1 + sym.c * y1;
                }
                const PlaneRot right = symmetric_jacobi(blk[0], blk[1], blk[3]);
                const PlaneRot right_t{right.c, -right.s};
                const PlaneRot left{sym.c * right_t.c - sym.s * right_t.s, sym.c * right_t.s + sym.s * right_t.c};
                rotate_pair(&W.m[3 * p], &W.m[3 * q], 1, left);    // rows p, q of W
                rotate_pair(&U.m[p], &U.m[q], 3, left);            // columns p, q of U
                rotate_pair(&W.m[p], &W.m[q], 3, right_t);         // columns p, q of W
                rotate_pair(&V.m[p], &V.m[q], 3, right_t);         // columns p, q of V
                max_diag = std::max(max_diag, std::max(std::fabs(W(p, p)), std::fabs(W(q, q))));
            }
    }
    for (int i = 0; i < 3; ++i) {
        const double a = W(i, i);
        s[i] = std::fabs(a);
        if (a < 0.0)
            for (int r = 0; r < 3; ++r)
                U(r, i) = -U(r, i);
    }
    for (int i = 0; i < 3; ++i)
        s[i] *= scale;
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        for (int k = i + 1; k < 3; ++k)
            if (s[k] > s[pos])
                pos = k;
        if (s[pos] == 0.0)
            break;
        if (pos != i) {
            std::swap(s[i], s[pos]);
            for (int r = 0; r < 3; ++r) {
                std::swap(U(r, i), U(r, pos));
                std::swap(V(r, i), V(r, pos));
            }
        }
    }
}

} // namespace pl
