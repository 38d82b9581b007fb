// ORACLE — TEST INFRASTRUCTURE ONLY (stand-in for a header-only dependency that is absent from this image).
// Two-sided Jacobi SVD of a real 3x3 matrix in the operation order of Eigen 3.4's JacobiSVD::compute for a
// square real matrix (the published algorithm: Eigen/src/SVD/JacobiSVD.h "step 2 ... step 4",
// internal::real_2x2_jacobi_svd of Eigen/src/misc/RealSvd2x2.h, JacobiRotation::makeJacobi / operator* /
// apply_rotation_in_the_plane of Eigen/src/Jacobi/Jacobi.h), restated from those descriptions:
//
//   1. scale = max |a_ij| (1 if 0), W = A / scale, U = V = I
//   2. sweeps over the index pairs (p, q), p = 1 .. n-1, q = 0 .. p-1; a pair is processed when
//      |W_pq| or |W_qp| exceeds max(DBL_MIN, 2 eps * maxDiag); the 2x2 block is made symmetric by a
//      left rotation rot1 (t = W_pp + W_qq, d = W_qp - W_pq, u = t / d), diagonalised by the Jacobi
//      rotation j_right of the symmetrised block, j_left = rot1 * j_right^T;
//      W <- j_left W, U <- U j_left^T, W <- W j_right, V <- V j_right; maxDiag follows the diagonal
//   3. negative diagonal entries: singular value = |.|, the COLUMN OF U is negated
//   4. selection sort, descending, swapping the columns of U and V together
//
// The SIGN STRUCTURE is what the hot path depends on: U and V leave step 2 as products of plane rotations
// (det +1), step 3 flips det(U) once per negative diagonal entry, step 4 flips both determinants together.
// FactorizedFundamentalMatrix (PoseLib/robust/optim/optim_utils.h:57-72) negates U resp. V when its
// determinant is negative, so the sign of every LO-refined F is the sign of the product of W's diagonal after
// step 2 - for a rank-2 F the sign of a rounding-level third entry, i.e. a function of this exact operation
// order.  One routine therefore serves the shim's Eigen::JacobiSVD (reference sources), the oracle
// (oracle/src/refine.cc svd3) and is restated once more in the product (poselib_amd/csrc/driver.cc svd3,
// which may not include anything under oracle/).
//
// Arithmetic: plain IEEE operations in the order written, no fused multiply-add (the reference's Release
// build has no -march; its packet path computes c*x + s*y / c*y - s*x element-wise with separate
// multiplications and one addition, which rounds like the scalar form below).
#ifndef EIGEN_SHIM_JACOBI_SVD_3X3_H
#define EIGEN_SHIM_JACOBI_SVD_3X3_H
#include <cmath>
#include <limits>
#include <utility>

namespace eigen_shim_detail {

template <typename T> struct PlaneRot { // JacobiRotation: [c s; -s c] acting as x' = c x + s y, y' = -s x + c y
    T c, s;
};

// rows/cols p, q of a 3x3 array through the rotation j (Jacobi.h apply_rotation_in_the_plane)
template <typename T> inline void rot_rows(T (&M)[3][3], int p, int q, const PlaneRot<T> &j) {
    if (j.c == T(1) && j.s == T(0))
        return;
    for (int i = 0; i < 3; ++i) {
        const T xi = M[p][i], yi = M[q][i];
        M[p][i] = j.c * xi + j.s * yi;
        M[q][i] = -j.s * xi + j.c * yi;
    }
}
template <typename T> inline void rot_cols(T (&M)[3][3], int p, int q, const PlaneRot<T> &j) {
    if (j.c == T(1) && j.s == T(0))
        return;
    for (int i = 0; i < 3; ++i) {
        const T xi = M[i][p], yi = M[i][q];
        M[i][p] = j.c * xi + j.s * yi;
        M[i][q] = -j.s * xi + j.c * yi;
    }
}

// JacobiRotation::makeJacobi(x, y, z) for the symmetric block [x y; y z]
template <typename T> inline PlaneRot<T> make_jacobi(T x, T y, T z) {
    PlaneRot<T> r;
    const T deno = T(2) * std::abs(y);
    if (deno < (std::numeric_limits<T>::min)()) {
        r.c = T(1), r.s = T(0);
        return r;
    }
    const T tau = (x - z) / deno;
    const T w = std::sqrt(tau * tau + T(1));
    const T t = (tau > T(0)) ? T(1) / (tau + w) : T(1) / (tau - w);
    const T sign_t = t > T(0) ? T(1) : T(-1);
    const T n = T(1) / std::sqrt(t * t + T(1));
    r.s = -sign_t * (y / std::abs(y)) * std::abs(t) * n;
    r.c = n;
    return r;
}

// A = U diag(S) V^T, S descending and non-negative.  Arrays are [row][col].
template <typename T> inline void jacobi_svd3(const T (&A)[3][3], T (&U)[3][3], T (&S)[3], T (&V)[3][3]) {
    const T precision = T(2) * std::numeric_limits<T>::epsilon();
    const T consider_as_zero = (std::numeric_limits<T>::min)();
    T scale = T(0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const T a = std::abs(A[i][j]);
            if (!(a <= scale)) // NaN propagates
                scale = a;
        }
    T W[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            U[i][j] = V[i][j] = (i == j) ? T(1) : T(0);
            W[i][j] = A[i][j];
        }
    if (!std::isfinite(scale)) { // Eigen: InvalidInput, factors unspecified - identity factors, NaN values
        S[0] = S[1] = S[2] = std::numeric_limits<T>::quiet_NaN();
        return;
    }
    if (scale == T(0))
        scale = T(1);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            W[i][j] = A[i][j] / scale;

    T max_diag = std::abs(W[0][0]);
    for (int i = 1; i < 3; ++i)
        if (std::abs(W[i][i]) > max_diag)
            max_diag = std::abs(W[i][i]);

    bool finished = false;
    while (!finished) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const T thr_a = precision * max_diag;
                const T threshold = consider_as_zero > thr_a ? consider_as_zero : thr_a;
                if (std::abs(W[p][q]) > threshold || std::abs(W[q][p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd on [W_pp W_pq; W_qp W_qq]
                    T m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    PlaneRot<T> rot1;
                    const T t = m00 + m11;
                    const T d = m10 - m01;
                    if (std::abs(d) < (std::numeric_limits<T>::min)()) {
                        rot1.s = T(0), rot1.c = T(1);
                    } else {
                        const T u = t / d;
                        const T tmp = std::sqrt(T(1) + u * u);
                        rot1.s = T(1) / tmp;
                        rot1.c = u / tmp;
                    }
                    if (!(rot1.c == T(1) && rot1.s == T(0))) { // m.applyOnTheLeft(0, 1, rot1)
                        const T x0 = m00, y0 = m10, x1 = m01, y1 = m11;
                        m00 = rot1.c * x0 + rot1.s * y0;
                        m10 = -rot1.s * x0 + rot1.c * y0;
                        m01 = rot1.c * x1 + rot1.s * y1;
                        m11 = -rot1.s * x1 + rot1.c * y1;
                    }
                    (void)m10;
                    const PlaneRot<T> jr = make_jacobi(m00, m01, m11);
                    // j_left = rot1 * jr^T, jr^T = (c, -s)
                    PlaneRot<T> jl;
                    const T os = -jr.s;
                    jl.c = rot1.c * jr.c - rot1.s * os;
                    jl.s = rot1.c * os + rot1.s * jr.c;

                    rot_rows(W, p, q, jl); // W.applyOnTheLeft(p, q, j_left)
                    rot_cols(U, p, q, jl); // U.applyOnTheRight(p, q, j_left^T) = rotation by (j_left^T)^T
                    const PlaneRot<T> jrt = {jr.c, -jr.s};
                    rot_cols(W, p, q, jrt); // W.applyOnTheRight(p, q, j_right) = rotation by j_right^T
                    rot_cols(V, p, q, jrt);

                    const T a = std::abs(W[p][p]), b = std::abs(W[q][q]);
                    const T ab = a > b ? a : b; // numext::maxi(a, b) = a < b ? b : a
                    if (max_diag < ab)
                        max_diag = ab;
                }
            }
    }

    for (int i = 0; i < 3; ++i) {
        const T a = W[i][i];
        S[i] = std::abs(a);
        if (a < T(0))
            for (int r = 0; r < 3; ++r)
                U[r][i] = -U[r][i];
    }
    for (int i = 0; i < 3; ++i)
        S[i] *= scale;

    for (int i = 0; i < 3; ++i) { // selection sort; the first maximum wins (maxCoeff)
        int pos = i;
        for (int k = i + 1; k < 3; ++k)
            if (S[k] > S[pos])
                pos = k;
        if (S[pos] == T(0))
            break;
        if (pos != i) {
            std::swap(S[i], S[pos]);
            for (int r = 0; r < 3; ++r) {
                std::swap(U[r][i], U[r][pos]);
                std::swap(V[r][i], V[r][pos]);
            }
        }
    }
}

} // namespace eigen_shim_detail
#endif
