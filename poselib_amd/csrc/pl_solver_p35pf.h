// poselib_amd - P3.5Pf: absolute pose and focal length from three 2D-3D correspondences and the x coordinate of a fourth.
//
// Interface of the reference's solver (PoseLib/solvers/p35pf.h:39-54: image points relative to the principal point; the image
// points are scaled by their mean norm, p35pf.cc:45-58; solutions with det > 0, |third row| = 1, focal = mean norm of the first
// two rows, p35pf.cc:903-921).  The ALGORITHM is not the reference's generated elimination template but this project's own,
// derived from first principles (FocalAbsolutePoseEstimator): P = sum alpha_k N_k over the 5-dimensional null space
// of the seven linear constraints, alpha_5 = 1; with a1, a2, a3 the rows of the left 3 x 3 block of P
//     a1.a2 = a1.a3 = a2.a3 = 0, |a1|^2 = |a2|^2                              (4 quadrics, each also times x1..x4)
//     (a2 x a3)_i (a2)_j = (a3 x a1)_j (a1)_i,  i, j = 1..3                    (9 cubics: they remove the six f = 0 roots)
// are 29 equations, linear in the 35 monomials of degree <= 3 in (x1..x4); one Gauss-Jordan elimination of 25 monomials leaves
// the multiplication by x4 on the standard monomials {x3^2, x1 x4, x2 x4, x3 x4, x4^2, x1, x2, x3, x4, 1} as a 10 x 10 matrix,
// whose real eigenvalues (Hessenberg + Francis QR) and null vectors give the solutions, ascending in x4.
// Operation for operation the same as the oracle's statement of this algorithm (oracle/src/solvers_focal.cc - written
// independently of this file's storage layout); tests/test_hostmath_vs_oracle.py compares the two bit for bit on the host.
//
// Storage: the 29 x 35 elimination matrix of a sample (8 KB) lives in a workspace the caller provides - element (r, c) at
// work[(r * 35 + c) * stride]: on the device one lane = one sample and stride = samples of the launch, so that the 64 lanes of a
// wavefront touch consecutive doubles.
#pragma once
#include "pl_math.h"
#if defined(PL_EIG_SHADOW_CHECK) && !defined(__HIPCC__)
#include "pl_eigen_packed.h" // (tests/hostmath: the packed eigenvalue / null-vector routines shadow the serial ones, below)
#include "pl_nullvec_packed.h"
#endif

namespace pl {

// monomials of degree <= 3 in x1..x4, graded: 0..19 cubic, 20..29 quadratic, 30..33 = x1..x4, 34 = 1
// kP35Prod[i][j]: index of monomial i * monomial j (-1: degree > 3)
static constexpr int8_t kP35Prod[35][35] = {
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 2},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 3},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 4},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 5},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 6},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 7},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 8},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 9},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 10},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 11},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 13},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 14},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 15},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 16},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 17},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 18},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 19},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 1, 2, 3, 20},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1, 4, 5, 6, 21},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 2, 5, 7, 8, 22},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 3, 6, 8, 9, 23},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 4, 10, 11, 12, 24},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 5, 11, 13, 14, 25},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 6, 12, 14, 15, 26},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 7, 13, 16, 17, 27},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 8, 14, 17, 18, 28},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 9, 15, 18, 19, 29},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 20, 21, 22, 23, 30},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1, 4, 5, 6, 10, 11, 12, 13, 14, 15, 21, 24, 25, 26, 31},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 2, 5, 7, 8, 11, 13, 14, 16, 17, 18, 22, 25, 27, 28, 32},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 3, 6, 8, 9, 12, 14, 15, 17, 18, 19, 23, 26, 28, 29, 33},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34},
};
static constexpr uint8_t kP35Basis[10] = {27, 23, 26, 28, 29, 30, 31, 32, 33, 34};
static constexpr uint8_t kP35Elim[25] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 25};
static constexpr int8_t kP35Shifted[10] = {-18, -10, -16, -19, -20, 1, 2, 3, 4, 8};

constexpr int kP35Rows = 29, kP35Cols = 35;
constexpr int kP35WorkDoubles = kP35Rows * kP35Cols; // per sample

// polynomial with coefficients of the monomials F..34 only (F = 30: linear, 20: quadratic, 0: cubic)
template <int F> struct P35Poly {
    double c[35 - F];
};
template <int F> PL_HD void p35_zero(P35Poly<F> &p) {
    PL_UNROLL
    for (int i = 0; i < 35 - F; ++i)
        p.c[i] = 0.0;
}
// r += p * q, term by term in ascending (i, j) - zero coefficients are skipped (as the oracle's dense product does).
// (unrolled: kP35Prod folds to constants and the coefficients stay in registers - a run-time index puts them into scratch memory)
template <int FR, int FP, int FQ> PL_HD void p35_mul(const P35Poly<FP> &p, const P35Poly<FQ> &q, P35Poly<FR> &r) {
    p35_zero(r);
    PL_UNROLL
    for (int i = FP; i < 35; ++i)
        if (p.c[i - FP] != 0) {
            PL_UNROLL
            for (int j = FQ; j < 35; ++j)
                if (q.c[j - FQ] != 0)
                    r.c[kP35Prod[i][j] - FR] += p.c[i - FP] * q.c[j - FQ];
        }
}
typedef P35Poly<30> P35Lin;
typedef P35Poly<20> P35Quad;
typedef P35Poly<0> P35Cubic;

PL_HD void p35_dot(const P35Lin *a, const P35Lin *b, P35Quad &out) { // (a0 b0 + a1 b1) + a2 b2, coefficient by coefficient
    P35Quad m0, m1, m2;
    p35_mul(a[0], b[0], m0);
    p35_mul(a[1], b[1], m1);
    p35_mul(a[2], b[2], m2);
    PL_UNROLL
    for (int i = 0; i < 15; ++i)
        out.c[i] = (m0.c[i] + m1.c[i]) + m2.c[i];
}
PL_HD void p35_cross(const P35Lin *a, const P35Lin *b, P35Quad *out) {
    P35Quad m0, m1;
    PL_UNROLL
    for (int k = 0; k < 3; ++k) {
        const int i = (k + 1) % 3, j = (k + 2) % 3;
        p35_mul(a[i], b[j], m0);
        p35_mul(a[j], b[i], m1);
        PL_UNROLL
        for (int t = 0; t < 15; ++t)
            out[k].c[t] = m0.c[t] - m1.c[t];
    }
}

// element e of a per-sample array that lives in a strided workspace (LDS on the device: element-major over the samples)
struct StridedArr {
    double *base;
    size_t stride;
    PL_HD double &operator[](int e) const { return base[(size_t)e * stride]; }
    PL_HD StridedArr at(int off) const { return StridedArr{base + (size_t)off * stride, stride}; }
};
struct P35Work {
    double *base;
    size_t stride;
    PL_HD double &at(int r, int c) const { return base[(size_t)(r * kP35Cols + c) * stride]; }
    PL_HD StridedArr region(int off) const { return StridedArr{base + (size_t)off * stride, stride}; }
};
// row r of the elimination matrix = eq scaled to unit maximum
PL_HD void p35_store_row(const P35Work &w, int r, const P35Cubic &eq) {
    double mx = 0;
    PL_UNROLL
    for (int c = 0; c < 35; ++c)
        mx = fmax(mx, fabs(eq.c[c]));
    PL_UNROLL
    for (int c = 0; c < 35; ++c)
        w.at(r, c) = mx > 0 ? eq.c[c] / mx : 0.0;
}

// Orthonormal basis of the complement of span(columns of A), A ROWS x COLS column-major: full-pivoting Householder QR, then the
// last ROWS - COLS columns of Q - pl_solver_rel.h complement_basis9_indexed with the row count as a parameter
template <int ROWS, int COLS> PL_HD void complement_basis_indexed(double *qr /* ROWS*COLS, destroyed */, double *basis) {
    double tau[COLS];
    int rowswap[COLS];
    double biggest = 0;
    const double precision = 2.220446049250313e-16 * COLS;
    for (int k = 0; k < COLS; ++k) {
        int pr = k, pc = k;
        double best = fabs(qr[k * ROWS + k]);
        for (int c = k; c < COLS; ++c)
            for (int r = k; r < ROWS; ++r) {
                const double v = fabs(qr[c * ROWS + r]);
                if (v > best) {
                    best = v;
                    pr = r;
                    pc = c;
                }
            }
        if (k == 0)
            biggest = best;
        if (best <= biggest * precision) {
            for (int i = k; i < COLS; ++i) {
                rowswap[i] = i;
                tau[i] = 0;
            }
            break;
        }
        rowswap[k] = pr;
        if (pr != k)
            for (int c = k; c < COLS; ++c) {
                const double t = qr[c * ROWS + k];
                qr[c * ROWS + k] = qr[c * ROWS + pr];
                qr[c * ROWS + pr] = t;
            }
        if (pc != k)
            for (int r = 0; r < ROWS; ++r) {
                const double t = qr[k * ROWS + r];
                qr[k * ROWS + r] = qr[pc * ROWS + r];
                qr[pc * ROWS + r] = t;
            }
        double tail_sq = 0;
        for (int r = k + 1; r < ROWS; ++r)
            tail_sq += qr[k * ROWS + r] * qr[k * ROWS + r];
        const double c0 = qr[k * ROWS + k];
        double beta;
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0;
            beta = c0;
            for (int r = k + 1; r < ROWS; ++r)
                qr[k * ROWS + r] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0)
                beta = -beta;
            for (int r = k + 1; r < ROWS; ++r)
                qr[k * ROWS + r] = qr[k * ROWS + r] / (c0 - beta);
            tau[k] = (beta - c0) / beta;
        }
        qr[k * ROWS + k] = beta;
        if (tau[k] != 0)
            for (int c = k + 1; c < COLS; ++c) {
                double t = 0;
                for (int r = k + 1; r < ROWS; ++r)
                    t += qr[k * ROWS + r] * qr[c * ROWS + r];
                t += qr[c * ROWS + k];
                qr[c * ROWS + k] -= tau[k] * t;
                for (int r = k + 1; r < ROWS; ++r)
                    qr[c * ROWS + r] -= tau[k] * qr[k * ROWS + r] * t;
            }
    }
    for (int j = 0; j < ROWS - COLS; ++j) {
        double v[ROWS];
        for (int r = 0; r < ROWS; ++r)
            v[r] = (r == COLS + j) ? 1.0 : 0.0;
        for (int k = COLS - 1; k >= 0; --k) {
            if (tau[k] != 0) {
                double t = 0;
                for (int r = k + 1; r < ROWS; ++r)
                    t += qr[k * ROWS + r] * v[r];
                t += v[k];
                v[k] -= tau[k] * t;
                for (int r = k + 1; r < ROWS; ++r)
                    v[r] -= tau[k] * qr[k * ROWS + r] * t;
            }
            if (rowswap[k] != k) {
                const double t = v[k];
                v[k] = v[rowswap[k]];
                v[rowswap[k]] = t;
            }
        }
        for (int r = 0; r < ROWS; ++r)
            basis[j * ROWS + r] = v[r];
    }
}

// Real eigenvalues of an n x n matrix (row-major, destroyed; Arr: a pointer or anything indexable that yields double&), ascending: Householder reduction to Hessenberg form, then the
// Francis double-shift QR iteration in its textbook form; an eigenvalue counts as real when |imag| <= tol (1 + |real|).
// No convergence after 60 sweeps on one block: no eigenvalues (the sample is dropped).
#ifndef PL_EIG_MARK
#define PL_EIG_MARK() // (scripts/exp/p35_phases.cc: cycle counter between the Hessenberg reduction and the QR iteration)
#endif
#if defined(PL_EIG_SHADOW_CHECK) && !defined(__HIPCC__)
// tests/hostmath: every matrix the solvers hand to this routine also goes through the packed form of pl_eigen_packed.h (the device's
// routine since round 5, here with the lane loops as loops); calls and disagreements (count or any bit of an eigenvalue) are counted
extern unsigned long long pl_eig_shadow_counters[6]; // [eigenvalue calls, disagreements, balancing calls, disagreements, null-vector calls, disagreements]
template <int n, class Arr> inline int pl_real_eigenvalues_serial(Arr a_, double *out, double tol);
template <int n, class Arr> inline int pl_real_eigenvalues(Arr a_, double *out, double tol) {
    double shadow[n * n + 4 * n];
    for (int e = 0; e < n * n; ++e)
        shadow[e] = a_[e];
    for (int e = n * n; e < n * n + 4 * n; ++e)
        shadow[e] = 0.0;
    const int m = pl_real_eigenvalues_serial<n, Arr>(a_, out, tol);
    EigFlatHost<n> cx{shadow};
    const int m2 = pl_real_eigenvalues_packed<n>(cx, true, tol);
    bool same = m == m2;
    for (int i = 0; same && i < m; ++i)
        same = std::memcmp(&out[i], &cx.out(i), sizeof(double)) == 0;
    pl_eig_shadow_counters[0]++;
    pl_eig_shadow_counters[1] += same ? 0 : 1;
    return m;
}
#define pl_real_eigenvalues_impl pl_real_eigenvalues_serial
#else
#define pl_real_eigenvalues_impl pl_real_eigenvalues
#endif
template <int n, class Arr> PL_HD int pl_real_eigenvalues_impl(Arr a_, double *out, double tol) {
#define PL_A(i, j) a_[(i) * n + (j)]
    for (int k = 0; k + 2 < n; ++k) {
        double tail = 0;
        for (int r = k + 2; r < n; ++r)
            tail += PL_A(r, k) * PL_A(r, k);
        if (tail <= 1e-300)
            continue;
        const double c0 = PL_A(k + 1, k);
        double beta = sqrt(c0 * c0 + tail);
        if (c0 >= 0)
            beta = -beta;
        double v[n];
        for (int r = 0; r < n; ++r)
            v[r] = 0.0;
        v[k + 1] = 1.0;
        for (int r = k + 2; r < n; ++r)
            v[r] = PL_A(r, k) / (c0 - beta);
        const double tau = (beta - c0) / beta;
        for (int c = 0; c < n; ++c) {
            double t = 0;
            for (int r = k + 1; r < n; ++r)
                t += v[r] * PL_A(r, c);
            for (int r = k + 1; r < n; ++r)
                PL_A(r, c) -= tau * v[r] * t;
        }
        for (int r = 0; r < n; ++r) {
            double t = 0;
            for (int c = k + 1; c < n; ++c)
                t += PL_A(r, c) * v[c];
            for (int c = k + 1; c < n; ++c)
                PL_A(r, c) -= tau * t * v[c];
        }
        PL_A(k + 1, k) = beta;
        for (int r = k + 2; r < n; ++r)
            PL_A(r, k) = 0;
    }
    PL_EIG_MARK();
    double wr[n], wi[n];
    for (int i = 0; i < n; ++i)
        wr[i] = wi[i] = 0.0;
    const double eps = 2.220446049250313e-16;
    double anorm = 0;
    for (int i = 0; i < n; ++i)
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < n; ++j)
            anorm += fabs(PL_A(i, j));
    int nn = n - 1;
    double t = 0, p = 0, q = 0, r = 0, s = 0, w = 0, x = 0, y = 0, z = 0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                s = fabs(PL_A(l - 1, l - 1)) + fabs(PL_A(l, l));
                if (s == 0)
                    s = anorm;
                if (fabs(PL_A(l, l - 1)) <= eps * s) {
                    PL_A(l, l - 1) = 0;
                    break;
                }
            }
            x = PL_A(nn, nn);
            if (l == nn) {
                wr[nn] = x + t;
                wi[nn--] = 0;
            } else {
                y = PL_A(nn - 1, nn - 1);
                w = PL_A(nn, nn - 1) * PL_A(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = sqrt(fabs(q));
                    x += t;
                    if (q >= 0) {
                        z = p + (p >= 0 ? fabs(z) : -fabs(z));
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0)
                            wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn - 1] = z;
                        wi[nn] = -z;
                    }
                    nn -= 2;
                } else {
                    if (its == 60)
                        return 0;
                    if (its == 10 || its == 20) {
                        t += x;
                        for (int i = 0; i <= nn; ++i)
                            PL_A(i, i) -= x;
                        s = fabs(PL_A(nn, nn - 1)) + fabs(PL_A(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = PL_A(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / PL_A(m + 1, m) + PL_A(m, m + 1);
                        q = PL_A(m + 1, m + 1) - z - r - s;
                        r = PL_A(m + 2, m + 1);
                        s = fabs(p) + fabs(q) + fabs(r);
                        p /= s, q /= s, r /= s;
                        if (m == l)
                            break;
                        const double u = fabs(PL_A(m, m - 1)) * (fabs(q) + fabs(r));
                        const double v = fabs(p) * (fabs(PL_A(m - 1, m - 1)) + fabs(z) + fabs(PL_A(m + 1, m + 1)));
                        if (u <= eps * v)
                            break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        PL_A(i, i - 2) = 0;
                        if (i != m + 2)
                            PL_A(i, i - 3) = 0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = PL_A(k, k - 1);
                            q = PL_A(k + 1, k - 1);
                            r = (k != nn - 1) ? PL_A(k + 2, k - 1) : 0.0;
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0)
                                p /= x, q /= x, r /= x;
                        }
                        const double sq = sqrt(p * p + q * q + r * r);
                        if ((s = (p >= 0 ? sq : -sq)) != 0) {
                            if (k == m) {
                                if (l != m)
                                    PL_A(k, k - 1) = -PL_A(k, k - 1);
                            } else {
                                PL_A(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s, y = q / s, z = r / s;
                            q /= p, r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = PL_A(k, j) + q * PL_A(k + 1, j);
                                if (k != nn - 1) {
                                    p += r * PL_A(k + 2, j);
                                    PL_A(k + 2, j) -= p * z;
                                }
                                PL_A(k + 1, j) -= p * y;
                                PL_A(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * PL_A(i, k) + y * PL_A(i, k + 1);
                                if (k != nn - 1) {
                                    p += z * PL_A(i, k + 2);
                                    PL_A(i, k + 2) -= p * r;
                                }
                                PL_A(i, k + 1) -= p * q;
                                PL_A(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
#undef PL_A
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (fabs(wi[i]) <= tol * (1.0 + fabs(wr[i]))) { // insertion into the ascending list
            int j = m++;
            while (j > 0 && out[j - 1] > wr[i]) {
                out[j] = out[j - 1];
                --j;
            }
            out[j] = wr[i];
        }
    return m;
}

PL_HD int p35_real_eigenvalues(double *a_, double *out, double tol) { return pl_real_eigenvalues<10, double *>(a_, out, tol); }

// null vector of the singular n x n matrix B (row-major, destroyed): Gaussian elimination with complete pivoting, the last
// permuted unknown set to 1
#if defined(PL_EIG_SHADOW_CHECK) && !defined(__HIPCC__)
// tests/hostmath: every matrix also goes through the packed form (pl_nullvec_packed.h: 16 lanes per matrix on the device, here with the
// lane loops as loops); calls and disagreements (any bit of the null vector) are counted in pl_eig_shadow_counters[4], [5]
template <int n, class Arr> inline void pl_null_vector_serial(Arr B, double *v);
template <int n, class Arr> inline void pl_null_vector(Arr B, double *v) {
    NullFlatHost<n> cx;
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < n; ++j)
            cx.b[r][j] = B[r * n + j];
    pl_null_vector_serial<n, Arr>(B, v);
    pl_null_vector_packed<n>(cx, true);
    bool same = true;
    for (int j = 0; j < n; ++j)
        same = same && std::memcmp(&v[j], &cx.yv[j], sizeof(double)) == 0;
    pl_eig_shadow_counters[4]++;
    pl_eig_shadow_counters[5] += same ? 0 : 1;
}
#define pl_null_vector_impl pl_null_vector_serial
#else
#define pl_null_vector_impl pl_null_vector
#endif
template <int n, class Arr> PL_HD void pl_null_vector_impl(Arr B, double *v) {
    int colperm[n];
    for (int i = 0; i < n; ++i)
        colperm[i] = i;
    for (int k = 0; k < n - 1; ++k) {
        int pr = k, pc = k;
        double best = 0;
        for (int i = k; i < n; ++i)
            for (int j = k; j < n; ++j)
                if (fabs(B[i * n + j]) > best)
                    best = fabs(B[i * n + j]), pr = i, pc = j;
        if (best == 0)
            break;
        for (int j = 0; j < n; ++j) {
            const double t = B[k * n + j];
            B[k * n + j] = B[pr * n + j];
            B[pr * n + j] = t;
        }
        for (int i = 0; i < n; ++i) {
            const double t = B[i * n + k];
            B[i * n + k] = B[i * n + pc];
            B[i * n + pc] = t;
        }
        const int tc = colperm[k];
        colperm[k] = colperm[pc];
        colperm[pc] = tc;
        for (int i = k + 1; i < n; ++i) {
            const double f = B[i * n + k] / B[k * n + k];
            for (int j = k; j < n; ++j)
                B[i * n + j] -= f * B[k * n + j];
        }
    }
    double y[n];
    for (int i = 0; i < n; ++i)
        y[i] = 0.0;
    y[n - 1] = 1.0;
    for (int i = n - 2; i >= 0; --i) {
        double s = 0;
        for (int j = i + 1; j < n; ++j)
            s += B[i * n + j] * y[j];
        y[i] = -s / B[i * n + i];
    }
    for (int i = 0; i < n; ++i)
        v[colperm[i]] = y[i];
}
PL_HD void p35_null_vector(double *B, double *v) { pl_null_vector<10, double *>(B, v); }

struct P35Solution {
    Quat q;
    Vec3 t;
    double focal;
};

// x: four image points (x, y) relative to the principal point - of the fourth only x is used -, X: the 3-D points.
// Returns the number of solutions (<= 10), ascending in the eigenvalue.
#ifndef PL_P35_MARK
#define PL_P35_MARK(i) // (scripts/exp/p35_phases.cc: cycle counter at the phase boundaries)
#endif
// The solver in three stages - on the device three kernels (focal.hip: one lane per sample / one WAVEFRONT per sample with the
// matrix in registers / one lane per sample again), on the host and in p35pf() below one after the other.  Measured shares of the
// one-lane-per-sample form, matrices in LDS (scripts/exp/p35_phases.cc): null space 3 %, equations 9 %, elimination 27 %,
// eigenvalues 37 %, null vectors + poses 24 % of 1.5 ms.
//
// Stage 1: null space N (12 x 5, column k at N[12 k ..]) of the seven linear constraints, scale f0 of the image points, and the 29
// equations as the rows of the elimination matrix w.
// sink(r, eq): takes equation r (on the host and in p35pf(): p35_store_row - the row scaled to unit maximum; on the device the raw
// coefficients and the maximum, the 35 divisions of a row are then done by the 35 lanes that hold its columns, focal.hip)
template <class Sink>
PL_HD void p35pf_setup_t(const double *xs /* 4 x 2 */, const Vec3 *X, Sink &&sink, double *N /* 60 */, double &f0_out) {
    PL_P35_MARK(0);
    double f0 = 0;
    for (int i = 0; i < 4; ++i)
        f0 += sqrt(xs[2 * i] * xs[2 * i] + xs[2 * i + 1] * xs[2 * i + 1]);
    f0 /= 4;
    f0_out = f0;
    double x[8];
    for (int i = 0; i < 8; ++i)
        x[i] = xs[i] / f0;

    // the 7 linear constraints on the 12 entries of P (row-major) as the columns of a 12 x 7 matrix
    {
        double A[12 * 7];
        for (int i = 0; i < 12 * 7; ++i)
            A[i] = 0.0;
        int row = 0;
        for (int i = 0; i < 4; ++i) {
            const double Xh[4] = {X[i].x, X[i].y, X[i].z, 1.0};
            for (int k = 0; k < 4; ++k) {
                A[row * 12 + k] = Xh[k];
                A[row * 12 + 8 + k] = -x[2 * i] * Xh[k];
            }
            ++row;
            if (i < 3) {
                for (int k = 0; k < 4; ++k) {
                    A[row * 12 + 4 + k] = Xh[k];
                    A[row * 12 + 8 + k] = -x[2 * i + 1] * Xh[k];
                }
                ++row;
            }
        }
        complement_basis_indexed<12, 7>(A, N);
    }
    PL_P35_MARK(1);
    {
        // rows of the left 3 x 3 block as vectors of linear polynomials: a[r][i] = sum_k N(4 r + i, k) x_k + N(4 r + i, 4)
        P35Lin a[3][3];
        PL_UNROLL
        for (int r = 0; r < 3; ++r) {
            PL_UNROLL
            for (int i = 0; i < 3; ++i) {
                PL_UNROLL
                for (int k = 0; k < 5; ++k)
                    a[r][i].c[k] = N[k * 12 + 4 * r + i];
            }
        }
        P35Cubic eq;
        int ne = 0;
        {
            P35Quad quad, d1;
            PL_UNROLL
            for (int qn = 0; qn < 4; ++qn) { // (unrolled throughout: every coefficient index is a constant, the polynomials stay in registers)
                if (qn < 3) {
                    p35_dot(a[qn == 2 ? 1 : 0], a[qn == 0 ? 1 : 2], quad);
                } else {
                    p35_dot(a[0], a[0], quad);
                    p35_dot(a[1], a[1], d1);
                    PL_UNROLL
                    for (int i = 0; i < 15; ++i)
                        quad.c[i] = quad.c[i] - d1.c[i];
                }
                PL_UNROLL
                for (int i = 0; i < 20; ++i)
                    eq.c[i] = 0.0;
                PL_UNROLL
                for (int i = 0; i < 15; ++i)
                    eq.c[20 + i] = quad.c[i];
                sink(ne++, eq);
                PL_UNROLL
                for (int k = 0; k < 4; ++k) {
                    P35Lin shift;
                    p35_zero(shift);
                    shift.c[k] = 1.0;
                    p35_mul(quad, shift, eq);
                    sink(ne++, eq);
                }
            }
        }
        P35Quad c23[3], c31[3];
        p35_cross(a[1], a[2], c23);
        p35_cross(a[2], a[0], c31);
        PL_UNROLL
        for (int i = 0; i < 3; ++i) {
            PL_UNROLL
            for (int j = 0; j < 3; ++j) {
                P35Cubic m1;
                p35_mul(c23[i], a[1][j], eq);
                p35_mul(c31[j], a[0][i], m1);
                PL_UNROLL
                for (int c = 0; c < 35; ++c)
                    eq.c[c] = eq.c[c] - m1.c[c];
                sink(ne++, eq);
            }
        }
    }
    PL_P35_MARK(2);
}
PL_HD void p35pf_setup(const double *xs /* 4 x 2 */, const Vec3 *X, const P35Work &w, double *N /* 60 */, double &f0_out) {
    p35pf_setup_t(xs, X, [&](int r, const P35Cubic &eq) { p35_store_row(w, r, eq); }, N, f0_out);
}

// the five rows of the action matrix that come out of the elimination: row i of the action matrix = - (reduced row of the pivot of
// eliminated monomial kP35ActionPivot[i]) restricted to the basis columns (kP35Shifted < 0: -sh - 1)
static constexpr uint8_t kP35ActionPivot[5] = {17, 9, 15, 18, 19};
constexpr int kP35ActionDoubles = 50; // E[i * 10 + j] = w.at(pivot_row[kP35ActionPivot[i]], kP35Basis[j])

// Stage 2: Gauss-Jordan over the 25 eliminated monomials: pivot = the largest remaining entry of the column among the unused rows
// (the first of equals).  false: degenerate sample.  E: the 50 entries stage 3 needs.
PL_HD bool p35pf_eliminate(const P35Work &w, double *E /* 50 */) {
    uint32_t used = 0;
    uint8_t pivot_row[25];
    for (int k = 0; k < 25; ++k) {
        const int col = kP35Elim[k];
        int pr = -1;
        double best = 0;
        for (int r = 0; r < kP35Rows; ++r) {
            const double v = fabs(w.at(r, col));
            if (!((used >> r) & 1u) && v > best)
                best = v, pr = r;
        }
        if (pr < 0 || best < 1e-13)
            return false; // degenerate sample
        used |= 1u << pr;
        pivot_row[k] = (uint8_t)pr;
        const double inv = 1.0 / w.at(pr, col);
        double prow[kP35Cols];
        PL_UNROLL
        for (int c = 0; c < kP35Cols; ++c) {
            prow[c] = w.at(pr, c) * inv;
            w.at(pr, c) = prow[c];
        }
        for (int r = 0; r < kP35Rows; ++r) {
            if (r == pr)
                continue;
            const double f = w.at(r, col);
            if (f != 0) {
                PL_UNROLL
                for (int c = 0; c < kP35Cols; ++c)
                    w.at(r, c) -= f * prow[c];
            }
        }
    }
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 10; ++j)
            E[i * 10 + j] = w.at(pivot_row[kP35ActionPivot[i]], kP35Basis[j]);
    PL_P35_MARK(3);
    return true;
}

// Stage 3: action matrix of x4 on the standard monomials (am), its real eigenvalues, the null vectors, the poses.  am, wk: two
// workspaces of 100 doubles (LDS on the device: as local arrays they were scratch memory behind the vector memory path).
PL_HD double p35pf_action_entry(const double *E /* 50 */, int k, int j) {
    const int sh = kP35Shifted[k];
    return sh >= 0 ? (j == sh ? 1.0 : 0.0) : -E[k * 10 + j];
}
PL_HD int p35pf_poses(const StridedArr &am, const StridedArr &wk, const double *ev, int nroots, const double *N /* 60 */, double f0,
                      P35Solution *out);
// Returns the number of solutions (<= 10), ascending in the eigenvalue.
PL_HD int p35pf_finish(const double *E /* 50 */, const double *N /* 60 */, double f0, const StridedArr &am, const StridedArr &wk,
                       P35Solution *out) {
    for (int k = 0; k < 10; ++k)
        for (int j = 0; j < 10; ++j)
            am[k * 10 + j] = p35pf_action_entry(E, k, j);
    double ev[10];
    for (int i = 0; i < 100; ++i)
        wk[i] = am[i];
    const int nroots = pl_real_eigenvalues<10>(wk, ev, 1e-8);
    PL_P35_MARK(4);
    return p35pf_poses(am, wk, ev, nroots, N, f0, out);
}
// one root: the null vector of (action matrix - eigenvalue), P = sum alpha_k N_k, the pose and focal length.  false: no solution
// the pose and focal length that belong to the null vector v of (action matrix - eigenvalue): P = sum alpha_k N_k
PL_HD bool p35pf_pose_from_null_vector(const double *v /* 10 */, const double *N /* 60 */, double f0, P35Solution &out);
PL_HD bool p35pf_pose_of_root(const StridedArr &am, const StridedArr &wk, double ev, const double *N /* 60 */, double f0,
                              P35Solution &out) {
    double v[10];
    for (int i = 0; i < 100; ++i)
        wk[i] = am[i];
    for (int i = 0; i < 10; ++i)
        wk[i * 10 + i] -= ev;
    pl_null_vector<10>(wk, v);
    return p35pf_pose_from_null_vector(v, N, f0, out);
}
PL_HD bool p35pf_pose_from_null_vector(const double *v, const double *N, double f0, P35Solution &out) {
    if (v[9] == 0)
        return false;
    const double al[5] = {v[5] / v[9], v[6] / v[9], v[7] / v[9], v[8] / v[9], 1.0};
    double P[12];
    for (int i = 0; i < 12; ++i) {
        double sum = 0;
        for (int k = 0; k < 5; ++k)
            sum += N[k * 12 + i] * al[k];
        P[i] = sum;
    }
    Mat3 R;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R.m[3 * r + c] = P[4 * r + c];
    double t[3] = {P[3], P[7], P[11]};
    const double det = R.m[0] * (R.m[4] * R.m[8] - R.m[5] * R.m[7]) - R.m[1] * (R.m[3] * R.m[8] - R.m[5] * R.m[6]) +
                       R.m[2] * (R.m[3] * R.m[7] - R.m[4] * R.m[6]);
    const double sgn = det < 0 ? -1.0 : 1.0;
    const double n3 = sqrt(R.m[6] * R.m[6] + R.m[7] * R.m[7] + R.m[8] * R.m[8]);
    if (!(n3 > 0))
        return false;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            R.m[3 * r + c] *= sgn / n3;
        t[r] *= sgn / n3;
    }
    const double n1 = sqrt(R.m[0] * R.m[0] + R.m[1] * R.m[1] + R.m[2] * R.m[2]);
    const double n2 = sqrt(R.m[3] * R.m[3] + R.m[4] * R.m[4] + R.m[5] * R.m[5]);
    const double focal = 0.5 * (n1 + n2);
    for (int c = 0; c < 3; ++c) {
        R.m[c] /= focal;
        R.m[3 + c] /= focal;
    }
    t[0] /= focal;
    t[1] /= focal;
    out.q = rotmat_to_quat(R);
    out.t = v3(t[0], t[1], t[2]);
    out.focal = focal * f0;
    return true;
}
// every root, ascending (on the device the roots of a sample go to the lanes of its wavefront, focal.hip k_focal_finish)
PL_HD int p35pf_poses(const StridedArr &am, const StridedArr &wk, const double *ev, int nroots, const double *N /* 60 */, double f0,
                      P35Solution *out) {
    int n = 0;
    for (int s = 0; s < nroots; ++s)
        if (p35pf_pose_of_root(am, wk, ev[s], N, f0, out[n]))
            ++n;
    PL_P35_MARK(5);
    return n;
}

// x: four image points (x, y) relative to the principal point - of the fourth only x is used -, X: the 3-D points; w: workspace of
// kP35WorkDoubles.  Returns the number of solutions (<= 10), ascending in the eigenvalue.
PL_HD int p35pf(const double *xs /* 4 x 2 */, const Vec3 *X, const P35Work &w, P35Solution *out) {
    double N[12 * 5], f0, E[kP35ActionDoubles];
    p35pf_setup(xs, X, w, N, f0);
    if (!p35pf_eliminate(w, E))
        return 0;
    // the elimination matrix is consumed: its storage holds the action matrix and the working copy of stage 3
    return p35pf_finish(E, N, f0, w.region(0), w.region(100), out);
}

} // namespace pl
