// poselib_amd - relative pose of two views that share ONE unknown focal length, from six correspondences.
//
// Interface of the reference's solver (PoseLib/solvers/relpose_6pt_focal.h:12-13, .cc:1083-1144: unit bearings with the principal
// point at the origin, null space of the six epipolar constraints from a full-pivoting Householder QR, F = N0 + x N1 + y N2,
// solutions with w = 1 / f^2 < 1e-8 dropped, E = K F K, motion_from_essential on the bearings at that focal length; solutions
// ascending in y - the reference's Sturm bisection finds the roots of its action variable from left to right, :1071-1078).
// The ALGORITHM is not the reference's generated 31 x 46 elimination template but this project's own, derived from first
// principles (SharedFocalRelativePoseEstimator): with Q = diag(1, 1, w)
//     det F = 0,      2 (F Q F^T) Q F - trace(F Q F^T Q) F = 0
// are ten cubics in (x, y), the nine of the trace constraint quadratic in w: (C0 + w C1 + w^2 C2) m = 0 over the ten monomials of
// degree <= 3.  The w^2 part of every equation carries the factor F33, so C2 has rank <= 6: six steps of Gaussian elimination on
// C2 leave six rows of degree 2 in w, three of degree 1 and the determinant of degree 0 - the row degrees add up to the 15
// solutions, and with L the matrix of the leading row coefficients and u = L m the problem is the standard eigenvalue problem of a
// 15 x 15 companion matrix in the state (u_0..u_8, w u_0..w u_5).  Its real eigenvalues (balanced, Hessenberg + Francis QR) are
// the w, the null vector of C0 + w C1 + w^2 C2 holds (x, y).
// Operation for operation the same as the oracle's statement of this algorithm (oracle/src/solvers_focal.cc, written on dense
// row-major arrays); tests/test_hostmath_vs_oracle.py compares the two bit for bit on the host.
//
// Storage: the matrices of a sample (kSixWorkDoubles doubles) live in a workspace the caller provides, element e at
// work[e * stride]: on the device one lane = one sample and stride = samples of the launch, so that the 64 lanes of a wavefront
// touch consecutive doubles.
#pragma once
#include "pl_solver_p35pf.h" // complement_basis_indexed, pl_real_eigenvalues, pl_null_vector
#include "pl_solver_rel.h"   // motion_from_essential_emit

namespace pl {

// monomials x^a y^b of degree <= 3, graded: x3 x2y xy2 y3 x2 xy y2 x y 1; kSixProd[i][j] = index of the product (-1: degree > 3)
static constexpr int8_t kSixProd[10][10] = {
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, 0}, {-1, -1, -1, -1, -1, -1, -1, -1, -1, 1}, {-1, -1, -1, -1, -1, -1, -1, -1, -1, 2},
    {-1, -1, -1, -1, -1, -1, -1, -1, -1, 3}, {-1, -1, -1, -1, -1, -1, -1, 0, 1, 4},   {-1, -1, -1, -1, -1, -1, -1, 1, 2, 5},
    {-1, -1, -1, -1, -1, -1, -1, 2, 3, 6},   {-1, -1, -1, -1, 0, 1, 2, 4, 5, 7},      {-1, -1, -1, -1, 1, 2, 3, 5, 6, 8},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9},
};

struct SixPoly { // dense over the ten monomials
    double c[10];
};
// r = p * q, products added in ascending (i, j); FP / FQ: first monomial a factor can hold (7: linear, 4: quadratic)
template <int FP, int FQ> PL_HD void six_mul(const SixPoly &p, const SixPoly &q, SixPoly &r) {
    // (unrolled: kSixProd folds to constants and the coefficients stay in registers - a run-time index puts them into scratch memory)
    PL_UNROLL
    for (int i = 0; i < 10; ++i)
        r.c[i] = 0.0;
    PL_UNROLL
    for (int i = FP; i < 10; ++i) {
        PL_UNROLL
        for (int j = FQ; j < 10; ++j)
            r.c[kSixProd[i][j]] += p.c[i] * q.c[j];
    }
}

// workspace of one sample
constexpr int kSixC = 0;        // C[3][100]: the problem, kept for the null vectors
constexpr int kSixCw = 300;     // copy that the row reduction destroys
constexpr int kSixT = 600;      // 15 x 15 companion
constexpr int kSixA = 825;      // 10 x 10: L^T, later C0 + w C1 + w^2 C2
constexpr int kSixB = 925;      // 10 x 15 right-hand sides
constexpr int kSixWorkDoubles = 1075;
typedef StridedArr SixWork; // (pl_solver_p35pf.h)

// C[k][r * 10 + c]: coefficient of monomial c in the w^k part of equation r (0: det F, 1 + 3 i + j: entry (i, j) of the trace
// constraint), every equation scaled to unit maximum.  nb: 9 x 3 null-space basis, column-major, vec index e = 3 col + row.
PL_HD void six_equations(const double *nb, const SixWork &C) {
    SixPoly F[3][3];
    PL_UNROLL
    for (int i = 0; i < 3; ++i)
        PL_UNROLL
        for (int j = 0; j < 3; ++j) {
            const int e = 3 * j + i;
            PL_UNROLL
            for (int k = 0; k < 7; ++k)
                F[i][j].c[k] = 0.0;
            F[i][j].c[9] = nb[e];
            F[i][j].c[7] = nb[9 + e];
            F[i][j].c[8] = nb[18 + e];
        }
    SixPoly G0[3][3], G1[3][3], a, b, c;
    PL_UNROLL
    for (int i = 0; i < 3; ++i)
        PL_UNROLL
        for (int j = 0; j < 3; ++j) {
            six_mul<7, 7>(F[i][0], F[j][0], a);
            six_mul<7, 7>(F[i][1], F[j][1], b);
            PL_UNROLL
            for (int k = 0; k < 10; ++k)
                G0[i][j].c[k] = a.c[k] + b.c[k];
            six_mul<7, 7>(F[i][2], F[j][2], G1[i][j]);
        }
    SixPoly tr0, tr1, tr2;
    PL_UNROLL
    for (int k = 0; k < 10; ++k) {
        tr0.c[k] = G0[0][0].c[k] + G0[1][1].c[k];
        tr1.c[k] = (G1[0][0].c[k] + G1[1][1].c[k]) + G0[2][2].c[k];
        tr2.c[k] = G1[2][2].c[k];
    }
    auto store = [&](int r, const SixPoly *e) { // one equation: scale and store
        double mx = 0;
        PL_UNROLL
        for (int k = 0; k < 3; ++k)
            PL_UNROLL
            for (int m = 0; m < 10; ++m)
                mx = fmax(mx, fabs(e[k].c[m]));
        const double s = mx > 0 ? 1.0 / mx : 0.0;
        PL_UNROLL
        for (int k = 0; k < 3; ++k)
            PL_UNROLL
            for (int m = 0; m < 10; ++m)
                C[k * 100 + r * 10 + m] = e[k].c[m] * s;
    };
    SixPoly eq[3];
    { // det F = (F00 (F11 F22 - F12 F21) - F01 (F10 F22 - F12 F20)) + F02 (F10 F21 - F11 F20)
        SixPoly m0, m1, d0, d1, d2;
        auto minor = [&](int r0, int c0, int r1, int c1, SixPoly &out) { // F[r0][c0] F[r1][c1] - F[r0][c1] F[r1][c0]
            six_mul<7, 7>(F[r0][c0], F[r1][c1], m0);
            six_mul<7, 7>(F[r0][c1], F[r1][c0], m1);
            PL_UNROLL
            for (int k = 0; k < 10; ++k)
                out.c[k] = m0.c[k] - m1.c[k];
        };
        minor(1, 1, 2, 2, a);
        six_mul<7, 4>(F[0][0], a, d0);
        minor(1, 0, 2, 2, a);
        six_mul<7, 4>(F[0][1], a, d1);
        minor(1, 0, 2, 1, a);
        six_mul<7, 4>(F[0][2], a, d2);
        PL_UNROLL
        for (int k = 0; k < 10; ++k) {
            eq[0].c[k] = (d0.c[k] - d1.c[k]) + d2.c[k];
            eq[1].c[k] = 0.0;
            eq[2].c[k] = 0.0;
        }
        store(0, eq);
    }
    PL_UNROLL
    for (int i = 0; i < 3; ++i)
        PL_UNROLL
        for (int j = 0; j < 3; ++j) {
            SixPoly t;
            // w^0: 2 (G0_i0 F_0j + G0_i1 F_1j) - tr0 F_ij
            six_mul<4, 7>(G0[i][0], F[0][j], a);
            six_mul<4, 7>(G0[i][1], F[1][j], b);
            six_mul<4, 7>(tr0, F[i][j], t);
            PL_UNROLL
            for (int k = 0; k < 10; ++k)
                eq[0].c[k] = 2.0 * (a.c[k] + b.c[k]) - t.c[k];
            // w^1: 2 ((G1_i0 F_0j + G1_i1 F_1j) + G0_i2 F_2j) - tr1 F_ij
            six_mul<4, 7>(G1[i][0], F[0][j], a);
            six_mul<4, 7>(G1[i][1], F[1][j], b);
            six_mul<4, 7>(G0[i][2], F[2][j], c);
            six_mul<4, 7>(tr1, F[i][j], t);
            PL_UNROLL
            for (int k = 0; k < 10; ++k)
                eq[1].c[k] = 2.0 * ((a.c[k] + b.c[k]) + c.c[k]) - t.c[k];
            // w^2: 2 G1_i2 F_2j - tr2 F_ij
            six_mul<4, 7>(G1[i][2], F[2][j], a);
            six_mul<4, 7>(tr2, F[i][j], t);
            PL_UNROLL
            for (int k = 0; k < 10; ++k)
                eq[2].c[k] = 2.0 * a.c[k] - t.c[k];
            store(1 + 3 * i + j, eq);
        }
}

// 15 x 15 companion matrix T (row-major) of the row-reduced problem; false: degenerate sample.  Cw is destroyed; A, B scratch.
PL_HD bool six_companion(const SixWork &Cw, const SixWork &T, const SixWork &A, const SixWork &B) {
#define PL_C(k, r, col) Cw[(k) * 100 + (r) * 10 + (col)]
#define PL_LA(r, col) A[(r) * 10 + (col)]
#define PL_RB(r, col) B[(r) * 15 + (col)]
    for (int k = 0; k < 6; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = 1 + k; r < 10; ++r)
            for (int col = 0; col < 10; ++col) {
                const double v = fabs(PL_C(2, r, col));
                if (v > best)
                    best = v, pr = r, pc = col;
            }
        if (pr < 0)
            return false;
        if (pr != 1 + k)
            for (int m = 0; m < 3; ++m)
                for (int col = 0; col < 10; ++col) {
                    const double t = PL_C(m, 1 + k, col);
                    PL_C(m, 1 + k, col) = PL_C(m, pr, col);
                    PL_C(m, pr, col) = t;
                }
        for (int r = 2 + k; r < 10; ++r) {
            const double f = PL_C(2, r, pc) / PL_C(2, 1 + k, pc);
            if (f == 0)
                continue;
            for (int m = 0; m < 3; ++m)
                for (int col = 0; col < 10; ++col)
                    PL_C(m, r, col) -= f * PL_C(m, 1 + k, col);
            PL_C(2, r, pc) = 0;
        }
    }
    for (int col = 0; col < 10; ++col) { // A = L^T, B = (right-hand rows)^T
        for (int i = 0; i < 6; ++i)
            PL_LA(col, i) = PL_C(2, 1 + i, col);
        for (int j = 0; j < 3; ++j)
            PL_LA(col, 6 + j) = PL_C(1, 7 + j, col);
        PL_LA(col, 9) = PL_C(0, 0, col);
        for (int i = 0; i < 6; ++i) {
            PL_RB(col, i) = PL_C(0, 1 + i, col);
            PL_RB(col, 6 + i) = PL_C(1, 1 + i, col);
        }
        for (int j = 0; j < 3; ++j)
            PL_RB(col, 12 + j) = PL_C(0, 7 + j, col);
    }
    for (int k = 0; k < 10; ++k) { // LU with partial pivoting
        int pr = k;
        double best = fabs(PL_LA(k, k));
        for (int r = k + 1; r < 10; ++r) {
            const double v = fabs(PL_LA(r, k));
            if (v > best)
                best = v, pr = r;
        }
        if (best == 0)
            return false;
        if (pr != k) {
            for (int col = 0; col < 10; ++col) {
                const double t = PL_LA(k, col);
                PL_LA(k, col) = PL_LA(pr, col);
                PL_LA(pr, col) = t;
            }
            for (int col = 0; col < 15; ++col) {
                const double t = PL_RB(k, col);
                PL_RB(k, col) = PL_RB(pr, col);
                PL_RB(pr, col) = t;
            }
        }
        for (int r = k + 1; r < 10; ++r) {
            const double f = PL_LA(r, k) / PL_LA(k, k);
            if (f == 0)
                continue;
            for (int col = k + 1; col < 10; ++col)
                PL_LA(r, col) -= f * PL_LA(k, col);
            for (int col = 0; col < 15; ++col)
                PL_RB(r, col) -= f * PL_RB(k, col);
        }
    }
    for (int col = 0; col < 15; ++col)
        for (int r = 9; r >= 0; --r) {
            double s = PL_RB(r, col);
            for (int m = r + 1; m < 10; ++m)
                s -= PL_LA(r, m) * PL_RB(m, col);
            PL_RB(r, col) = s / PL_LA(r, r);
        }
    for (int e = 0; e < 225; ++e)
        T[e] = 0.0;
    for (int i = 0; i < 6; ++i)
        T[i * 15 + 9 + i] = 1.0;
    for (int j = 0; j < 3; ++j)
        for (int m = 0; m < 9; ++m)
            T[(6 + j) * 15 + m] = -PL_RB(m, 12 + j);
    for (int i = 0; i < 6; ++i) {
        for (int m = 0; m < 6; ++m)
            T[(9 + i) * 15 + 9 + m] = -PL_RB(m, 6 + i);
        for (int m = 0; m < 9; ++m) {
            double s = -PL_RB(m, i);
            for (int j = 0; j < 3; ++j)
                s += PL_RB(6 + j, 6 + i) * PL_RB(m, 12 + j);
            T[(9 + i) * 15 + m] = s;
        }
    }
#undef PL_C
#undef PL_LA
#undef PL_RB
    return true;
}

// Parlett-Reinsch balancing without the permutation step: similarity scaling by powers of two (exact)
template <int n, class Arr> PL_HD void pl_balance_pow2(Arr a) {
    bool done = false;
    for (int sweep = 0; sweep < 64 && !done; ++sweep) { // (typically 3 - 6 sweeps; the cap bounds the loop on overflowing input)
        done = true;
        for (int i = 0; i < n; ++i) {
            double c = 0, r = 0;
            for (int j = 0; j < n; ++j)
                if (j != i) {
                    c += fabs(a[j * n + i]);
                    r += fabs(a[i * n + j]);
                }
            if (c == 0 || r == 0)
                continue;
            double g = r / 2.0, f = 1.0;
            const double s = c + r;
            while (c < g) {
                f *= 2.0;
                c *= 4.0;
            }
            g = r * 2.0;
            while (c >= g) {
                f /= 2.0;
                c /= 4.0;
            }
            if ((c + r) / f < 0.95 * s) {
                done = false;
                g = 1.0 / f;
                for (int j = 0; j < n; ++j)
                    a[i * n + j] *= g;
                for (int j = 0; j < n; ++j)
                    a[j * n + i] *= f;
            }
        }
    }
}

constexpr int kSixMaxModels = 60; // 15 solutions x 4 poses (relpose_6pt_focal.cc:1101)

// The solver in three stages - on the device three kernels with three different footprints (sfocal.hip), on the host and in
// relpose_6pt_shared_focal() below one after the other.
//
// Stage 1: null space nb (9 x 3) of the six epipolar constraints, the ten equations C (kept for stage 3) and the 15 x 15 companion
// matrix T.  false: a vanishing pivot in the row reduction (no models).  Needs the whole workspace w.
// (the first half of stage 1 alone: null space and equations - on the device the row reduction is done by a wavefront, sfocal.hip)
PL_HD void six_nullspace_equations(const Vec3 *x1, const Vec3 *x2, const SixWork &C, double *nb /* 27 */) {
    {
        double A[54];
        for (int i = 0; i < 6; ++i) {
            const double a[3] = {x1[i].x, x1[i].y, x1[i].z};
            for (int j = 0; j < 3; ++j) {
                A[i * 9 + 3 * j + 0] = a[j] * x2[i].x;
                A[i * 9 + 3 * j + 1] = a[j] * x2[i].y;
                A[i * 9 + 3 * j + 2] = a[j] * x2[i].z;
            }
        }
        complement_basis_indexed<9, 6>(A, nb);
    }
    six_equations(nb, C);
}
PL_HD bool six_setup(const Vec3 *x1, const Vec3 *x2, const SixWork &w, double *nb /* 27 */) {
    {
        double A[54];
        for (int i = 0; i < 6; ++i) {
            const double a[3] = {x1[i].x, x1[i].y, x1[i].z};
            for (int j = 0; j < 3; ++j) {
                A[i * 9 + 3 * j + 0] = a[j] * x2[i].x;
                A[i * 9 + 3 * j + 1] = a[j] * x2[i].y;
                A[i * 9 + 3 * j + 2] = a[j] * x2[i].z;
            }
        }
        complement_basis_indexed<9, 6>(A, nb);
    }
    const SixWork C = w.at(kSixC), Cw = w.at(kSixCw), T = w.at(kSixT), A = w.at(kSixA), B = w.at(kSixB);
    six_equations(nb, C);
    for (int e = 0; e < 300; ++e)
        Cw[e] = C[e];
    return six_companion(Cw, T, A, B);
}
// Stage 2: the real eigenvalues w = 1 / f^2 of the companion matrix T (15 x 15, destroyed), ascending.  Needs T only.
PL_HD int six_eigenvalues(const SixWork &T, double *ev /* 15 */) {
    for (int e = 0; e < 225; ++e)
        if (!isfinite(T[e]))
            return 0; // (a vanishing pivot: the balancing below would not terminate on an infinite entry)
#if defined(PL_EIG_SHADOW_CHECK) && !defined(__HIPCC__)
    { // tests/hostmath: the packed balancing (pl_eigen_packed.h) on a copy, every element compared
        double shadow[225 + 60];
        for (int e = 0; e < 225; ++e)
            shadow[e] = T[e];
        EigFlatHost<15> cx{shadow};
        pl_balance_pow2_packed<15>(cx, true);
        pl_balance_pow2<15>(T);
        bool same = true;
        for (int e = 0; e < 225; ++e)
            same = same && std::memcmp(&shadow[e], &T[e], sizeof(double)) == 0;
        pl_eig_shadow_counters[2]++;
        pl_eig_shadow_counters[3] += same ? 0 : 1;
        return pl_real_eigenvalues<15>(T, ev, 1e-8);
    }
#endif
    pl_balance_pow2<15>(T);
    return pl_real_eigenvalues<15>(T, ev, 1e-8);
}
// Stage 3: (x, y) of every root from the null vector of C0 + w C1 + w^2 C2, the essential matrices, the poses.  C: the equations
// of stage 1 (read only), A: a workspace of 100 doubles.
// one root w: false when it is dropped (w < 1e-8: focal length beyond 1e4; no null vector)
PL_HD bool six_root_xy(const SixWork &C, const SixWork &A, double wv, double &x, double &y) {
    if (wv < 1e-8)
        return false;
    double v[10];
    for (int e = 0; e < 100; ++e)
        A[e] = C[e] + wv * (C[100 + e] + wv * C[200 + e]);
    pl_null_vector<10>(A, v);
    if (v[9] == 0)
        return false;
    x = v[7] / v[9], y = v[8] / v[9];
    return true;
}
// stable insertion of (x, y, w) into the list of ns solutions, ascending in y
PL_HD void six_insert_solution(double *sx, double *sy, double *sw, int &ns, double x, double y, double wv) {
    int j = ns++;
    while (j > 0 && sy[j - 1] > y) {
        sx[j] = sx[j - 1], sy[j] = sy[j - 1], sw[j] = sw[j - 1];
        --j;
    }
    sx[j] = x, sy[j] = y, sw[j] = wv;
}
// the poses of one solution: emit(q, t, focal) for every one of them (<= 4), in the reference's order
template <class Emit>
PL_HD void six_solution_poses(const Vec3 *x1, const Vec3 *x2, const double *nb, double sx, double sy, double sw, Emit &&emit) {
    const double focal = sqrt(1.0 / sw);
    double Fv[9], nrm = 0;
    for (int e = 0; e < 9; ++e) {
        Fv[e] = nb[e] + sx * nb[9 + e] + sy * nb[18 + e];
        nrm += Fv[e] * Fv[e];
    }
    nrm = sqrt(nrm);
    Mat3 E;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double ki = i < 2 ? focal : 1.0, kj = j < 2 ? focal : 1.0;
            E(i, j) = ki * ((Fv[3 * j + i] / nrm) * kj);
        }
    Vec3 u1[6], u2[6];
    for (int i = 0; i < 6; ++i) {
        u1[i] = normalized(v3(x1[i].x / focal, x1[i].y / focal, x1[i].z));
        u2[i] = normalized(v3(x2[i].x / focal, x2[i].y / focal, x2[i].z));
    }
    motion_from_essential_emit<6>(E, u1, u2, [&](Quat q, Vec3 t) { emit(q, t, focal); });
}
// emit(q, t, focal) is called for every model, in the reference's order (on the device the roots, then the solutions, of a sample
// go to the lanes of its wavefront: sfocal.hip k_sfocal_finish)
template <class Emit>
PL_HD int six_finish(const Vec3 *x1, const Vec3 *x2, const double *nb, const SixWork &C, const SixWork &A, const double *ev, int nroots,
                     Emit &&emit) {
    double sx[15], sy[15], sw[15];
    int ns = 0;
    for (int s = 0; s < nroots; ++s) {
        double x, y;
        if (six_root_xy(C, A, ev[s], x, y))
            six_insert_solution(sx, sy, sw, ns, x, y, ev[s]);
    }
    int n = 0;
    for (int s = 0; s < ns; ++s)
        six_solution_poses(x1, x2, nb, sx[s], sy[s], sw[s], [&](Quat q, Vec3 t, double focal) {
            emit(q, t, focal);
            ++n;
        });
    return n;
}

// x1, x2: six unit bearings.  emit(q, t, focal) is called for every model, in the reference's order.  Returns their number.
template <class Emit> PL_HD int relpose_6pt_shared_focal(const Vec3 *x1, const Vec3 *x2, const SixWork &w, Emit &&emit) {
    double nb[27], ev[15];
    if (!six_setup(x1, x2, w, nb))
        return 0;
    const int nroots = six_eigenvalues(w.at(kSixT), ev);
    return six_finish(x1, x2, nb, w.at(kSixC), w.at(kSixA), ev, nroots, emit);
}

} // namespace pl
