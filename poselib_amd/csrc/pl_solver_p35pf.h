// poselib_amd - P3.5Pf: absolute pose and focal length from three 2D-3D correspondences and the x coordinate of a fourth.
//
// The reference's solver restated (PoseLib/solvers/p35pf.cc:42-925; interface solvers/p35pf.h:39-54: image points relative to the
// principal point, scaled by their mean norm) so that it returns the reference's solutions - the same roots in the same order, to
// the last bit of oracle/_ref's build:
//   1. the seven linear constraints on P (3 x 4) as the columns of a 12 x 7 matrix, its null space N (12 x 5) from the Q of an
//      unpivoted Householder QR (p35pf.cc:68-84)                                                    p35pf_nullspace
//   2. 235 coefficients of four quadrics and five cubics in the null-space coordinates, the 25 x 35 template [C0 | C1], the last
//      five rows of C0^-1 C1 (p35pf.cc:86-871)                              pl_action_template.h + pl_focal_templates.h
//   3. the 10 x 10 action matrix of the multiplication by alpha_3 on the basis {1, x, xw, y, yw, z, zz, zw, w, ww}, its
//      eigenvalues (Hessenberg + Francis QR: EigenSolver), the real ones |imag| < 1e-6 in the routine's order (p35pf.cc:873-896)
//   4. per eigenvalue the other three unknowns from a 5 x 4 least-squares problem (p35pf.cc:7-40), P = N v, the pose and the focal
//      length (p35pf.cc:903-921)                                                                    p35pf_root_solution
// On the device: step 1 one lane per sample (k_focal_setup), steps 2 - 3 one wavefront per sample with the matrices in LDS, step 4
// one lane per root (focal.hip).  p35pf() below is the serial statement of the whole solver: the host build of tests/hostmath
// holds it against the oracle (oracle/src/solvers_focal.cc, written independently of this file) bit for bit.
#pragma once
#include "pl_action_template.h"
#include "pl_math.h"

namespace pl {

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


// EigenSolver(A, false).eigenvalues() (what the reference's template solvers call) in the operation order of oracle/eigen_shim:
// Householder reduction to Hessenberg form, Francis double-shift QR iteration with exceptional shifts after 10 and 20 sweeps.
// a_: n x n row-major (Arr: a pointer or anything indexable that yields double&), destroyed.  wr / wi: real and imaginary parts
// in the order the deflation leaves them.
template <int n, class Arr> PL_HD void pl_general_eigenvalues(Arr a_, double *wr, double *wi) {
#define PL_A(i, j) a_[(i) * n + (j)]
    for (int k = 0; k + 2 < n; ++k) {
        double tail = 0;
        for (int r = k + 2; r < n; ++r)
            tail += PL_A(r, k) * PL_A(r, k);
        if (tail <= 2.2250738585072014e-308)
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
                    if (its == 60) { // no convergence: what is left counts as NaN (and passes the reference's filter as such)
                        for (int i = 0; i <= nn; ++i) {
                            wr[i] = __builtin_nan("");
                            wi[i] = 0;
                        }
                        return;
                    }
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
}

struct P35Solution {
    Quat q;
    Vec3 t;
    double focal;
};

// Step 1.  xs: four image points (x, y) relative to the principal point; N: 12 x 5 column-major (column k at N[12 k ..], each
// column a 3 x 4 matrix, column-major); f0: the scale of the image points.
PL_HD void p35pf_nullspace(const double *xs /* 4 x 2 */, const Vec3 *X, double *N /* 60 */, double &f0_out) {
    double f0 = 0;
    for (int i = 0; i < 4; ++i)
        f0 += sqrt(xs[2 * i] * xs[2 * i] + xs[2 * i + 1] * xs[2 * i + 1]);
    f0 /= 4;
    f0_out = f0;
    double M[12 * 7], tau[7];
    for (int c = 0; c < 7; ++c) { // columns 2 i, 2 i + 1: the u and v rows of point i; column 6: the u row of the fourth point
        const int i = c < 6 ? c / 2 : 3, is_v = c < 6 ? c & 1 : 0;
        const double u = xs[2 * i + is_v] / f0;
        const double Xi[3] = {X[i].x, X[i].y, X[i].z};
        double *m = M + 12 * c;
        for (int k = 0; k < 3; ++k) {
            m[3 * k + 0] = is_v ? 0.0 : -Xi[k];
            m[3 * k + 1] = is_v ? -Xi[k] : 0.0;
            m[3 * k + 2] = Xi[k] * u;
        }
        m[9] = is_v ? 0.0 : -1.0;
        m[10] = is_v ? -1.0 : 0.0;
        m[11] = u;
    }
    for (int k = 0; k < 7; ++k) { // householderQr()
        double tail_sq = 0;
        for (int r = k + 1; r < 12; ++r)
            tail_sq += M[k * 12 + r] * M[k * 12 + r];
        const double c0 = M[k * 12 + k];
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0;
            for (int r = k + 1; r < 12; ++r)
                M[k * 12 + r] = 0;
        } else {
            double beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0)
                beta = -beta;
            for (int r = k + 1; r < 12; ++r)
                M[k * 12 + r] = M[k * 12 + r] / (c0 - beta);
            tau[k] = (beta - c0) / beta;
            M[k * 12 + k] = beta;
        }
        if (tau[k] != 0)
            for (int c = k + 1; c < 7; ++c) {
                double t = 0;
                for (int r = k + 1; r < 12; ++r)
                    t += M[k * 12 + r] * M[c * 12 + r];
                t += M[c * 12 + k];
                M[c * 12 + k] -= tau[k] * t;
                for (int r = k + 1; r < 12; ++r)
                    M[c * 12 + r] -= tau[k] * M[k * 12 + r] * t;
            }
    }
    for (int j = 0; j < 5; ++j) { // householderQ().rightCols(5)
        double *q = N + 12 * j;
        for (int r = 0; r < 12; ++r)
            q[r] = r == 7 + j ? 1.0 : 0.0;
        for (int k = 6; k >= 0; --k) {
            if (tau[k] == 0)
                continue;
            double t = 0;
            for (int r = k + 1; r < 12; ++r)
                t += M[k * 12 + r] * q[r];
            t += q[k];
            q[k] -= tau[k] * t;
            for (int r = k + 1; r < 12; ++r)
                q[r] -= tau[k] * M[k * 12 + r] * t;
        }
    }
}

// the rows of the action matrix that are not shifts come from rows 20 .. 24 of C0^-1 C1 (p35pf.cc:873-885)
static constexpr int8_t kP35Reduced[5] = {2, 4, 6, 7, 9};
// entry (k, j) of the action matrix; X(r, j): row 20 + r of column j of C0^-1 C1
template <class Tail> PL_HD double p35pf_action_entry(const Tail &X, int k, int j) {
    switch (k) {
    case 0:
        return j == 8 ? 1.0 : 0.0;
    case 1:
        return j == 2 ? 1.0 : 0.0;
    case 3:
        return j == 4 ? 1.0 : 0.0;
    case 5:
        return j == 7 ? 1.0 : 0.0;
    case 8:
        return j == 9 ? 1.0 : 0.0;
    default:
        return -X(k == 2 ? 0 : k == 4 ? 1 : k == 6 ? 2 : k == 7 ? 3 : 4, j);
    }
}

// Step 4 for one eigenvalue z: AM the action matrix (10 x 10 row-major), N, f0 from step 1.
template <class Arr> PL_HD void p35pf_root_solution(const Arr &AM, double z0, const double *N, double f0, P35Solution &out) {
    const double z1 = z0 * z0, z2 = z1 * z0;
    double AA[25], rhs[5], s[4]; // column-major 5 x 5
    for (int r = 0; r < 5; ++r) {
        const int row = kP35Reduced[r] * 10;
        AA[0 * 5 + r] = AM[row + 1] + z0 * AM[row + 2];
        AA[1 * 5 + r] = AM[row + 3] + z0 * AM[row + 4];
        AA[2 * 5 + r] = AM[row + 6];
        AA[3 * 5 + r] = AM[row + 5] + z0 * AM[row + 7];
        AA[4 * 5 + r] = AM[row + 0] + z0 * AM[row + 8] + z1 * AM[row + 9];
    }
    AA[0] = AA[0] - z1;
    AA[6] = AA[6] - z1;
    AA[12] = AA[12] - z0;
    AA[18] = AA[18] - z1;
    AA[24] = AA[24] - z2;
    for (int r = 0; r < 5; ++r)
        rhs[r] = -AA[20 + r];
    colpiv_qr_solve<5, 4>(AA, rhs, s);
    const double v[5] = {s[0], s[1], s[3], z0, 1.0};
    double P[12]; // 3 x 4 column-major
    for (int e = 0; e < 12; ++e) {
        double sum = N[e] * v[0];
        for (int k = 1; k < 5; ++k)
            sum += N[k * 12 + e] * v[k];
        P[e] = sum;
    }
    const double det = P[0] * (P[4] * P[8] - P[7] * P[5]) - P[3] * (P[1] * P[8] - P[7] * P[2]) + P[6] * (P[1] * P[5] - P[4] * P[2]);
    if (det < 0)
        for (int e = 0; e < 12; ++e)
            P[e] = -P[e];
    double n3 = 0;
    for (int c = 0; c < 3; ++c)
        n3 += P[3 * c + 2] * P[3 * c + 2];
    n3 = sqrt(n3);
    for (int e = 0; e < 12; ++e)
        P[e] = P[e] / n3;
    double n1 = 0, n2 = 0;
    for (int c = 0; c < 3; ++c) {
        n1 += P[3 * c] * P[3 * c];
        n2 += P[3 * c + 1] * P[3 * c + 1];
    }
    const double focal = (sqrt(n1) + sqrt(n2)) / 2;
    for (int c = 0; c < 4; ++c) {
        P[3 * c] = P[3 * c] / focal;
        P[3 * c + 1] = P[3 * c + 1] / focal;
    }
    Mat3 R;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R.m[3 * r + c] = P[3 * c + r];
    out.q = rotmat_to_quat(R);
    out.t = v3(P[9], P[10], P[11]);
    out.focal = focal * f0;
}

// The whole solver, serially (host; tests/hostmath).  Returns the number of solutions (<= 10) in the reference's order.
PL_HD int p35pf(const double *xs /* 4 x 2 */, const Vec3 *X, P35Solution *out) {
    double N[60], f0;
    p35pf_nullspace(xs, X, N, f0);
    constexpr int S = 36;
    double coef[kP35Coeffs], C[25 * S];
    for (int k = 0; k < kP35Coeffs; ++k)
        coef[k] = template_coefficient<false>(N, kP35TermStart, kP35TermPacked, k);
    for (int e = 0; e < 25 * S; ++e)
        C[e] = 0.0;
    for (int c = 0; c < kP35Cols; ++c)
        for (int e = kP35ColStart[c]; e < kP35ColStart[c + 1]; ++e)
            C[kP35EntryRow[e] * S + c] = coef[kP35EntryCoeff[e]];
    lu_solve_tail<25, 35, S, 5>(C);
    double AM[100], H[100], wr[10], wi[10];
    auto tail = [&](int r, int j) { return C[(20 + r) * S + 25 + j]; };
    for (int k = 0; k < 10; ++k)
        for (int j = 0; j < 10; ++j)
            H[k * 10 + j] = AM[k * 10 + j] = p35pf_action_entry(tail, k, j);
    pl_general_eigenvalues<10, double *>(H, wr, wi);
    int n = 0;
    for (int i = 0; i < 10; ++i)
        if (fabs(wi[i]) < 1e-6)
            p35pf_root_solution(AM, wr[i], N, f0, out[n++]);
    return n;
}

} // namespace pl
