// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// P3.5Pf - absolute pose and focal length from three 2D-3D correspondences and one coordinate of a fourth - restated FROM FIRST
// PRINCIPLES, not from the reference's generated elimination template (solvers/p35pf.cc: 235 coefficient polynomials and the index
// tables of an automatic generator, which cannot be re-derived by hand and must not be copied).  What is taken from the reference
// is the problem statement and the conventions of its interface (solvers/p35pf.h:39-54: image points relative to the principal
// point, the x coordinate of the fourth point, scaling of the image points by their mean norm; p35pf.cc:903-921: det > 0,
// |third row| = 1, focal = mean norm of the first two rows).
//
// Derivation.  P = s K [R | t], K = diag(f, f, 1), satisfies 7 linear equations (u (P3 . Xh) = P1 . Xh for all four points,
// v (P3 . Xh) = P2 . Xh for the first three): P = sum_k alpha_k N_k over the 5-dimensional null space, alpha_5 = 1 - four unknowns
// x1..x4.  With a1, a2, a3 the rows of the left 3 x 3 block A = s K R:
//     A A^T = s^2 diag(f^2, f^2, 1)       =>  a1.a2 = 0,  a1.a3 = 0,  a2.a3 = 0,  |a1|^2 = |a2|^2                 (4 quadrics)
//     cof(A) = det(A) A^-T = det(A) D^-1 A  =>  a2 x a3 = k a1 and a3 x a1 = k a2 with the SAME k
//                                           =>  (a2 x a3)_i (a2)_j = (a3 x a1)_j (a1)_i,  i, j = 1..3                (9 cubics)
// The quadrics alone have 16 roots; six of them have a1 || a2 isotropic (f = 0) and are removed exactly by the cubics.  The four
// quadrics times {1, x1, x2, x3, x4} and the nine cubics are 29 equations, linear in the 35 monomials of degree <= 3: rank 25, so the
// other 25 monomials are expressed in the ten standard monomials {x3^2, x1 x4, x2 x4, x3 x4, x4^2, x1, x2, x3, x4, 1} by one
// Gauss-Jordan elimination, the multiplication by x4 becomes a 10 x 10 matrix on them, its real eigenvalues are the x4 of the
// solutions and the eigenvectors hold (x1, x2, x3).  scripts/exp/p35pf_compact_template.py is the numpy experiment behind this
// (1772 of 1778 solutions of the reference's p35pf reproduced on 400 random minimal problems).
#include "solvers.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace orc {

namespace {

constexpr int NM = 35; // monomials of degree <= 3 in x1..x4

struct Monomials {
    int e[NM][4];
    int index[4][4][4][4];
    int basis[10], elim[25];
    Monomials() {
        std::memset(index, -1, sizeof(index));
        int n = 0;
        for (int d = 3; d >= 0; --d) // graded, any fixed order inside a degree
            for (int a = d; a >= 0; --a)
                for (int b = d - a; b >= 0; --b)
                    for (int c = d - a - b; c >= 0; --c) {
                        const int dd = d - a - b - c;
                        e[n][0] = a, e[n][1] = b, e[n][2] = c, e[n][3] = dd;
                        index[a][b][c][dd] = n++;
                    }
        const int B[10][4] = {{0, 0, 2, 0}, {1, 0, 0, 1}, {0, 1, 0, 1}, {0, 0, 1, 1}, {0, 0, 0, 2},
                              {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {0, 0, 0, 0}};
        bool is_basis[NM] = {false};
        for (int k = 0; k < 10; ++k) {
            basis[k] = index[B[k][0]][B[k][1]][B[k][2]][B[k][3]];
            is_basis[basis[k]] = true;
        }
        int m = 0;
        for (int i = 0; i < NM; ++i)
            if (!is_basis[i])
                elim[m++] = i;
    }
    int of(const int *a, const int *b) const { return index[a[0] + b[0]][a[1] + b[1]][a[2] + b[2]][a[3] + b[3]]; }
};
const Monomials &mono() {
    static const Monomials M;
    return M;
}

struct Poly { // dense over the 35 monomials (entries above the polynomial's degree stay zero)
    double c[NM];
    Poly() { std::memset(c, 0, sizeof(c)); }
};
Poly mul(const Poly &p, const Poly &q) {
    const Monomials &M = mono();
    Poly r;
    for (int i = 0; i < NM; ++i)
        if (p.c[i] != 0)
            for (int j = 0; j < NM; ++j)
                if (q.c[j] != 0)
                    r.c[M.of(M.e[i], M.e[j])] += p.c[i] * q.c[j];
    return r;
}
Poly sub(const Poly &p, const Poly &q) {
    Poly r;
    for (int i = 0; i < NM; ++i)
        r.c[i] = p.c[i] - q.c[i];
    return r;
}
Poly add(const Poly &p, const Poly &q) {
    Poly r;
    for (int i = 0; i < NM; ++i)
        r.c[i] = p.c[i] + q.c[i];
    return r;
}
struct PVec {
    Poly v[3];
};
Poly dot(const PVec &a, const PVec &b) { return add(add(mul(a.v[0], b.v[0]), mul(a.v[1], b.v[1])), mul(a.v[2], b.v[2])); }
PVec cross(const PVec &a, const PVec &b) {
    PVec r;
    r.v[0] = sub(mul(a.v[1], b.v[2]), mul(a.v[2], b.v[1]));
    r.v[1] = sub(mul(a.v[2], b.v[0]), mul(a.v[0], b.v[2]));
    r.v[2] = sub(mul(a.v[0], b.v[1]), mul(a.v[1], b.v[0]));
    return r;
}

// eigenvalues of a real n x n matrix (row-major, destroyed): Householder reduction to Hessenberg form + Francis double-shift QR
// (EISPACK orthes / hqr in their textbook form); returns the REAL eigenvalues (|imag| <= tol (1 + |real|))
int real_eigenvalues(double *a_, int n, double *out, double tol) {
    auto a = [&](int i, int j) -> double & { return a_[i * n + j]; };
    for (int k = 0; k + 2 < n; ++k) {
        double tail = 0;
        for (int r = k + 2; r < n; ++r)
            tail += a(r, k) * a(r, k);
        if (tail <= 1e-300)
            continue;
        const double c0 = a(k + 1, k);
        double beta = std::sqrt(c0 * c0 + tail);
        if (c0 >= 0)
            beta = -beta;
        std::vector<double> v(n, 0.0);
        v[k + 1] = 1.0;
        for (int r = k + 2; r < n; ++r)
            v[r] = a(r, k) / (c0 - beta);
        const double tau = (beta - c0) / beta;
        for (int c = 0; c < n; ++c) {
            double t = 0;
            for (int r = k + 1; r < n; ++r)
                t += v[r] * a(r, c);
            for (int r = k + 1; r < n; ++r)
                a(r, c) -= tau * v[r] * t;
        }
        for (int r = 0; r < n; ++r) {
            double t = 0;
            for (int c = k + 1; c < n; ++c)
                t += a(r, c) * v[c];
            for (int c = k + 1; c < n; ++c)
                a(r, c) -= tau * t * v[c];
        }
        a(k + 1, k) = beta;
        for (int r = k + 2; r < n; ++r)
            a(r, k) = 0;
    }
    std::vector<double> wr(n, 0.0), wi(n, 0.0);
    const double eps = 2.220446049250313e-16;
    double anorm = 0;
    for (int i = 0; i < n; ++i)
        for (int j = std::max(i - 1, 0); j < n; ++j)
            anorm += std::fabs(a(i, j));
    int nn = n - 1;
    double t = 0, p = 0, q = 0, r = 0, s = 0, w = 0, x = 0, y = 0, z = 0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                s = std::fabs(a(l - 1, l - 1)) + std::fabs(a(l, l));
                if (s == 0)
                    s = anorm;
                if (std::fabs(a(l, l - 1)) <= eps * s) {
                    a(l, l - 1) = 0;
                    break;
                }
            }
            x = a(nn, nn);
            if (l == nn) {
                wr[nn] = x + t;
                wi[nn--] = 0;
            } else {
                y = a(nn - 1, nn - 1);
                w = a(nn, nn - 1) * a(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = std::sqrt(std::fabs(q));
                    x += t;
                    if (q >= 0) {
                        z = p + (p >= 0 ? std::fabs(z) : -std::fabs(z));
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
                        return 0; // no convergence: no solutions (the caller treats the sample as degenerate)
                    if (its == 10 || its == 20) {
                        t += x;
                        for (int i = 0; i <= nn; ++i)
                            a(i, i) -= x;
                        s = std::fabs(a(nn, nn - 1)) + std::fabs(a(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = a(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / a(m + 1, m) + a(m, m + 1);
                        q = a(m + 1, m + 1) - z - r - s;
                        r = a(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s, q /= s, r /= s;
                        if (m == l)
                            break;
                        const double u = std::fabs(a(m, m - 1)) * (std::fabs(q) + std::fabs(r));
                        const double v = std::fabs(p) * (std::fabs(a(m - 1, m - 1)) + std::fabs(z) + std::fabs(a(m + 1, m + 1)));
                        if (u <= eps * v)
                            break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        a(i, i - 2) = 0;
                        if (i != m + 2)
                            a(i, i - 3) = 0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = a(k, k - 1);
                            q = a(k + 1, k - 1);
                            r = (k != nn - 1) ? a(k + 2, k - 1) : 0.0;
                            if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0)
                                p /= x, q /= x, r /= x;
                        }
                        const double sq = std::sqrt(p * p + q * q + r * r);
                        if ((s = (p >= 0 ? sq : -sq)) != 0) {
                            if (k == m) {
                                if (l != m)
                                    a(k, k - 1) = -a(k, k - 1);
                            } else {
                                a(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s, y = q / s, z = r / s;
                            q /= p, r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = a(k, j) + q * a(k + 1, j);
                                if (k != nn - 1) {
                                    p += r * a(k + 2, j);
                                    a(k + 2, j) -= p * z;
                                }
                                a(k + 1, j) -= p * y;
                                a(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * a(i, k) + y * a(i, k + 1);
                                if (k != nn - 1) {
                                    p += z * a(i, k + 2);
                                    a(i, k + 2) -= p * r;
                                }
                                a(i, k + 1) -= p * q;
                                a(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (std::fabs(wi[i]) <= tol * (1.0 + std::fabs(wr[i])))
            out[m++] = wr[i];
    std::sort(out, out + m);
    return m;
}

// null vector of the singular n x n matrix B (row-major, destroyed) by Gaussian elimination with complete pivoting
void null_vector(double *B, int n, double *v) {
    std::vector<int> colperm(n);
    for (int i = 0; i < n; ++i)
        colperm[i] = i;
    auto b = [&](int i, int j) -> double & { return B[i * n + j]; };
    int rank = 0;
    for (int k = 0; k < n - 1; ++k, ++rank) {
        int pr = k, pc = k;
        double best = 0;
        for (int i = k; i < n; ++i)
            for (int j = k; j < n; ++j)
                if (std::fabs(b(i, j)) > best)
                    best = std::fabs(b(i, j)), pr = i, pc = j;
        if (best == 0)
            break;
        for (int j = 0; j < n; ++j)
            std::swap(b(k, j), b(pr, j));
        for (int i = 0; i < n; ++i)
            std::swap(b(i, k), b(i, pc));
        std::swap(colperm[k], colperm[pc]);
        for (int i = k + 1; i < n; ++i) {
            const double f = b(i, k) / b(k, k);
            for (int j = k; j < n; ++j)
                b(i, j) -= f * b(k, j);
        }
    }
    // free variable: the last permuted column; back substitution over the rank x rank upper triangle
    std::vector<double> y(n, 0.0);
    y[n - 1] = 1.0;
    for (int i = n - 2; i >= 0; --i) {
        double s = 0;
        for (int j = i + 1; j < n; ++j)
            s += b(i, j) * y[j];
        y[i] = -s / b(i, i);
    }
    for (int i = 0; i < n; ++i)
        v[colperm[i]] = y[i];
}

} // namespace

// x: four image points relative to the principal point (only the x coordinate of the fourth is used), X: the 3-D points.
// Returns the number of solutions (<= 10): poses and focal lengths, ascending in the eigenvalue x4.
int p35pf(const V2 x_in[4], const V3 X[4], Pose out[10], double focals[10]) {
    const Monomials &M = mono();
    // p35pf.cc:45-58: scale the image points by their mean norm
    double f0 = 0;
    for (int i = 0; i < 4; ++i)
        f0 += std::sqrt(x_in[i].x * x_in[i].x + x_in[i].y * x_in[i].y);
    f0 /= 4;
    V2 x[4];
    for (int i = 0; i < 4; ++i)
        x[i] = V2{x_in[i].x / f0, x_in[i].y / f0};

    // the 7 linear constraints (rows) on the 12 entries of P (row-major), as the columns of a 12 x 7 matrix
    double A[12 * 7];
    std::memset(A, 0, sizeof(A));
    int row = 0;
    for (int i = 0; i < 4; ++i) {
        const double Xh[4] = {X[i].x, X[i].y, X[i].z, 1.0};
        for (int k = 0; k < 4; ++k) {
            A[row * 12 + k] = Xh[k];
            A[row * 12 + 8 + k] = -x[i].x * Xh[k];
        }
        ++row;
        if (i < 3) {
            for (int k = 0; k < 4; ++k) {
                A[row * 12 + 4 + k] = Xh[k];
                A[row * 12 + 8 + k] = -x[i].y * Xh[k];
            }
            ++row;
        }
    }
    double N[12 * 5]; // column-major 12 x 5
    householder_complement(A, 12, 7, N);

    // rows of the left 3 x 3 block as polynomial vectors: a_r[i] = sum_k N(4 r + i, k) x_k + N(4 r + i, 4)
    const int lin[5] = {M.index[1][0][0][0], M.index[0][1][0][0], M.index[0][0][1][0], M.index[0][0][0][1], M.index[0][0][0][0]};
    PVec a[3];
    for (int r = 0; r < 3; ++r)
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 5; ++k)
                a[r].v[i].c[lin[k]] = N[k * 12 + 4 * r + i];

    Poly eq[29];
    int ne = 0;
    const Poly quads[4] = {dot(a[0], a[1]), dot(a[0], a[2]), dot(a[1], a[2]), sub(dot(a[0], a[0]), dot(a[1], a[1]))};
    Poly shift[5];
    for (int k = 0; k < 5; ++k)
        shift[k].c[lin[k == 4 ? 4 : k]] = 1.0;
    for (int q = 0; q < 4; ++q) {
        eq[ne++] = quads[q];
        for (int k = 0; k < 4; ++k)
            eq[ne++] = mul(quads[q], shift[k]);
    }
    const PVec c23 = cross(a[1], a[2]), c31 = cross(a[2], a[0]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            eq[ne++] = sub(mul(c23.v[i], a[1].v[j]), mul(c31.v[j], a[0].v[i]));

    // 29 x 35, rows scaled to unit maximum; Gauss-Jordan over the 25 eliminated monomials, pivot = largest remaining entry of
    // the column among the rows not used yet
    double C[29][NM];
    for (int r = 0; r < 29; ++r) {
        double mx = 0;
        for (int c = 0; c < NM; ++c)
            mx = std::max(mx, std::fabs(eq[r].c[c]));
        for (int c = 0; c < NM; ++c)
            C[r][c] = mx > 0 ? eq[r].c[c] / mx : 0.0;
    }
    bool used[29] = {false};
    int pivot_row[25];
    for (int k = 0; k < 25; ++k) {
        const int col = M.elim[k];
        int pr = -1;
        double best = 0;
        for (int r = 0; r < 29; ++r)
            if (!used[r] && std::fabs(C[r][col]) > best)
                best = std::fabs(C[r][col]), pr = r;
        if (pr < 0 || best < 1e-13)
            return 0; // degenerate sample
        used[pr] = true;
        pivot_row[k] = pr;
        const double inv = 1.0 / C[pr][col];
        for (int c = 0; c < NM; ++c)
            C[pr][c] *= inv;
        for (int r = 0; r < 29; ++r)
            if (r != pr && C[r][col] != 0) {
                const double f = C[r][col];
                for (int c = 0; c < NM; ++c)
                    C[r][c] -= f * C[pr][c];
            }
    }
    // action matrix of x4 on the standard monomials: x4 b is a standard monomial itself or an eliminated one
    double AM[100];
    std::memset(AM, 0, sizeof(AM));
    const int x4[4] = {0, 0, 0, 1};
    for (int k = 0; k < 10; ++k) {
        const int m = M.of(M.e[M.basis[k]], x4);
        bool done = false;
        for (int j = 0; j < 10 && !done; ++j)
            if (M.basis[j] == m) {
                AM[k * 10 + j] = 1.0;
                done = true;
            }
        for (int e = 0; e < 25 && !done; ++e)
            if (M.elim[e] == m) {
                for (int j = 0; j < 10; ++j)
                    AM[k * 10 + j] = -C[pivot_row[e]][M.basis[j]];
                done = true;
            }
    }
    double work[100], ev[10];
    std::memcpy(work, AM, sizeof(AM));
    const int nroots = real_eigenvalues(work, 10, ev, 1e-8);
    int n = 0;
    for (int s = 0; s < nroots; ++s) {
        double B[100], v[10];
        for (int i = 0; i < 100; ++i)
            B[i] = AM[i];
        for (int i = 0; i < 10; ++i)
            B[i * 10 + i] -= ev[s];
        null_vector(B, 10, v);
        if (v[9] == 0)
            continue;
        const double al[5] = {v[5] / v[9], v[6] / v[9], v[7] / v[9], v[8] / v[9], 1.0};
        double P[12];
        for (int i = 0; i < 12; ++i) {
            double sum = 0;
            for (int k = 0; k < 5; ++k)
                sum += N[k * 12 + i] * al[k];
            P[i] = sum;
        }
        M3 R;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                R.m[r][c] = P[4 * r + c];
        V3 t{P[3], P[7], P[11]};
        const double det = R.m[0][0] * (R.m[1][1] * R.m[2][2] - R.m[1][2] * R.m[2][1]) -
                           R.m[0][1] * (R.m[1][0] * R.m[2][2] - R.m[1][2] * R.m[2][0]) +
                           R.m[0][2] * (R.m[1][0] * R.m[2][1] - R.m[1][1] * R.m[2][0]);
        const double sgn = det < 0 ? -1.0 : 1.0;
        const double n3 = std::sqrt(R.m[2][0] * R.m[2][0] + R.m[2][1] * R.m[2][1] + R.m[2][2] * R.m[2][2]);
        if (!(n3 > 0))
            continue;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                R.m[r][c] *= sgn / n3;
            t[r] *= sgn / n3;
        }
        const double n1 = std::sqrt(R.m[0][0] * R.m[0][0] + R.m[0][1] * R.m[0][1] + R.m[0][2] * R.m[0][2]);
        const double n2 = std::sqrt(R.m[1][0] * R.m[1][0] + R.m[1][1] * R.m[1][1] + R.m[1][2] * R.m[1][2]);
        const double focal = 0.5 * (n1 + n2);
        for (int c = 0; c < 3; ++c) {
            R.m[0][c] /= focal;
            R.m[1][c] /= focal;
        }
        t.x /= focal;
        t.y /= focal;
        out[n] = Pose(R, t);
        focals[n] = focal * f0;
        ++n;
    }
    return n;
}


// =====================================================================================================================
// Relative pose with one unknown focal length shared by both cameras, from six correspondences - the problem the reference solves
// in solvers/relpose_6pt_focal.cc with a generated 31 x 46 elimination template and a Sturm chain of degree 15.  Restated FROM
// FIRST PRINCIPLES as a polynomial eigenvalue problem (hidden variable w = 1 / f^2); nothing of the template is used.  Taken from
// the reference: the interface and its conventions (relpose_6pt_focal.cc:1083-1144: unit bearings, null space of the six epipolar
// constraints from a full-pivoting Householder QR, F = N0 + x N1 + y N2, solutions with w < 1e-8 dropped, focal = sqrt(1 / w),
// E = K F K, motion_from_essential on the bearings at that focal length) and the ORDER of the solutions (ascending in y: the
// reference's action variable, found by Sturm bisection from left to right, :1071-1078).
//
// Derivation.  With Q = diag(1, 1, w) the fundamental matrix of two cameras K = diag(f, f, 1) satisfies
//     det F = 0,        2 (F Q F^T) Q F - trace(F Q F^T Q) F = 0                          (the trace constraint of E = K F K)
// - ten equations, cubic in (x, y), the nine of the trace constraint quadratic in w: (C0 + w C1 + w^2 C2) m = 0 with m the ten
// monomials of degree <= 3 in (x, y).  Every entry of the w^2 part carries the factor F33 (a linear form), so C2 has rank <= 6:
// Gaussian elimination on C2 leaves six equations of degree 2 in w, three of degree 1 and the determinant of degree 0 - the row
// degrees add up to 15, the number of solutions, so with L the matrix of the leading row coefficients and u = L m
//     w^2 u_i + w (a1_i . u) + a0_i . u = 0  (i < 6),     w u_j + b0_j . u = 0  (j = 6..8),     u_9 = 0
// is a 15 x 15 standard eigenvalue problem in the state (u_0..u_8, w u_0..w u_5).  Its real eigenvalues w >= 1e-8 are the
// solutions; (x, y) come from the null vector of C0 + w C1 + w^2 C2.  scripts/exp/sixpt_focal_polyeig.py is the numpy experiment.
namespace {

constexpr int kIdx6[4][4] = {{9, 8, 6, 3}, {7, 5, 2, -1}, {4, 1, -1, -1}, {0, -1, -1, -1}}; // x^a y^b -> column (graded)
constexpr int kExp6[10][2] = {{3, 0}, {2, 1}, {1, 2}, {0, 3}, {2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};

struct P6 {
    double c[10];
    P6() { std::memset(c, 0, sizeof(c)); }
};
P6 mul6(const P6 &p, const P6 &q) {
    P6 r;
    for (int i = 0; i < 10; ++i)
        if (p.c[i] != 0)
            for (int j = 0; j < 10; ++j)
                if (q.c[j] != 0)
                    r.c[kIdx6[kExp6[i][0] + kExp6[j][0]][kExp6[i][1] + kExp6[j][1]]] += p.c[i] * q.c[j];
    return r;
}
P6 add6(const P6 &p, const P6 &q) {
    P6 r;
    for (int i = 0; i < 10; ++i)
        r.c[i] = p.c[i] + q.c[i];
    return r;
}
P6 sub6(const P6 &p, const P6 &q) {
    P6 r;
    for (int i = 0; i < 10; ++i)
        r.c[i] = p.c[i] - q.c[i];
    return r;
}
P6 twice_minus(const P6 &p, const P6 &q) { // 2 p - q
    P6 r;
    for (int i = 0; i < 10; ++i)
        r.c[i] = 2.0 * p.c[i] - q.c[i];
    return r;
}

} // namespace

// The polynomial eigenvalue problem of a sample: C[k] (10 x 10, row-major), k = power of w; row 0 = det F, rows 1 + 3 i + j = entry
// (i, j) of the trace constraint; every row scaled to unit maximum over the three matrices.  nb: 9 x 3 null-space basis.
static void sixpt_equations(const double *nb, double C[3][100]) {
    P6 F[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int e = 3 * j + i; // column-major F
            F[i][j].c[9] = nb[0 * 9 + e];
            F[i][j].c[7] = nb[1 * 9 + e];
            F[i][j].c[8] = nb[2 * 9 + e];
        }
    P6 G0[3][3], G1[3][3]; // F Q F^T = G0 + w G1
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            G0[i][j] = add6(mul6(F[i][0], F[j][0]), mul6(F[i][1], F[j][1]));
            G1[i][j] = mul6(F[i][2], F[j][2]);
        }
    const P6 tr0 = add6(G0[0][0], G0[1][1]);
    const P6 tr1 = add6(add6(G1[0][0], G1[1][1]), G0[2][2]);
    const P6 tr2 = G1[2][2];
    P6 eq[10][3];
    eq[0][0] = add6(sub6(mul6(F[0][0], sub6(mul6(F[1][1], F[2][2]), mul6(F[1][2], F[2][1]))),
                         mul6(F[0][1], sub6(mul6(F[1][0], F[2][2]), mul6(F[1][2], F[2][0])))),
                    mul6(F[0][2], sub6(mul6(F[1][0], F[2][1]), mul6(F[1][1], F[2][0]))));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const P6 p0 = add6(mul6(G0[i][0], F[0][j]), mul6(G0[i][1], F[1][j]));
            const P6 p1 = add6(add6(mul6(G1[i][0], F[0][j]), mul6(G1[i][1], F[1][j])), mul6(G0[i][2], F[2][j]));
            const P6 p2 = mul6(G1[i][2], F[2][j]);
            P6 *e = eq[1 + 3 * i + j];
            e[0] = twice_minus(p0, mul6(tr0, F[i][j]));
            e[1] = twice_minus(p1, mul6(tr1, F[i][j]));
            e[2] = twice_minus(p2, mul6(tr2, F[i][j]));
        }
    for (int r = 0; r < 10; ++r) {
        double mx = 0;
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 10; ++c)
                mx = std::fmax(mx, std::fabs(eq[r][k].c[c]));
        const double s = mx > 0 ? 1.0 / mx : 0.0;
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 10; ++c)
                C[k][r * 10 + c] = eq[r][k].c[c] * s;
    }
}

// Parlett-Reinsch balancing (EISPACK balanc without the permutation step): similarity scaling by powers of two until the row and
// column 1-norms of every index are within a factor of two - exact in floating point, eigenvalues unchanged
static void balance_pow2(double *a, int n) {
    bool done = false;
    for (int sweep = 0; sweep < 64 && !done; ++sweep) { // (typically 3 - 6 sweeps; the cap bounds the loop on overflowing input)
        done = true;
        for (int i = 0; i < n; ++i) {
            double c = 0, r = 0;
            for (int j = 0; j < n; ++j)
                if (j != i) {
                    c += std::fabs(a[j * n + i]);
                    r += std::fabs(a[i * n + j]);
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

// 15 x 15 companion matrix T (row-major) of the row-reduced problem; false for a degenerate sample.  C is destroyed.
static bool sixpt_companion(double C[3][100], double *T) {
    auto c = [&](int k, int r, int col) -> double & { return C[k][r * 10 + col]; };
    // rows 1..9: six steps of Gaussian elimination with complete pivoting on C2, the row operations applied to C0 and C1 as well
    for (int k = 0; k < 6; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = 1 + k; r < 10; ++r)
            for (int col = 0; col < 10; ++col)
                if (std::fabs(c(2, r, col)) > best)
                    best = std::fabs(c(2, r, col)), pr = r, pc = col;
        if (pr < 0)
            return false;
        if (pr != 1 + k)
            for (int m = 0; m < 3; ++m)
                for (int col = 0; col < 10; ++col)
                    std::swap(c(m, 1 + k, col), c(m, pr, col));
        for (int r = 2 + k; r < 10; ++r) {
            const double f = c(2, r, pc) / c(2, 1 + k, pc);
            if (f == 0)
                continue;
            for (int m = 0; m < 3; ++m)
                for (int col = 0; col < 10; ++col)
                    c(m, r, col) -= f * c(m, 1 + k, col);
            c(2, r, pc) = 0;
        }
    }
    // L: leading row coefficients, rows (C2[1..6], C1[7..9], C0[0]); X L = R for the 15 right-hand rows R = (C0[1..6], C1[1..6],
    // C0[7..9]) by LU with partial pivoting on L^T
    double A[10][10], B[10][15]; // A = L^T, B = R^T
    for (int col = 0; col < 10; ++col) {
        for (int i = 0; i < 6; ++i)
            A[col][i] = c(2, 1 + i, col);
        for (int j = 0; j < 3; ++j)
            A[col][6 + j] = c(1, 7 + j, col);
        A[col][9] = c(0, 0, col);
        for (int i = 0; i < 6; ++i) {
            B[col][i] = c(0, 1 + i, col);
            B[col][6 + i] = c(1, 1 + i, col);
        }
        for (int j = 0; j < 3; ++j)
            B[col][12 + j] = c(0, 7 + j, col);
    }
    for (int k = 0; k < 10; ++k) {
        int pr = k;
        double best = std::fabs(A[k][k]);
        for (int r = k + 1; r < 10; ++r)
            if (std::fabs(A[r][k]) > best)
                best = std::fabs(A[r][k]), pr = r;
        if (best == 0)
            return false;
        if (pr != k) {
            for (int col = 0; col < 10; ++col)
                std::swap(A[k][col], A[pr][col]);
            for (int col = 0; col < 15; ++col)
                std::swap(B[k][col], B[pr][col]);
        }
        for (int r = k + 1; r < 10; ++r) {
            const double f = A[r][k] / A[k][k];
            if (f == 0)
                continue;
            for (int col = k + 1; col < 10; ++col)
                A[r][col] -= f * A[k][col];
            for (int col = 0; col < 15; ++col)
                B[r][col] -= f * B[k][col];
        }
    }
    for (int col = 0; col < 15; ++col) // back substitution: B[.][col] becomes the row `col` of R L^-1
        for (int r = 9; r >= 0; --r) {
            double s = B[r][col];
            for (int m = r + 1; m < 10; ++m)
                s -= A[r][m] * B[m][col];
            B[r][col] = s / A[r][r];
        }
    // a0_i = B[.][i], a1_i = B[.][6 + i] (i < 6), b0_j = B[.][12 + j] (j < 3); state z = (u_0..u_8, w u_0..w u_5)
    std::memset(T, 0, sizeof(double) * 225);
    for (int i = 0; i < 6; ++i)
        T[i * 15 + 9 + i] = 1.0;
    for (int j = 0; j < 3; ++j)
        for (int m = 0; m < 9; ++m)
            T[(6 + j) * 15 + m] = -B[m][12 + j];
    for (int i = 0; i < 6; ++i) {
        double *row = T + (9 + i) * 15;
        for (int m = 0; m < 6; ++m)
            row[9 + m] = -B[m][6 + i];
        for (int m = 0; m < 9; ++m) {
            double s = -B[m][i];
            for (int j = 0; j < 3; ++j)
                s += B[6 + j][6 + i] * B[m][12 + j];
            row[m] = s;
        }
    }
    return true;
}

// Interface of solvers/relpose_6pt_focal.h:12-13.  At most 15 solutions x 4 poses.
int relpose_6pt_shared_focal(const V3 x1[6], const V3 x2[6], Pose out[60], double focals[60]) {
    double A[54];
    for (int i = 0; i < 6; ++i) // relpose_6pt_focal.cc:1087-1090: column i = (x1_0 x2, x1_1 x2, x1_2 x2)
        for (int j = 0; j < 3; ++j) {
            A[i * 9 + 3 * j + 0] = x1[i][j] * x2[i].x;
            A[i * 9 + 3 * j + 1] = x1[i][j] * x2[i].y;
            A[i * 9 + 3 * j + 2] = x1[i][j] * x2[i].z;
        }
    double nb[27]; // 9 x 3, column-major
    householder_complement(A, 9, 6, nb);
    double C[3][100], Cw[3][100], T[225], ev[15];
    sixpt_equations(nb, C);
    std::memcpy(Cw, C, sizeof(C));
    if (!sixpt_companion(Cw, T))
        return 0;
    for (int e = 0; e < 225; ++e)
        if (!std::isfinite(T[e]))
            return 0; // (a vanishing pivot: the balancing below would not terminate on an infinite entry)
    balance_pow2(T, 15);
    const int nroots = real_eigenvalues(T, 15, ev, 1e-8);
    struct Sol {
        double x, y, w;
    } sols[15];
    int ns = 0;
    for (int s = 0; s < nroots; ++s) {
        const double w = ev[s];
        if (w < 1e-8) // relpose_6pt_focal.cc:1106
            continue;
        double M[100], v[10];
        for (int i = 0; i < 100; ++i)
            M[i] = C[0][i] + w * (C[1][i] + w * C[2][i]);
        null_vector(M, 10, v);
        if (v[9] == 0)
            continue;
        sols[ns++] = Sol{v[7] / v[9], v[8] / v[9], w};
    }
    std::stable_sort(sols, sols + ns, [](const Sol &a, const Sol &b) { return a.y < b.y; });
    int n = 0;
    for (int s = 0; s < ns; ++s) {
        const double focal = std::sqrt(1.0 / sols[s].w);
        double Fv[9], nrm = 0;
        for (int e = 0; e < 9; ++e) {
            Fv[e] = nb[e] + sols[s].x * nb[9 + e] + sols[s].y * nb[18 + e];
            nrm += Fv[e] * Fv[e];
        }
        nrm = std::sqrt(nrm);
        M3 E;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double ki = i < 2 ? focal : 1.0, kj = j < 2 ? focal : 1.0;
                E.m[i][j] = ki * ((Fv[3 * j + i] / nrm) * kj); // K * (F * K), relpose_6pt_focal.cc:1122
            }
        V3 u1[6], u2[6];
        for (int i = 0; i < 6; ++i) {
            u1[i] = normalized(V3{x1[i].x / focal, x1[i].y / focal, x1[i].z});
            u2[i] = normalized(V3{x2[i].x / focal, x2[i].y / focal, x2[i].z});
        }
        const int m = motion_from_essential(E, u1, u2, 6, out + n);
        for (int i = 0; i < m; ++i)
            focals[n + i] = focal;
        n += m;
    }
    return n;
}

} // namespace orc
