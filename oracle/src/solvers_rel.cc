// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the two-view minimal solvers and their helpers.
//   Sturm bracketing      : PoseLib/misc/sturm.h:47-84 (sequence), :98-112 (sign changes), :144-150 (Cauchy bound),
//                           :153-208 (Ridders + Newton), :210-231 (isolation), :233-274 (driver)
//   essential helpers     : PoseLib/misc/essential.cc:35-38, :40-57, :103-169
//   5-point (Nister)      : PoseLib/solvers/relpose_5pt.cc:101-157 (constraints), :159-395 (E), :397-409 (poses)
//   7-point               : PoseLib/solvers/relpose_7pt.cc:10-60
//   4-point homography    : PoseLib/solvers/homography_4pt.cc:36-128  (closed-form ACA method, not DLT)
// Eigen decompositions used by the reference (fullPivHouseholderQr / partialPivLu /
// colPivHouseholderQr) are restated from the published algorithms; their results agree with
// Eigen's at rounding level only ("parity unpinned" there, see oracle/README.md).
#include "solvers.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

// =============================================================================== Sturm
namespace {

struct SturmChain {
    int N;
    double f[17];      // monic polynomial, f[N] == 1
    double fp[16];     // derivative / N (monic, degree N-1)
    double q0[16], q1[16], c[16];
    double tail0, tail1, last;
};

inline double horner_monic(const double *p, int deg, double x) {
    double v = x + p[deg - 1];
    for (int i = deg - 2; i >= 0; --i)
        v = x * v + p[i];
    return v;
}

void build_chain(SturmChain &S) { // sturm.h:47-84
    const int N = S.N;
    double bufA[17] = {0}, bufB[17] = {0}, bufC[17] = {0};
    double *hi = bufA, *lo = bufB, *rem = bufC;
    for (int i = 0; i <= N; ++i)
        hi[i] = S.f[i];
    for (int i = 0; i < N; ++i)
        lo[i] = S.fp[i];
    for (int i = 0; i < N - 1; ++i) {
        const int dh = N - i;     // degree of hi
        const int dl = N - 1 - i; // degree of lo
        const double a1 = hi[dh] * lo[dl];
        const double a0 = hi[dh - 1] * lo[dl] - hi[dh] * lo[dl - 1];
        rem[0] = hi[0] - a0 * lo[0];
        for (int j = 1; j < dl; ++j)
            rem[j] = hi[j] - a1 * lo[j - 1] - a0 * lo[j];
        const double scale = -std::abs(rem[dl - 1]);
        const double inv = 1.0 / scale;
        for (int j = 0; j < dl; ++j)
            rem[j] = rem[j] * inv;
        S.q0[i] = a0;
        S.q1[i] = a1;
        S.c[i] = scale;
        double *t = hi;
        hi = lo;
        lo = rem;
        rem = t;
    }
    S.tail0 = hi[0];
    S.tail1 = hi[1];
    S.last = lo[0];
}

int sign_variations(const SturmChain &S, double x) { // sturm.h:98-112
    const int N = S.N;
    double v[18];
    v[N] = S.last;
    v[N - 1] = S.tail0 + x * S.tail1;
    for (int i = N - 2; i >= 0; --i)
        v[i] = (S.q0[i] + x * S.q1[i]) * v[i + 1] + S.c[i] * v[i + 2];
    int count = 0;
    for (int i = 0; i < N; ++i)
        if ((v[i] < 0) != (v[i + 1] < 0))
            ++count;
    return count;
}

void ridders_then_newton(const SturmChain &S, double a, double b, double *roots, int &n, double tol) { // :153-208
    const int N = S.N;
    double fa = horner_monic(S.f, N, a);
    double fb = horner_monic(S.f, N, b);
    if (!((fa < 0) ^ (fb < 0)))
        return;
    for (int it = 0; it < 30; ++it) {
        if (std::abs(a - b) < 1e-3)
            break;
        const double c = (a + b) * 0.5;
        const double fc = horner_monic(S.f, N, c);
        const double s = std::sqrt(fc * fc - fa * fb);
        if (!s)
            break;
        const double d = (fa < fb) ? c + (a - c) * fc / s : c + (c - a) * fc / s;
        const double fd = horner_monic(S.f, N, d);
        if (fd >= 0 ? (fc < 0) : (fc > 0)) {
            a = c;
            fa = fc;
            b = d;
            fb = fd;
        } else if (fd >= 0 ? (fa < 0) : (fa > 0)) {
            b = d;
            fb = fd;
        } else {
            a = d;
            fa = fd;
        }
    }
    double x = (a + b) * 0.5;
    for (int it = 0; it < 10; ++it) {
        const double fx = horner_monic(S.f, N, x);
        if (std::abs(fx) < tol)
            break;
        const double fpx = static_cast<double>(N) * horner_monic(S.fp, N - 1, x);
        const double dx = fx / fpx;
        x = x - dx;
        if (std::abs(dx) < tol)
            break;
    }
    roots[n++] = x;
}

} // namespace

int sturm_real_roots(const double *coef, int N, double *roots, double tol) { // sturm.h:233-274
    if (coef[N] == 0.0)
        return 0;
    SturmChain S;
    S.N = N;
    const double lead_inv = 1.0 / coef[N];
    for (int i = 0; i < N; ++i)
        S.f[i] = coef[i] * lead_inv;
    S.f[N] = 1.0;
    for (int i = 0; i < N - 1; ++i)
        S.fp[i] = S.f[i + 1] * ((i + 1) / static_cast<double>(N));
    S.fp[N - 1] = 1.0;
    build_chain(S);

    double bound = 0;
    for (int i = 0; i < N; ++i)
        bound = std::max(bound, std::abs(S.f[i]));
    bound = 1.0 + bound; // Cauchy, sturm.h:144-150

    const int sa = sign_variations(S, -bound), sb = sign_variations(S, bound);
    if (sa - sb == 0)
        return 0;

    // Depth-first, left interval first — same visiting order as the recursive reference (:210-231).
    struct Item {
        double a, b;
        int sa, sb, depth;
    };
    std::vector<Item> stack;
    stack.push_back({-bound, bound, sa, sb, 0});
    int n = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.depth > 300)
            continue;
        if (it.b - it.a < tol) {
            roots[n++] = it.b;
            continue;
        }
        const int k = it.sa - it.sb;
        if (k > 1) {
            const double mid = (it.a + it.b) * 0.5;
            const int sm = sign_variations(S, mid);
            stack.push_back({mid, it.b, sm, it.sb, it.depth + 1});
            stack.push_back({it.a, mid, it.sa, sm, it.depth + 1});
        } else if (k == 1) {
            ridders_then_newton(S, it.a, it.b, roots, n, tol);
        }
    }
    return n;
}

int sturm_real_roots_deg10(const double c[11], double roots[10], double tol) {
    return sturm_real_roots(c, 10, roots, tol);
}

// =============================================================================== essential helpers
M3 essential_from_motion(const Pose &p) { // essential.cc:35-38   E = [t]x R
    M3 Tx;
    Tx.m[0][1] = -p.t.z;
    Tx.m[0][2] = p.t.y;
    Tx.m[1][0] = p.t.z;
    Tx.m[1][2] = -p.t.x;
    Tx.m[2][0] = -p.t.y;
    Tx.m[2][1] = p.t.x;
    return Tx * p.R();
}

bool check_cheirality(const Pose &p, const V3 &x1, const V3 &x2, double min_depth) { // essential.cc:40-57
    const V3 Rx1 = p.rotate(x1);
    const double a = -dot(Rx1, x2);
    const double b1 = -dot(Rx1, p.t);
    const double b2 = dot(x2, p.t);
    const double l1 = b1 - a * b2;
    const double l2 = -a * b1 + b2;
    min_depth = min_depth * (1 - a * a);
    return l1 > min_depth && l2 > min_depth;
}

static bool cheirality_all(const Pose &p, const V3 *x1, const V3 *x2, int n) { // essential.cc:81-89
    for (int i = 0; i < n; ++i)
        if (!check_cheirality(p, x1[i], x2[i], 0.0))
            return false;
    return true;
}

int motion_from_essential(const M3 &E, const V3 *x1, const V3 *x2, int npts, Pose *out) { // essential.cc:103-169
    const V3 e0 = E.col(0), e1 = E.col(1), e2 = E.col(2);
    const V3 u12 = cross(e0, e1), u13 = cross(e0, e2), u23 = cross(e1, e2);
    const double n12 = sqnorm(u12), n13 = sqnorm(u13), n23 = sqnorm(u23);
    V3 c1, c2; // columns 1 and 2 of U*W
    if (n12 > n13) {
        if (n12 > n23) {
            c1 = normalized(e0);
            c2 = u12 / std::sqrt(n12);
        } else {
            c1 = normalized(e1);
            c2 = u23 / std::sqrt(n23);
        }
    } else {
        if (n13 > n23) {
            c1 = normalized(e0);
            c2 = u13 / std::sqrt(n13);
        } else {
            c1 = normalized(e1);
            c2 = u23 / std::sqrt(n23);
        }
    }
    V3 c0 = -cross(c2, c1);

    const M3 Et = transpose(E);
    V3 r0 = Et * c1;          // c1^T E
    V3 r1 = Et * (-c0);       // -c0^T E
    r0 = normalized(r0);
    r1 = r1 - dot(r0, r1) * r0;
    r1 = normalized(r1);
    const V3 r2 = cross(r0, r1);
    M3 Vt;
    Vt.set_row(0, r0);
    Vt.set_row(1, r1);
    Vt.set_row(2, r2);

    M3 UW;
    UW.set_col(0, c0);
    UW.set_col(1, c1);
    UW.set_col(2, c2);

    int n = 0;
    Pose p;
    p.q = rotmat_to_quat(UW * Vt);
    p.t = c2;
    if (cheirality_all(p, x1, x2, npts))
        out[n++] = p;
    p.t = -p.t;
    if (cheirality_all(p, x1, x2, npts))
        out[n++] = p;
    UW.set_col(0, -c0);
    UW.set_col(1, -c1);
    p.q = rotmat_to_quat(UW * Vt);
    if (cheirality_all(p, x1, x2, npts))
        out[n++] = p;
    p.t = -p.t;
    if (cheirality_all(p, x1, x2, npts))
        out[n++] = p;
    return n;
}

// =============================================================================== complement basis
void householder_complement(const double *Ain, int rows, int cols, double *basis) {
    // Full-pivoting Householder QR followed by accumulation of Q, in the order Eigen uses
    // (pivot = largest |a_ij| of the trailing corner, column-major scan, first maximum wins).
    std::vector<double> qr(Ain, Ain + rows * cols);
    auto a = [&](int r, int c) -> double & { return qr[c * rows + r]; };
    std::vector<double> tau(cols, 0.0);
    std::vector<int> rowswap(cols);
    double biggest = 0;
    const double precision = std::numeric_limits<double>::epsilon() * cols;
    int rank = cols;
    for (int k = 0; k < cols; ++k) {
        int pr = k, pc = k;
        double best = std::abs(a(k, k));
        for (int c = k; c < cols; ++c)
            for (int r = k; r < rows; ++r)
                if (std::abs(a(r, c)) > best) {
                    best = std::abs(a(r, c));
                    pr = r;
                    pc = c;
                }
        if (k == 0)
            biggest = best;
        if (best <= biggest * precision) {
            rank = k;
            for (int i = k; i < cols; ++i) {
                rowswap[i] = i;
                tau[i] = 0;
            }
            break;
        }
        rowswap[k] = pr;
        if (pr != k)
            for (int c = k; c < cols; ++c)
                std::swap(a(k, c), a(pr, c));
        if (pc != k)
            for (int r = 0; r < rows; ++r)
                std::swap(a(r, k), a(r, pc));
        // reflector for column k
        double tail_sq = 0;
        for (int r = k + 1; r < rows; ++r)
            tail_sq += a(r, k) * a(r, k);
        const double c0 = a(k, k);
        double beta;
        if (tail_sq <= std::numeric_limits<double>::min()) {
            tau[k] = 0;
            beta = c0;
            for (int r = k + 1; r < rows; ++r)
                a(r, k) = 0;
        } else {
            beta = std::sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0)
                beta = -beta;
            for (int r = k + 1; r < rows; ++r)
                a(r, k) = a(r, k) / (c0 - beta);
            tau[k] = (beta - c0) / beta;
        }
        a(k, k) = beta;
        if (tau[k] != 0)
            for (int c = k + 1; c < cols; ++c) {
                double t = 0;
                for (int r = k + 1; r < rows; ++r)
                    t += a(r, k) * a(r, c);
                t += a(k, c);
                a(k, c) -= tau[k] * t;
                for (int r = k + 1; r < rows; ++r)
                    a(r, c) -= tau[k] * a(r, k) * t;
            }
    }
    (void)rank;
    std::vector<double> Q(rows * rows, 0.0);
    auto q = [&](int r, int c) -> double & { return Q[c * rows + r]; };
    for (int i = 0; i < rows; ++i)
        q(i, i) = 1.0;
    for (int k = cols - 1; k >= 0; --k) {
        if (tau[k] != 0)
            for (int c = k; c < rows; ++c) {
                double t = 0;
                for (int r = k + 1; r < rows; ++r)
                    t += a(r, k) * q(r, c);
                t += q(k, c);
                q(k, c) -= tau[k] * t;
                for (int r = k + 1; r < rows; ++r)
                    q(r, c) -= tau[k] * a(r, k) * t;
            }
        if (rowswap[k] != k)
            for (int c = 0; c < rows; ++c)
                std::swap(q(k, c), q(rowswap[k], c));
    }
    const int nb = rows - cols;
    for (int j = 0; j < nb; ++j)
        for (int r = 0; r < rows; ++r)
            basis[j * rows + r] = q(r, cols + j);
}

// =============================================================================== 5-point
namespace {

// Polynomials in (x,y,z) of total degree <= 3, stored densely by exponent triple.
struct Poly {
    double c[4][4][4];
    Poly() { std::memset(c, 0, sizeof(c)); }
};
Poly linear(const double l[4]) { // l = coefficients of [x, y, z, 1]
    Poly p;
    p.c[1][0][0] = l[0];
    p.c[0][1][0] = l[1];
    p.c[0][0][1] = l[2];
    p.c[0][0][0] = l[3];
    return p;
}
Poly mul(const Poly &a, int da, const Poly &b, int db) {
    Poly r;
    for (int i = 0; i <= da; ++i)
        for (int j = 0; i + j <= da; ++j)
            for (int k = 0; i + j + k <= da; ++k) {
                const double av = a.c[i][j][k];
                if (av == 0)
                    continue;
                for (int l = 0; l <= db; ++l)
                    for (int m = 0; l + m <= db; ++m)
                        for (int n = 0; l + m + n <= db; ++n)
                            r.c[i + l][j + m][k + n] += av * b.c[l][m][n];
            }
    return r;
}
void axpy(Poly &y, double s, const Poly &x) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k)
                y.c[i][j][k] += s * x.c[i][j][k];
}
// Nister's ordering of the 20 cubic monomials (relpose_5pt.cc:57-58)
const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                          {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                          {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
void to_row(const Poly &p, double row[20]) {
    for (int m = 0; m < 20; ++m)
        row[m] = p.c[kMono[m][0]][kMono[m][1]][kMono[m][2]];
}

// small univariate polynomial helper for the final 3x3 polynomial determinant
struct UPoly {
    double c[11];
    int deg;
};
UPoly umul(const UPoly &a, const UPoly &b) {
    UPoly r;
    r.deg = a.deg + b.deg;
    for (int i = 0; i <= r.deg; ++i)
        r.c[i] = 0;
    for (int i = 0; i <= a.deg; ++i)
        for (int j = 0; j <= b.deg; ++j)
            r.c[i + j] += a.c[i] * b.c[j];
    return r;
}

} // namespace

int essential_5pt(const V3 x1[5], const V3 x2[5], M3 Eout[10]) { // relpose_5pt.cc:159-395
    // 9x5 epipolar system: column i = kron(x1_i, x2_i)
    double A[45];
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i * 9 + 3 * j + 0] = x1[i][j] * x2[i].x;
            A[i * 9 + 3 * j + 1] = x1[i][j] * x2[i].y;
            A[i * 9 + 3 * j + 2] = x1[i][j] * x2[i].z;
        }
    double nb[36]; // 9 x 4 column-major: nb[b*9 + e]
    householder_complement(A, 9, 5, nb);

    // E(i,j) as a linear form in (x,y,z,1); vectorised index e = 3*j + i (column-major E)
    Poly Ep[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int e = 3 * j + i;
            const double l[4] = {nb[0 * 9 + e], nb[1 * 9 + e], nb[2 * 9 + e], nb[3 * 9 + e]};
            Ep[i][j] = linear(l);
        }

    double M[10][20];
    // rows 0..8 : (E E^T - 1/2 trace(E E^T) I) E ;  row 9 : det(E)   (relpose_5pt.cc:101-157)
    Poly EEt[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            Poly s = mul(Ep[i][0], 1, Ep[j][0], 1);
            axpy(s, 1.0, mul(Ep[i][1], 1, Ep[j][1], 1));
            axpy(s, 1.0, mul(Ep[i][2], 1, Ep[j][2], 1));
            EEt[i][j] = s;
            EEt[j][i] = s;
        }
    Poly half_trace;
    axpy(half_trace, 0.5, EEt[0][0]);
    axpy(half_trace, 0.5, EEt[1][1]);
    axpy(half_trace, 0.5, EEt[2][2]);
    for (int i = 0; i < 3; ++i)
        axpy(EEt[i][i], -1.0, half_trace);
    int r = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Poly s = mul(EEt[i][0], 2, Ep[0][j], 1);
            axpy(s, 1.0, mul(EEt[i][1], 2, Ep[1][j], 1));
            axpy(s, 1.0, mul(EEt[i][2], 2, Ep[2][j], 1));
            to_row(s, M[r++]);
        }
    {
        Poly m0 = mul(Ep[0][1], 1, Ep[1][2], 1);
        axpy(m0, -1.0, mul(Ep[0][2], 1, Ep[1][1], 1));
        Poly m1 = mul(Ep[0][2], 1, Ep[1][0], 1);
        axpy(m1, -1.0, mul(Ep[0][0], 1, Ep[1][2], 1));
        Poly m2 = mul(Ep[0][0], 1, Ep[1][1], 1);
        axpy(m2, -1.0, mul(Ep[0][1], 1, Ep[1][0], 1));
        Poly d = mul(m0, 2, Ep[2][0], 1);
        axpy(d, 1.0, mul(m1, 2, Ep[2][1], 1));
        axpy(d, 1.0, mul(m2, 2, Ep[2][2], 1));
        to_row(d, M[9]);
    }

    // Gauss-Jordan of the first 10 columns: X = M[:, :10]^{-1} M[:, 10:]  (partial-pivot LU, :173)
    double L[10][10], X[10][10];
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) {
            L[i][j] = M[i][j];
            X[i][j] = M[i][10 + j];
        }
    for (int k = 0; k < 10; ++k) {
        int piv = k;
        double best = std::abs(L[k][k]);
        for (int i = k + 1; i < 10; ++i)
            if (std::abs(L[i][k]) > best) {
                best = std::abs(L[i][k]);
                piv = i;
            }
        if (piv != k)
            for (int j = 0; j < 10; ++j) {
                std::swap(L[k][j], L[piv][j]);
                std::swap(X[k][j], X[piv][j]);
            }
        if (L[k][k] != 0.0)
            for (int i = k + 1; i < 10; ++i)
                L[i][k] /= L[k][k];
        for (int i = k + 1; i < 10; ++i) {
            const double f = L[i][k];
            for (int j = k + 1; j < 10; ++j)
                L[i][j] -= f * L[k][j];
        }
    }
    for (int c = 0; c < 10; ++c) { // forward (unit lower) then backward (upper) substitution
        for (int i = 1; i < 10; ++i) {
            double s = X[i][c];
            for (int j = 0; j < i; ++j)
                s -= L[i][j] * X[j][c];
            X[i][c] = s;
        }
        for (int i = 9; i >= 0; --i) {
            double s = X[i][c];
            for (int j = i + 1; j < 10; ++j)
                s -= L[i][j] * X[j][c];
            X[i][c] = s / L[i][i];
        }
    }

    // Eliminate x^2 z, y^2 z, x y z using row pairs (4,5), (6,7), (8,9)   (relpose_5pt.cc:176-189).
    // Per row i: px_i(z) x + py_i(z) y + pc_i(z) = 0 with deg 3, 3, 4.  Stored highest power first.
    double Az[3][13];
    for (int i = 0; i < 3; ++i) {
        const double *ev = X[4 + 2 * i], *od = X[5 + 2 * i];
        Az[i][0] = 0.0 - od[0];
        Az[i][1] = ev[0] - od[1];
        Az[i][2] = ev[1] - od[2];
        Az[i][3] = ev[2];
        Az[i][4] = 0.0 - od[3];
        Az[i][5] = ev[3] - od[4];
        Az[i][6] = ev[4] - od[5];
        Az[i][7] = ev[5];
        Az[i][8] = 0.0 - od[6];
        Az[i][9] = ev[6] - od[7];
        Az[i][10] = ev[7] - od[8];
        Az[i][11] = ev[8] - od[9];
        Az[i][12] = ev[9];
    }
    // degree-10 determinant of [px py pc] by polynomial arithmetic (the reference uses the
    // expanded closed form, relpose_5pt.cc:192-352; same polynomial up to rounding)
    UPoly px[3], py[3], pc[3];
    for (int i = 0; i < 3; ++i) {
        px[i].deg = 3;
        py[i].deg = 3;
        pc[i].deg = 4;
        for (int k = 0; k <= 3; ++k) {
            px[i].c[k] = Az[i][3 - k];
            py[i].c[k] = Az[i][7 - k];
        }
        for (int k = 0; k <= 4; ++k)
            pc[i].c[k] = Az[i][12 - k];
    }
    double c[11] = {0};
    auto add_term = [&](double sgn, const UPoly &a, const UPoly &b, const UPoly &d) {
        const UPoly t = umul(umul(a, b), d);
        for (int k = 0; k <= t.deg; ++k)
            c[k] += sgn * t.c[k];
    };
    add_term(+1, px[0], py[1], pc[2]);
    add_term(-1, px[0], pc[1], py[2]);
    add_term(-1, py[0], px[1], pc[2]);
    add_term(+1, py[0], pc[1], px[2]);
    add_term(+1, pc[0], px[1], py[2]);
    add_term(-1, pc[0], py[1], px[2]);

    double roots[10];
    const int nroots = sturm_real_roots_deg10(c, roots);

    for (int s = 0; s < nroots; ++s) { // back-substitution (relpose_5pt.cc:366-392)
        const double z = roots[s];
        const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2;
        double B[3][2], b[3];
        for (int i = 0; i < 3; ++i) {
            B[i][0] = Az[i][0] * z3 + Az[i][1] * z2 + Az[i][2] * z + Az[i][3];
            B[i][1] = Az[i][4] * z3 + Az[i][5] * z2 + Az[i][6] * z + Az[i][7];
            b[i] = Az[i][8] * z4 + Az[i][9] * z3 + Az[i][10] * z2 + Az[i][11] * z + Az[i][12];
        }
        const double dt = B[0][0] * B[1][1] - B[1][0] * B[0][1];
        const double idt = 1.0 / dt;
        double u0 = (B[1][1] * idt) * b[0] + (-B[0][1] * idt) * b[1];
        double u1 = (-B[1][0] * idt) * b[0] + (B[0][0] * idt) * b[1];
        if (std::abs(B[2][0] * u0 + B[2][1] * u1 - b[2]) > 1e-6) {
            // least squares over all three rows (the reference uses colPivHouseholderQr, :381);
            // solved here through a 2-step Householder QR with column pivoting.
            double Q[3][2] = {{B[0][0], B[0][1]}, {B[1][0], B[1][1]}, {B[2][0], B[2][1]}};
            double rhs[3] = {b[0], b[1], b[2]};
            const double n0 = Q[0][0] * Q[0][0] + Q[1][0] * Q[1][0] + Q[2][0] * Q[2][0];
            const double n1 = Q[0][1] * Q[0][1] + Q[1][1] * Q[1][1] + Q[2][1] * Q[2][1];
            const bool swapc = n1 > n0;
            if (swapc)
                for (int i = 0; i < 3; ++i)
                    std::swap(Q[i][0], Q[i][1]);
            double Rm[2][2] = {{0, 0}, {0, 0}};
            for (int k = 0; k < 2; ++k) {
                double tail = 0;
                for (int i = k + 1; i < 3; ++i)
                    tail += Q[i][k] * Q[i][k];
                const double c0 = Q[k][k];
                double beta = std::sqrt(c0 * c0 + tail);
                if (c0 >= 0)
                    beta = -beta;
                double v[3] = {0, 0, 0};
                double tau = 0;
                if (tail > std::numeric_limits<double>::min()) {
                    v[k] = 1.0;
                    for (int i = k + 1; i < 3; ++i)
                        v[i] = Q[i][k] / (c0 - beta);
                    tau = (beta - c0) / beta;
                } else {
                    beta = c0;
                }
                Rm[k][k] = beta;
                for (int cc = k + 1; cc < 2; ++cc) {
                    double t = 0;
                    for (int i = k; i < 3; ++i)
                        t += v[i] * Q[i][cc];
                    for (int i = k; i < 3; ++i)
                        Q[i][cc] -= tau * v[i] * t;
                    Rm[k][cc] = Q[k][cc];
                }
                double t = 0;
                for (int i = k; i < 3; ++i)
                    t += v[i] * rhs[i];
                for (int i = k; i < 3; ++i)
                    rhs[i] -= tau * v[i] * t;
            }
            double w1 = rhs[1] / Rm[1][1];
            double w0 = (rhs[0] - Rm[0][1] * w1) / Rm[0][0];
            if (swapc)
                std::swap(w0, w1);
            u0 = w0;
            u1 = w1;
        }
        const double x = -u0, y = -u1;
        const double inv_norm = 1.0 / std::sqrt(x * x + y * y + z * z + 1.0);
        M3 E;
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                const int e = 3 * j + i;
                E.m[i][j] = (nb[0 * 9 + e] * x + nb[1 * 9 + e] * y + nb[2 * 9 + e] * z + nb[3 * 9 + e]) * inv_norm;
            }
        Eout[s] = E;
    }
    return nroots;
}

int relpose_5pt(const V3 x1[5], const V3 x2[5], Pose out[40]) { // relpose_5pt.cc:397-409
    M3 E[10];
    const int ne = essential_5pt(x1, x2, E);
    int n = 0;
    for (int i = 0; i < ne; ++i)
        n += motion_from_essential(E[i], x1, x2, 5, out + n);
    return n;
}

// =============================================================================== 7-point
int relpose_7pt(const V3 x1[7], const V3 x2[7], M3 F[3]) { // relpose_7pt.cc:10-60
    double A[63];
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i * 9 + 3 * j + 0] = x1[i][j] * x2[i].x;
            A[i * 9 + 3 * j + 1] = x1[i][j] * x2[i].y;
            A[i * 9 + 3 * j + 2] = x1[i][j] * x2[i].z;
        }
    double nb[18]; // 9 x 2
    householder_complement(A, 9, 2 + 5, nb);
    const double *n0 = nb, *n1 = nb + 9;

    // det(x*F0 + F1) as a cubic in x (:22-37).  The reference spells the 48 monomials out in the lexicographic
    // order of a computer-algebra expansion - factor 1 from matrix column 0 (vector entries 0..2), factor 2 from
    // column 1 (3..5), factor 3 from column 2 (6..8), each with its null-vector index 0 (the x part) before 1 -
    // every product evaluated left to right and every coefficient summed in that order.  The order is part of the
    // result: the roots, hence F, hence the rounding-level det(F) that decides the sign of the refined F
    // (eigen_shim/Eigen/src/JacobiSVD3x3.h) depend on the last bits of c3..c0.  Generated here by the loop nest
    // that enumerates the same order.
    double c[4] = {0, 0, 0, 0};
    for (int ra = 0; ra < 3; ++ra)
        for (int ka = 0; ka < 2; ++ka)
            for (int rb = 0; rb < 3; ++rb) {
                if (rb == ra)
                    continue;
                const int rc = 3 - ra - rb;
                const bool even = (ra == 0 && rb == 1) || (ra == 1 && rb == 2) || (ra == 2 && rb == 0);
                for (int kb = 0; kb < 2; ++kb)
                    for (int kc = 0; kc < 2; ++kc) {
                        const double t = nb[9 * ka + ra] * nb[9 * kb + 3 + rb] * nb[9 * kc + 6 + rc];
                        const int deg = 3 - (ka + kb + kc);
                        c[deg] = even ? c[deg] + t : c[deg] - t;
                    }
            }

    double roots[3];
    int nr;
    if (std::abs(c[3]) < 1e-14) {
        nr = quadratic_real_roots(c[2], c[1], c[0], roots);
    } else {
        const double inv = 1.0 / c[3];
        nr = cubic_real_roots(c[2] * inv, c[1] * inv, c[0] * inv, roots);
    }
    for (int s = 0; s < nr; ++s) {
        double f[9], nn = 0;
        for (int k = 0; k < 9; ++k) {
            f[k] = n0[k] * roots[s] + n1[k];
            nn += f[k] * f[k];
        }
        nn = std::sqrt(nn);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i)
                F[s].m[i][j] = f[3 * j + i] / nn;
    }
    return nr;
}

// =============================================================================== homography
int homography_4pt(const V3 x1[4], const V3 x2[4], M3 *H, bool check) { // homography_4pt.cc:36-128
    if (check) { // orientation consistency of the four points (:38-55)
        V3 p = cross(x1[0], x1[1]), q = cross(x2[0], x2[1]);
        if (dot(p, x1[2]) * dot(q, x2[2]) < 0)
            return 0;
        if (dot(p, x1[3]) * dot(q, x2[3]) < 0)
            return 0;
        p = cross(x1[2], x1[3]);
        q = cross(x2[2], x2[3]);
        if (dot(p, x1[0]) * dot(q, x2[0]) < 0)
            return 0;
        if (dot(p, x1[1]) * dot(q, x2[1]) < 0)
            return 0;
    }
    double ax[4], ay[4], bx[4], by[4];
    for (int i = 0; i < 4; ++i) {
        ax[i] = x1[i].x / x1[i].z;
        ay[i] = x1[i].y / x1[i].z;
        bx[i] = x2[i].x / x2[i].z;
        by[i] = x2[i].y / x2[i].z;
    }
    // source-plane affine part
    const double n1x = ax[1] - ax[0], p1x = ax[2] - ax[0], q1x = ax[3] - ax[0];
    const double n1y = ay[1] - ay[0], p1y = ay[2] - ay[0], q1y = ay[3] - ay[0];
    const double fA1 = n1x * p1y - n1y * p1x;
    const double Q3x = p1y * q1x - p1x * q1y;
    const double Q3y = n1x * q1y - n1y * q1x;
    // target-plane affine part
    const double n2x = bx[1] - bx[0], p2x = bx[2] - bx[0], q2x = bx[3] - bx[0];
    const double n2y = by[1] - by[0], p2y = by[2] - by[0], q2y = by[3] - by[0];
    const double fA2 = n2x * p2y - n2y * p2x;
    const double Q4x = p2y * q2x - p2x * q2y;
    const double Q4y = n2x * q2y - n2y * q2x;
    // core transformation
    const double tt1 = fA1 - Q3x - Q3y;
    const double C11 = Q3y * Q4x * tt1;
    const double C22 = Q3x * Q4y * tt1;
    const double C33 = Q3x * Q3y * (fA2 - Q4x - Q4y);
    const double C31 = C11 - C33;
    const double C32 = C22 - C33;
    const double tt3 = bx[0] * C33;
    const double tt4 = by[0] * C33;
    const double H11 = bx[1] * C11 - tt3;
    const double H12 = bx[2] * C22 - tt3;
    const double H21 = by[1] * C11 - tt4;
    const double H22 = by[2] * C22 - tt4;
    double h[9];
    h[0] = H11 * p1y - H12 * n1y;
    h[1] = H12 * n1x - H11 * p1x;
    h[3] = H21 * p1y - H22 * n1y;
    h[4] = H22 * n1x - H21 * p1x;
    h[6] = C31 * p1y - C32 * n1y;
    h[7] = C32 * n1x - C31 * p1x;
    h[2] = tt3 * fA1 - h[0] * ax[0] - h[1] * ay[0];
    h[5] = tt4 * fA1 - h[3] * ax[0] - h[4] * ay[0];
    h[8] = C33 * fA1 - h[6] * ax[0] - h[7] * ay[0];
    M3 Hm; // h is row-major H
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Hm.m[i][j] = h[3 * i + j];
    const double nrm = frob(Hm);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Hm.m[i][j] = Hm.m[i][j] / nrm;
    *H = Hm;
    if (std::abs(det(Hm)) < 1e-8)
        return 0;
    return 1;
}

} // namespace orc
