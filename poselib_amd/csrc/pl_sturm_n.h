// poselib_amd - real roots of a polynomial of degree N by Sturm bisection, the reference's bisect_sturm<N> (misc/sturm.h:233-274:
// monic normalisation, chain :47-84, Cauchy bound :144-150, isolation by bisection :210-231 in recursion order - depth first, left
// half first -, Ridders' method then Newton on an isolating interval :153-208).  The degree-10 instance of the 5-point solver is
// pl_solver_rel.h's (tuned for the batched generator: flat isolation, leaves ranked afterwards); this is the plain form for any
// N <= 15 with the tolerance as a parameter - the shared-focal six-point solver calls bisect_sturm<15>(p, roots, 1e-12) on the
// characteristic polynomial of its action matrix (relpose_6pt_focal.cc:1069-1076).  One lane runs it; same operations, same order
// as oracle/src/solvers_rel.cc sturm_real_roots (tests/hostmath compares the two bit for bit).
#pragma once
#include "pl_math.h"

namespace pl {

template <int N> struct SturmN {
    double f[N + 1]; // monic
    double fp[N];    // derivative / N: monic of degree N - 1
    double q0[N - 1], q1[N - 1], c[N - 1];
    double tail0, tail1, last;
};

template <int N> PL_HD double sturm_n_horner(const double *p, int deg, double x) {
    double v = x + p[deg - 1];
    for (int i = deg - 2; i >= 0; --i)
        v = x * v + p[i];
    return v;
}

template <int N> PL_HD void sturm_n_build(SturmN<N> &S) { // sturm.h:47-84
    double buf[3][N + 1];
    int hi = 0, lo = 1, rem = 2;
    for (int i = 0; i <= N; ++i)
        buf[0][i] = S.f[i];
    for (int i = 0; i < N; ++i)
        buf[1][i] = S.fp[i];
    for (int i = 0; i < N - 1; ++i) {
        const int dh = N - i, dl = N - 1 - i;
        const double a1 = buf[hi][dh] * buf[lo][dl];
        const double a0 = buf[hi][dh - 1] * buf[lo][dl] - buf[hi][dh] * buf[lo][dl - 1];
        buf[rem][0] = buf[hi][0] - a0 * buf[lo][0];
        for (int j = 1; j < dl; ++j)
            buf[rem][j] = buf[hi][j] - a1 * buf[lo][j - 1] - a0 * buf[lo][j];
        const double scale = -fabs(buf[rem][dl - 1]);
        const double inv = 1.0 / scale;
        for (int j = 0; j < dl; ++j)
            buf[rem][j] = buf[rem][j] * inv;
        S.q0[i] = a0;
        S.q1[i] = a1;
        S.c[i] = scale;
        const int t = hi;
        hi = lo;
        lo = rem;
        rem = t;
    }
    S.tail0 = buf[hi][0];
    S.tail1 = buf[hi][1];
    S.last = buf[lo][0];
}

template <int N> PL_HD int sturm_n_variations(const SturmN<N> &S, double x) { // sturm.h:98-112
    double up2 = S.last;
    double up1 = S.tail0 + x * S.tail1;
    int count = ((up1 < 0) != (up2 < 0)) ? 1 : 0;
    for (int i = N - 2; i >= 0; --i) {
        const double v = (S.q0[i] + x * S.q1[i]) * up1 + S.c[i] * up2;
        count += ((v < 0) != (up1 < 0)) ? 1 : 0;
        up2 = up1;
        up1 = v;
    }
    return count;
}

template <int N> PL_HD void sturm_n_polish(const SturmN<N> &S, double a, double b, double *roots, int &n, double tol) { // sturm.h:153-208
    double fa = sturm_n_horner<N>(S.f, N, a);
    double fb = sturm_n_horner<N>(S.f, N, b);
    if (!((fa < 0) ^ (fb < 0)))
        return;
    for (int it = 0; it < 30; ++it) {
        if (fabs(a - b) < 1e-3)
            break;
        const double c = (a + b) * 0.5;
        const double fc = sturm_n_horner<N>(S.f, N, c);
        const double s = sqrt(fc * fc - fa * fb);
        if (!s)
            break;
        const double d = (fa < fb) ? c + (a - c) * fc / s : c + (c - a) * fc / s;
        const double fd = sturm_n_horner<N>(S.f, N, d);
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
        const double fx = sturm_n_horner<N>(S.f, N, x);
        if (fabs(fx) < tol)
            break;
        const double fpx = (double)N * sturm_n_horner<N>(S.fp, N - 1, x);
        const double dx = fx / fpx;
        x = x - dx;
        if (fabs(dx) < tol)
            break;
    }
    roots[n++] = x;
}

// coef[0 .. N] (coef[N] the leading coefficient) -> real roots in the reference's order; returns their number (<= N)
template <int N> PL_HD int sturm_n_roots(const double *coef, double *roots, double tol) {
    static_assert(N <= 15, "the sign-variation counts travel in 4 bits");
    if (coef[N] == 0.0)
        return 0;
    SturmN<N> S;
    const double lead_inv = 1.0 / coef[N];
    for (int i = 0; i < N; ++i)
        S.f[i] = coef[i] * lead_inv;
    S.f[N] = 1.0;
    for (int i = 0; i < N - 1; ++i)
        S.fp[i] = S.f[i + 1] * ((i + 1) / (double)N);
    S.fp[N - 1] = 1.0;
    sturm_n_build(S);
    double bound = 0;
    for (int i = 0; i < N; ++i)
        bound = fmax(bound, fabs(S.f[i]));
    bound = 1.0 + bound;
    const int sa0 = sturm_n_variations(S, -bound), sb0 = sturm_n_variations(S, bound);
    if (sa0 - sb0 == 0)
        return 0;
    // the recursion as a loop: the left half now, the right half deferred - kept only if visiting it has an effect (it holds a sign
    // variation, or it is narrower than tol: the reference reports the right end of such an interval whatever the counts say), so
    // at most N + 1 deferred halves are alive
    constexpr int kCap = N + 2;
    double sa_[kCap], sb_[kCap];
    unsigned si_[kCap];
    double a = -bound, b = bound;
    int sa = sa0, sb = sb0, depth = 0, sp = 0, n = 0;
    for (;;) {
        bool descend = false;
        if (depth <= 300) { // MAX_STURM_RECURSION_DEPTH_LIMIT
            if (b - a < tol) {
                if (n < N)
                    roots[n++] = b;
            } else {
                const int k = sa - sb;
                if (k > 1) {
                    const double mid = (a + b) * 0.5;
                    const int sm = sturm_n_variations(S, mid);
                    if ((sm - sb >= 1 || b - mid < tol) && sp < kCap) {
                        sa_[sp] = mid, sb_[sp] = b, si_[sp] = (unsigned)sm | ((unsigned)sb << 4) | ((unsigned)(depth + 1) << 8);
                        ++sp;
                    }
                    b = mid;
                    sb = sm;
                    depth += 1;
                    descend = true;
                } else if (k == 1 && n < N) {
                    sturm_n_polish(S, a, b, roots, n, tol);
                }
            }
        }
        if (descend)
            continue;
        if (sp == 0)
            break;
        --sp;
        a = sa_[sp], b = sb_[sp];
        sa = (int)(si_[sp] & 0xfu);
        sb = (int)((si_[sp] >> 4) & 0xfu);
        depth = (int)(si_[sp] >> 8);
    }
    return n;
}

} // namespace pl
