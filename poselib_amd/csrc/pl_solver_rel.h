// poselib_amd — two-view minimal solvers for one lane: 5-point essential (Nister) and 7-point
// fundamental, with their helpers.
//
// Reference semantics (citations relative to /root/reference/PoseLib):
//   misc/sturm.h:47-84,98-112,144-150,153-208,210-231,233-274   Sturm chain, sign variations, Cauchy bound,
//                                     Ridders + Newton polish, interval isolation (depth-first, left first)
//   misc/essential.cc:103-169         motion_from_essential: closed-form factorisation, 4 candidates, cheirality
//                                     on the sample points (min depth 0)
//   solvers/relpose_5pt.cc:101-157    trace + determinant constraints (10 x 20, Nister's monomial order)
//   solvers/relpose_5pt.cc:159-395    null space, 10x10 LU, elimination to a 3x3 polynomial matrix, degree-10
//                                     determinant, Sturm roots, back substitution, normalisation
//   solvers/relpose_7pt.cc:10-60      2-D null space, cubic in the mixing parameter, <= 3 F
//   misc/univariate.cc:48-61,94-126   real quadratic / all real cubic roots + one Newton step
//   robust/utils.cc:646-671           real-focal-length check (evaluated in float)
// The orthonormal null-space basis follows the algorithm of Eigen's fullPivHouseholderQr().matrixQ()
// (pivot = largest |a_ij| of the trailing corner, first maximum in column-major order).
#pragma once
#include "pl_math.h"

namespace pl {

// =============================================================================== univariate
PL_HD int quadratic_real_roots(double a, double b, double c, double *r) {
    const double disc = b * b - 4 * a * c;
    if (disc < 0)
        return 0;
    const double s = sqrt(disc);
    r[0] = (b > 0) ? (2 * c) / (-b - s) : (2 * c) / (-b + s);
    r[1] = c / (a * r[0]);
    return 2;
}
PL_HD int cubic_real_roots(double c2, double c1, double c0, double *r) {
    const double a = c1 - c2 * c2 / 3.0;
    double b = (2.0 * c2 * c2 * c2 - 9.0 * c2 * c1) / 27.0 + c0;
    double c = b * b / 4.0 + a * a * a / 27.0;
    int n;
    if (a == 0.0 && b == 0.0) {
        r[0] = r[1] = r[2] = -c2 / 3.0;
        n = 3;
    } else if (c > 0) {
        c = sqrt(c);
        b *= -0.5;
        r[0] = pl_cbrt(b + c) + pl_cbrt(b - c) - c2 / 3.0;
        n = 1;
    } else {
        c = 3.0 * b / (2.0 * a) * sqrt(-3.0 / a);
        const double d = 2.0 * sqrt(-a / 3.0);
        r[0] = d * pl_cos(pl_acos(c) / 3.0) - c2 / 3.0;
        r[1] = d * pl_cos(pl_acos(c) / 3.0 - 2.09439510239319526263557236234192) - c2 / 3.0;
        r[2] = d * pl_cos(pl_acos(c) / 3.0 - 4.18879020478639052527114472468384) - c2 / 3.0;
        n = 3;
    }
    for (int i = 0; i < n; ++i) {
        const double x = r[i];
        const double x2 = x * x;
        const double x3 = x * x2;
        const double dx = -(x3 + c2 * x2 + c1 * x + c0) / (3 * x2 + 2 * c2 * x + c1);
        r[i] += dx;
    }
    return n;
}

// =============================================================================== Sturm (degree 10)
struct Sturm10 {
    double f[11];  // monic
    double fp[10]; // derivative / 10, monic degree 9
    double q0[9], q1[9], c[9];
    double tail0, tail1, last;
};

PL_HD double horner_monic(const double *p, int deg, double x) {
    double v = x + p[deg - 1];
    for (int i = deg - 2; i >= 0; --i)
        v = x * v + p[i];
    return v;
}

PL_HD void sturm_build(Sturm10 &S) {
    constexpr int N = 10;
    double buf[3][11];
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

PL_HD int sturm_variations(const Sturm10 &S, double x) {
    constexpr int N = 10;
    double up2 = S.last;                    // v[i+2]
    double up1 = S.tail0 + x * S.tail1;     // v[i+1]
    int count = ((up1 < 0) != (up2 < 0)) ? 1 : 0;
    for (int i = N - 2; i >= 0; --i) {
        const double v = (S.q0[i] + x * S.q1[i]) * up1 + S.c[i] * up2;
        count += ((v < 0) != (up1 < 0)) ? 1 : 0;
        up2 = up1;
        up1 = v;
    }
    return count;
}

PL_HD void sturm_polish(const Sturm10 &S, double a, double b, double *roots, int &n, double tol) {
    constexpr int N = 10;
    double fa = horner_monic(S.f, N, a);
    double fb = horner_monic(S.f, N, b);
    if (!((fa < 0) ^ (fb < 0)))
        return;
    for (int it = 0; it < 30; ++it) {
        if (fabs(a - b) < 1e-3)
            break;
        const double c = (a + b) * 0.5;
        const double fc = horner_monic(S.f, N, c);
        const double s = sqrt(fc * fc - fa * fb);
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
        if (fabs(fx) < tol)
            break;
        const double fpx = (double)N * horner_monic(S.fp, N - 1, x);
        const double dx = fx / fpx;
        x = x - dx;
        if (fabs(dx) < tol)
            break;
    }
    roots[n++] = x;
}

// Work space of the root isolation: a stack of deferred right halves and the list of leaf intervals, both tiny (a
// deferred half is kept only if it can still produce a root, so at most ten are alive).  The host build and the bare
// solver entry points keep it in local arrays; the batched generator keeps it in LDS, one column per lane
// (kernels.hip), so that nothing of the root finder lives in scratch memory.
constexpr int kSturmSlots = 12;
struct SturmWorkLocal {
    static constexpr int kStackCap = kSturmSlots; // deferred halves that can be alive (see sturm_isolate_chain)
    double sa[kSturmSlots], sb[kSturmSlots], la[kSturmSlots], lb[kSturmSlots];
    unsigned si[kSturmSlots];
    PL_HD void push(int i, double a, double b, unsigned info) { sa[i] = a, sb[i] = b, si[i] = info; }
    PL_HD void pop(int i, double &a, double &b, unsigned &info) const { a = sa[i], b = sb[i], info = si[i]; }
    PL_HD void leaf_set(int i, double a, double b) { la[i] = a, lb[i] = b; }
    PL_HD void leaf_get(int i, double &a, double &b) const { a = la[i], b = lb[i]; }
    // flat isolation (sturm_isolate_flat): intervals to bisect share the stack arrays, leaves as found their own
    static constexpr int kPendCap = kSturmSlots, kLeafCap = 16;
    double ua[kLeafCap], ub[kLeafCap];
    PL_HD void pend_push(int i, double a, double b, unsigned info) { push(i, a, b, info); }
    PL_HD void pend_pop(int i, double &a, double &b, unsigned &info) const { pop(i, a, b, info); }
    PL_HD void uleaf_set(int i, double a, double b) { ua[i] = a, ub[i] = b; }
    PL_HD void uleaf_get(int i, double &a, double &b) const { a = ua[i], b = ub[i]; }
};

// Real roots of c[0] + c[1] z + ... + c[10] z^10, in the order the reference's recursion emits them (depth first,
// left half first: sturm.h:210-231).  Two phases so that the lanes of a wavefront stay together: (1) the bisection
// only records its leaves - intervals narrower than tol (the reference reports their right end as a root whatever
// the sign-variation count says) and intervals holding exactly one root - in recursion order; (2) the leaves are
// turned into roots one after the other (Ridders + Newton polish for the isolating intervals).  A right half without
// a sign variation is deferred only if it is narrower than tol (the one case in which visiting it has an effect).
// f (monic) and f' / N of the Sturm chain from the coefficients; false when the leading coefficient is zero
PL_HD bool sturm_monic(const double *coef, Sturm10 &S) {
    constexpr int N = 10;
    if (coef[N] == 0.0)
        return false;
    const double lead_inv = 1.0 / coef[N];
    PL_UNROLL
    for (int i = 0; i < N; ++i)
        S.f[i] = coef[i] * lead_inv;
    S.f[N] = 1.0;
    PL_UNROLL
    for (int i = 0; i < N - 1; ++i)
        S.fp[i] = S.f[i + 1] * ((i + 1) / (double)N);
    S.fp[N - 1] = 1.0;
    return true;
}

// phase 0: monic f, f' / N, the Sturm chain, the Cauchy bound and the sign-variation counts at its two ends.  Returns
// the number of distinct real roots the chain sees in (-bound, bound) - 0 also when the leading coefficient vanishes
// or the counts are inconsistent (sa0 < sb0: the recursion of sturm.h:210-231 then visits no interval of interest)
PL_HD int sturm_prepare(const double *coef, Sturm10 &S, double &bound, int &sa0, int &sb0) {
    constexpr int N = 10;
    bound = 0;
    sa0 = sb0 = 0;
    if (!sturm_monic(coef, S))
        return 0;
    sturm_build(S);
    PL_UNROLL
    for (int i = 0; i < N; ++i)
        bound = fmax(bound, fabs(S.f[i]));
    bound = 1.0 + bound;
    sa0 = sturm_variations(S, -bound), sb0 = sturm_variations(S, bound);
    return sa0 - sb0 > 0 ? sa0 - sb0 : 0;
}

// phase 1: the leaves of the bisection in recursion order (w.leaf_*); bit i of `tiny`: leaf i is narrower than tol.
// (S, bound, sa0, sb0) from sturm_prepare; sa0 - sb0 > 0.
template <class Work> PL_HD int sturm_isolate_chain(const Sturm10 &S, double bound, int sa0, int sb0, Work &w, unsigned &tiny) {
    const double tol = 1e-10;
    tiny = 0;
    double a = -bound, b = bound;
    int sa = sa0, sb = sb0, depth = 0;
    int sp = 0, nleaf = 0;
    for (;;) {
        bool descend = false;
        if (depth <= 300) { // MAX_STURM_RECURSION_DEPTH_LIMIT
            if (b - a < tol) {
                if (nleaf < kSturmSlots) {
                    w.leaf_set(nleaf, a, b);
                    tiny |= 1u << nleaf;
                    ++nleaf;
                }
            } else {
                const int k = sa - sb;
                if (k > 1) {
                    const double mid = (a + b) * 0.5;
                    const int sm = sturm_variations(S, mid);
                    // (at most TEN deferred halves are ever alive: a half is deferred with >= 1 sign variation - at most 8 of
                    // those next to the current interval's >= 2 - or as a narrow one, which happens at the bottom of a descent
                    // and is popped right after the narrow left leaf; Work::kStackCap >= 10 therefore never drops one)
                    if ((sm - sb >= 1 || b - mid < tol) && sp < Work::kStackCap) { // right half (mid, b): later
                        w.push(sp, mid, b, (unsigned)sm | ((unsigned)sb << 4) | ((unsigned)(depth + 1) << 8));
                        ++sp;
                    }
                    b = mid; // left half (a, mid): now
                    sb = sm;
                    depth += 1;
                    descend = true;
                } else if (k == 1) {
                    if (nleaf < kSturmSlots) {
                        w.leaf_set(nleaf, a, b);
                        ++nleaf;
                    }
                }
            }
        }
        if (descend)
            continue;
        if (sp == 0)
            break;
        --sp;
        unsigned info;
        w.pop(sp, a, b, info);
        sa = (int)(info & 0xfu);
        sb = (int)((info >> 4) & 0xfu);
        depth = (int)(info >> 8);
    }
    return nleaf;
}
template <class Work> PL_HD int sturm_isolate(const double *coef, Work &w, unsigned &tiny) {
    Sturm10 S;
    double bound;
    int sa0, sb0;
    tiny = 0;
    if (sturm_prepare(coef, S, bound, sa0, sb0) == 0)
        return 0;
    return sturm_isolate_chain(S, bound, sa0, sb0, w, tiny);
}

// phase 1, flat form (round 5; the batched generator): the same leaves from a loop whose EVERY round is one Sturm
// evaluation.  The recursion visits an interval (narrow -> leaf; more than one sign variation -> bisect; exactly one ->
// leaf; none -> nothing) and a visit costs nothing but comparisons - only the bisection evaluates the chain.  So the work
// list holds just the intervals that still have to be bisected (each carries >= 2 roots: at most five are alive), a round
// pops one, evaluates the chain at its midpoint and visits both halves at once (the right half under the reference's
// condition: it holds a sign variation or is narrow).  The recursion's loop needed a round for every visit - bisections,
// leaves, empty halves: about 3.3 per leaf on top of the evaluations - each with its own branches; here a wavefront runs
// max-over-lanes(evaluations) rounds of straight-line code.  The leaves are found in a different ORDER (depth first,
// left first = ascending interval order in the recursion); they are disjoint intervals, so their rank by left end
// restores the recursion's order: sturm_rank_leaves.  Work: pend_push / pend_pop (intervals to bisect), uleaf_set /
// uleaf_get (leaves as found), leaf_set (leaves in the recursion's order).
template <class Work> PL_HD int sturm_isolate_flat(const Sturm10 &S, double bound, int sa0, int sb0, Work &w, unsigned &tiny_found) {
    const double tol = 1e-10;
    tiny_found = 0;
    int np = 0, nl = 0;
    auto visit = [&](double a, double b, int sa, int sb, int depth) {
        if (depth > 300) // MAX_STURM_RECURSION_DEPTH_LIMIT
            return;
        const int k = sa - sb;
        const bool narrow = b - a < tol;
        if (narrow || k == 1) {
            if (nl < Work::kLeafCap) {
                w.uleaf_set(nl, a, b);
                tiny_found |= narrow ? (1u << nl) : 0u;
                ++nl;
            }
        } else if (k > 1 && np < Work::kPendCap) {
            w.pend_push(np, a, b, (unsigned)sa | ((unsigned)sb << 4) | ((unsigned)depth << 8));
            ++np;
        }
    };
    visit(-bound, bound, sa0, sb0, 0);
    while (np > 0) {
        --np;
        double a, b;
        unsigned info;
        w.pend_pop(np, a, b, info);
        const int sa = (int)(info & 0xfu), sb = (int)((info >> 4) & 0xfu), depth = (int)(info >> 8);
        const double mid = (a + b) * 0.5;
        const int sm = sturm_variations(S, mid);
        // (the right half first: it is pushed BELOW the left one, so the left subtree is bisected first like in the
        // recursion - not needed for the result, but it keeps the list as short as the recursion's stack)
        if (sm - sb >= 1 || b - mid < tol)
            visit(mid, b, sm, sb, depth + 1);
        visit(a, mid, sa, sm, depth + 1);
    }
    return nl;
}
// Leaves as found -> the recursion's order (ascending left end; the intervals are disjoint).  Keeps the first
// kSturmSlots of them like the recursion's list; bit i of `tiny`: leaf i (in order) is narrower than tol.
template <class Work> PL_HD int sturm_rank_leaves(int nl, unsigned tiny_found, Work &w, unsigned &tiny) {
    tiny = 0;
    int kept = 0;
    for (int i = 0; i < nl; ++i) {
        double ai, bi;
        w.uleaf_get(i, ai, bi);
        int rank = 0;
        for (int j = 0; j < nl; ++j) {
            double aj, bj;
            w.uleaf_get(j, aj, bj);
            rank += (aj < ai) ? 1 : 0;
        }
        if (rank < kSturmSlots) {
            w.leaf_set(rank, ai, bi);
            tiny |= ((tiny_found >> i) & 1u) << rank;
            ++kept;
        }
    }
    return kept;
}

// phase 2, one leaf: its root (the right end of a narrow leaf; Ridders + Newton on an isolating one, which reports
// nothing when the end points do not bracket a sign change).  Returns the number of roots written (0 or 1).
PL_HD int sturm_leaf_root(const Sturm10 &S, double la, double lb, bool is_tiny, double *root) {
    if (is_tiny) {
        *root = lb;
        return 1;
    }
    int n = 0;
    sturm_polish(S, la, lb, root, n, 1e-10);
    return n;
}

template <class Work> PL_HD int sturm_roots_deg10(const double *coef, double *roots, Work &w) {
    constexpr int N = 10;
    unsigned tiny;
    const int nleaf = sturm_isolate(coef, w, tiny);
    if (nleaf == 0)
        return 0;
    Sturm10 S;
    sturm_monic(coef, S);
    int n = 0;
    for (int i = 0; i < nleaf; ++i) {
        double la, lb;
        w.leaf_get(i, la, lb);
        if (n < N)
            n += sturm_leaf_root(S, la, lb, (tiny >> i) & 1u, roots + n);
    }
    return n;
}
PL_HD int sturm_roots_deg10(const double *coef, double *roots) {
    SturmWorkLocal w;
    return sturm_roots_deg10(coef, roots, w);
}
// the same roots through the flat isolation (what the batched generator runs; tests compare the two bit for bit)
template <class Work> PL_HD int sturm_roots_deg10_flat(const double *coef, double *roots, Work &w) {
    constexpr int N = 10;
    Sturm10 S;
    double bound;
    int sa0, sb0;
    if (sturm_prepare(coef, S, bound, sa0, sb0) == 0)
        return 0;
    unsigned tiny_found, tiny;
    const int nl = sturm_isolate_flat(S, bound, sa0, sb0, w, tiny_found);
    const int nleaf = sturm_rank_leaves(nl, tiny_found, w, tiny);
    int n = 0;
    for (int i = 0; i < nleaf; ++i) {
        double la, lb;
        w.leaf_get(i, la, lb);
        if (n < N)
            n += sturm_leaf_root(S, la, lb, (tiny >> i) & 1u, roots + n);
    }
    return n;
}

// =============================================================================== null space
// Orthonormal basis of the complement of span(columns of A); A is 9 x COLS column-major.
// basis: 9 x (9-COLS), column-major.
// Every array index below is a compile-time constant once the loops are unrolled - pivot rows / columns are swapped
// by comparison-selected exchanges, never through a run-time index - so that the device keeps the matrix in
// registers instead of scratch memory.
PL_HD void cswap(bool sw, double &a, double &b) {
    const double x = a, y = b;
    a = sw ? y : x;
    b = sw ? x : y;
}
template <int COLS> PL_HD void complement_basis9(double *qr /* 9*COLS, destroyed */, double *basis) {
    constexpr int ROWS = 9;
    double tau[COLS];
    int rowswap[COLS];
    PL_UNROLL
    for (int i = 0; i < COLS; ++i) {
        tau[i] = 0;
        rowswap[i] = i;
    }
    double biggest = 0;
    const double precision = 2.220446049250313e-16 * COLS;
    bool live = true; // false once the remaining corner is negligible (the reference breaks out of the loop)
    PL_UNROLL
    for (int k = 0; k < COLS; ++k) {
        if (!live)
            continue;
        int pr = k, pc = k;
        double best = fabs(qr[k * ROWS + k]);
        PL_UNROLL
        for (int c = k; c < COLS; ++c)
            PL_UNROLL
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
            live = false;
            continue;
        }
        rowswap[k] = pr;
        PL_UNROLL
        for (int r = k + 1; r < ROWS; ++r) {
            const bool sw = pr == r;
            PL_UNROLL
            for (int c = k; c < COLS; ++c)
                cswap(sw, qr[c * ROWS + k], qr[c * ROWS + r]);
        }
        PL_UNROLL
        for (int cc = k + 1; cc < COLS; ++cc) {
            const bool sw = pc == cc;
            PL_UNROLL
            for (int r = 0; r < ROWS; ++r)
                cswap(sw, qr[k * ROWS + r], qr[cc * ROWS + r]);
        }
        double tail_sq = 0;
        PL_UNROLL
        for (int r = k + 1; r < ROWS; ++r)
            tail_sq += qr[k * ROWS + r] * qr[k * ROWS + r];
        const double c0 = qr[k * ROWS + k];
        double beta;
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0;
            beta = c0;
            PL_UNROLL
            for (int r = k + 1; r < ROWS; ++r)
                qr[k * ROWS + r] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0)
                beta = -beta;
            PL_UNROLL
            for (int r = k + 1; r < ROWS; ++r)
                qr[k * ROWS + r] = qr[k * ROWS + r] / (c0 - beta);
            tau[k] = (beta - c0) / beta;
        }
        qr[k * ROWS + k] = beta;
        if (tau[k] != 0) {
            PL_UNROLL
            for (int c = k + 1; c < COLS; ++c) {
                double t = 0;
                PL_UNROLL
                for (int r = k + 1; r < ROWS; ++r)
                    t += qr[k * ROWS + r] * qr[c * ROWS + r];
                t += qr[c * ROWS + k];
                qr[c * ROWS + k] -= tau[k] * t;
                PL_UNROLL
                for (int r = k + 1; r < ROWS; ++r)
                    qr[c * ROWS + r] -= tau[k] * qr[k * ROWS + r] * t;
            }
        }
    }
    // columns COLS..8 of Q = (P0 H0)(P1 H1)... applied to unit vectors
    PL_UNROLL
    for (int j = 0; j < ROWS - COLS; ++j) {
        double v[ROWS];
        PL_UNROLL
        for (int r = 0; r < ROWS; ++r)
            v[r] = (r == COLS + j) ? 1.0 : 0.0;
        PL_UNROLL
        for (int k = COLS - 1; k >= 0; --k) {
            if (tau[k] != 0) {
                double t = 0;
                PL_UNROLL
                for (int r = k + 1; r < ROWS; ++r)
                    t += qr[k * ROWS + r] * v[r];
                t += v[k];
                v[k] -= tau[k] * t;
                PL_UNROLL
                for (int r = k + 1; r < ROWS; ++r)
                    v[r] -= tau[k] * qr[k * ROWS + r] * t;
            }
            PL_UNROLL
            for (int r = k + 1; r < ROWS; ++r)
                cswap(rowswap[k] == r, v[k], v[r]);
        }
        PL_UNROLL
        for (int r = 0; r < ROWS; ++r)
            basis[j * ROWS + r] = v[r];
    }
}

// The same algorithm with run-time row / column indices: fewer instructions (no comparison-selected exchanges), but
// the matrix lives in scratch memory on the device.  Used by the 7-point solver, whose 9 x 7 matrix would cost more
// in exchanges than the scratch accesses do (measured: 295 vs 165 us per 100 k samples).
template <int COLS> PL_HD void complement_basis9_indexed(double *qr /* 9*COLS, destroyed */, double *basis) {
    constexpr int ROWS = 9;
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
    // columns COLS..8 of Q = (P0 H0)(P1 H1)... applied to unit vectors
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

// =============================================================================== essential -> motion
struct PoseQT {
    Quat q;
    Vec3 t;
};

template <int NP> PL_HD bool cheirality_all(Quat q, Vec3 t, const Vec3 *x1, const Vec3 *x2) {
    for (int i = 0; i < NP; ++i)
        if (!check_cheirality(q, t, x1[i], x2[i], 0.0))
            return false;
    return true;
}
// cheirality_all for t (pos) and for -t (neg) in one pass.  With t negated b1, b2 and with them l1, l2 change sign
// exactly (every product and sum is the negated one, IEEE rounding is symmetric), and the depth bound is +-0: the test of
// (q, -t) is "l1 < 0 and l2 < 0" on the numbers computed for (q, t).
template <int NP> PL_HD void cheirality_pair(Quat q, Vec3 t, const Vec3 *x1, const Vec3 *x2, bool &pos, bool &neg) {
    pos = neg = true;
    PL_UNROLL
    for (int i = 0; i < NP; ++i) { // (unrolled: the points stay in registers - a run-time index puts them into scratch memory)
        if (pos || neg) {
            double l1, l2, a;
            cheirality_depths(q, t, x1[i], x2[i], l1, l2, a);
            pos = pos && (l1 > 0.0 && l2 > 0.0);
            neg = neg && (l1 < 0.0 && l2 < 0.0);
        }
    }
}

// essential.cc:103-169.  emit(q, t) is called for the candidates that pass cheirality on the NP sample points, in the
// reference's order.
template <int NP, class Emit> PL_HD void motion_from_essential_emit(const Mat3 &E, const Vec3 *x1, const Vec3 *x2, Emit &&emit) {
    const Vec3 e0 = col(E, 0), e1 = col(E, 1), e2 = col(E, 2);
    const Vec3 u12 = cross(e0, e1), u13 = cross(e0, e2), u23 = cross(e1, e2);
    const double n12 = dot(u12, u12), n13 = dot(u13, u13), n23 = dot(u23, u23);
    Vec3 c1, c2;
    if (n12 > n13) {
        if (n12 > n23) {
            c1 = normalized(e0);
            c2 = u12 / sqrt(n12);
        } else {
            c1 = normalized(e1);
            c2 = u23 / sqrt(n23);
        }
    } else {
        if (n13 > n23) {
            c1 = normalized(e0);
            c2 = u13 / sqrt(n13);
        } else {
            c1 = normalized(e1);
            c2 = u23 / sqrt(n23);
        }
    }
    const Vec3 c0 = -cross(c2, c1);
    Vec3 r0 = mul_t(E, c1);
    Vec3 r1 = mul_t(E, -c0);
    r0 = normalized(r0);
    r1 = r1 - dot(r0, r1) * r0;
    r1 = normalized(r1);
    const Vec3 r2 = cross(r0, r1);
    Mat3 Vt, UW;
    set_row(Vt, 0, r0);
    set_row(Vt, 1, r1);
    set_row(Vt, 2, r2);
    set_col(UW, 0, c0);
    set_col(UW, 1, c1);
    set_col(UW, 2, c2);
    Quat q = rotmat_to_quat(mul(UW, Vt));
    const Vec3 t = c2, tn = -c2;
    bool pos, neg;
    cheirality_pair<NP>(q, t, x1, x2, pos, neg); // essential.cc:152-157: (q, t), (q, -t) ...
    if (pos)
        emit(q, t);
    if (neg)
        emit(q, tn);
    set_col(UW, 0, -c0);
    set_col(UW, 1, -c1);
    q = rotmat_to_quat(mul(UW, Vt));
    cheirality_pair<NP>(q, t, x1, x2, pos, neg); // ... then (q', -t), (q', t)  (:160-167)
    if (neg)
        emit(q, tn);
    if (pos)
        emit(q, t);
}
template <int NP> PL_HD int motion_from_essential(const Mat3 &E, const Vec3 *x1, const Vec3 *x2, PoseQT *out) {
    int n = 0;
    motion_from_essential_emit<NP>(E, x1, x2, [&](Quat q, Vec3 t) {
        out[n].q = q, out[n].t = t;
        ++n;
    });
    return n;
}

// =============================================================================== 5-point
namespace detail5 {
// monomial tables.  linear: [x y z 1]; quadratic: [x2 xy xz x y2 yz y z2 z 1];
// cubic (Nister): [x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy xz2 xz x yz2 yz y z3 z2 z 1]
// (namespace-scope constexpr tables: with the loops unrolled every look-up folds to a constant on host and device)
constexpr int kQuadIndex[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
constexpr int kCubicIndex[10][4] = {{0, 2, 4, 5},   {2, 3, 8, 9},    {4, 8, 10, 11},  {5, 9, 11, 12},  {3, 1, 6, 7},
                                    {8, 6, 13, 14}, {9, 7, 14, 15}, {10, 13, 16, 17}, {11, 14, 17, 18}, {12, 15, 18, 19}};
PL_HD constexpr int quad_index(int i, int j) { return kQuadIndex[i][j]; }   // linear_i * linear_j (symmetric)
PL_HD constexpr int cubic_index(int q, int l) { return kCubicIndex[q][l]; } // quadratic_q * linear_l
// acc(quadratic) += s * (a(linear) * b(linear)).  Every coefficient of the product is formed on its own first - its
// terms in the order "constant towards x" (descending index) of a, then of b (the oracle skips exact zeros of `a`:
// adding their +-0 products changes no bit of a finite sum, which never is -0) - and then added: the association order of the oracle's polynomial arithmetic (oracle/src/solvers_rel.cc mul / axpy), so that
// both produce the same bits.  (Monomial by monomial: one scalar temporary instead of a whole product in registers.)
PL_HD void mac_lin_lin(double *acc, double s, const double *a, const double *b) {
    PL_UNROLL
    for (int m = 0; m < 10; ++m) {
        double t = 0.0;
        PL_UNROLL
        for (int i = 3; i >= 0; --i)
            PL_UNROLL
            for (int j = 3; j >= 0; --j)
                if (quad_index(i, j) == m)
                    t += a[i] * b[j];
        acc[m] += s * t;
    }
}
// acc(cubic) += a(quadratic) * b(linear), same order
PL_HD void mac_quad_lin(double *acc, const double *a, const double *b) {
    PL_UNROLL
    for (int m = 0; m < 20; ++m) {
        double t = 0.0;
        PL_UNROLL
        for (int q = 9; q >= 0; --q)
            PL_UNROLL
            for (int l = 3; l >= 0; --l)
                if (cubic_index(q, l) == m)
                    t += a[q] * b[l];
        acc[m] += 1.0 * t;
    }
}
} // namespace detail5

// The 5-point solver in three stages (the batched generator runs them as three kernels so that no stage carries the
// state of another; the bare solver entry points call them back to back):
//   rel5_front   bearings -> null-space basis nb[36] + the 3 x 3 polynomial matrix Az[3][13]
//   rel5_poly    Az -> coefficients of the degree-10 determinant      (then sturm_roots_deg10)
//   rel5_essential_at_root   one real root z -> one essential matrix
// All array indices are compile-time constants after unrolling (pivot rows are exchanged by comparison-selected
// swaps): the 10 x 20 elimination stays in registers.
PL_HD void rel5_front(const Vec3 *x1, const Vec3 *x2, double *nb /* 36: nb[b*9 + e], e = 3*col + row of E */,
                      double (*Az)[13]) {
    using namespace detail5;
    double A[45];
    PL_UNROLL
    for (int i = 0; i < 5; ++i) {
        const double a[3] = {x1[i].x, x1[i].y, x1[i].z};
        PL_UNROLL
        for (int j = 0; j < 3; ++j) {
            A[i * 9 + 3 * j + 0] = a[j] * x2[i].x;
            A[i * 9 + 3 * j + 1] = a[j] * x2[i].y;
            A[i * 9 + 3 * j + 2] = a[j] * x2[i].z;
        }
    }
    complement_basis9<5>(A, nb);

    // E(i,j) as linear form l[0..3] over (x,y,z,1)
    double L[3][3][4];
    PL_UNROLL
    for (int i = 0; i < 3; ++i)
        PL_UNROLL
        for (int j = 0; j < 3; ++j)
            PL_UNROLL
            for (int b = 0; b < 4; ++b)
                L[i][j][b] = nb[b * 9 + 3 * j + i];

    double M[10][20];
    PL_UNROLL
    for (int r = 0; r < 10; ++r)
        PL_UNROLL
        for (int c = 0; c < 20; ++c)
            M[r][c] = 0.0;
    // (E E^T - 1/2 tr(E E^T) I) E  -> rows 0..8
    {
        double EEt[3][3][10];
        PL_UNROLL
        for (int i = 0; i < 3; ++i)
            PL_UNROLL
            for (int j = i; j < 3; ++j) {
                PL_UNROLL
                for (int m = 0; m < 10; ++m)
                    EEt[i][j][m] = 0.0;
                PL_UNROLL
                for (int k = 0; k < 3; ++k)
                    mac_lin_lin(EEt[i][j], 1.0, L[i][k], L[j][k]);
            }
        PL_UNROLL
        for (int m = 0; m < 10; ++m) {
            double h = 0.0 + 0.5 * EEt[0][0][m]; // half the trace, summed like the oracle's axpy chain
            h += 0.5 * EEt[1][1][m];
            h += 0.5 * EEt[2][2][m];
            EEt[0][0][m] += -1.0 * h;
            EEt[1][1][m] += -1.0 * h;
            EEt[2][2][m] += -1.0 * h;
        }
        PL_UNROLL
        for (int i = 0; i < 3; ++i)
            PL_UNROLL
            for (int j = 0; j < 3; ++j)
                PL_UNROLL
                for (int k = 0; k < 3; ++k)
                    mac_quad_lin(M[3 * i + j], (i <= k) ? EEt[i][k] : EEt[k][i], L[k][j]);
    }
    // det(E) -> row 9
    {
        double m0[10], m1[10], m2[10];
        PL_UNROLL
        for (int m = 0; m < 10; ++m)
            m0[m] = m1[m] = m2[m] = 0.0;
        mac_lin_lin(m0, 1.0, L[0][1], L[1][2]);
        mac_lin_lin(m0, -1.0, L[0][2], L[1][1]);
        mac_lin_lin(m1, 1.0, L[0][2], L[1][0]);
        mac_lin_lin(m1, -1.0, L[0][0], L[1][2]);
        mac_lin_lin(m2, 1.0, L[0][0], L[1][1]);
        mac_lin_lin(m2, -1.0, L[0][1], L[1][0]);
        mac_quad_lin(M[9], m0, L[2][0]);
        mac_quad_lin(M[9], m1, L[2][1]);
        mac_quad_lin(M[9], m2, L[2][2]);
    }

    // X = M[:, :10]^-1 M[:, 10:]; only rows 4..9 of X are needed.  LU with partial pivoting, in place.
    PL_UNROLL
    for (int k = 0; k < 10; ++k) {
        int piv = k;
        double best = fabs(M[k][k]);
        PL_UNROLL
        for (int i = k + 1; i < 10; ++i)
            if (fabs(M[i][k]) > best) {
                best = fabs(M[i][k]);
                piv = i;
            }
        PL_UNROLL
        for (int i = k + 1; i < 10; ++i) {
            const bool sw = piv == i;
            PL_UNROLL
            for (int j = 0; j < 20; ++j)
                cswap(sw, M[k][j], M[i][j]);
        }
        if (M[k][k] != 0.0) {
            PL_UNROLL
            for (int i = k + 1; i < 10; ++i)
                M[i][k] /= M[k][k];
        }
        PL_UNROLL
        for (int i = k + 1; i < 10; ++i) {
            const double f = M[i][k];
            PL_UNROLL
            for (int j = k + 1; j < 10; ++j)
                M[i][j] -= f * M[k][j];
        }
    }
    PL_UNROLL
    for (int c = 10; c < 20; ++c) {
        PL_UNROLL
        for (int i = 1; i < 10; ++i) {
            double s = M[i][c];
            PL_UNROLL
            for (int j = 0; j < i; ++j)
                s -= M[i][j] * M[j][c];
            M[i][c] = s;
        }
        PL_UNROLL
        for (int i = 9; i >= 4; --i) {
            double s = M[i][c];
            PL_UNROLL
            for (int j = i + 1; j < 10; ++j)
                s -= M[i][j] * M[j][c];
            M[i][c] = s / M[i][i];
        }
    }

    // px_i(z) x + py_i(z) y + pc_i(z) = 0, highest power first
    PL_UNROLL
    for (int i = 0; i < 3; ++i) {
        const double *ev = &M[4 + 2 * i][10], *od = &M[5 + 2 * i][10];
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
}

// degree-10 determinant of the 3 x 3 polynomial matrix by polynomial arithmetic (ascending coefficients)
PL_HD void rel5_poly(const double (*Az)[13], double *c /* 11 */) {
    PL_UNROLL
    for (int k = 0; k <= 10; ++k)
        c[k] = 0.0;
    // term t: row r takes column perm[t][r]; columns: 0 = px (deg 3, Az[.][0..3]), 1 = py (deg 3, Az[.][4..7]),
    // 2 = pc (deg 4, Az[.][8..12])
    PL_UNROLL
    for (int t = 0; t < 6; ++t) {
        const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        const double sgn[6] = {1, -1, -1, 1, 1, -1};
        double prod[11];
        PL_UNROLL
        for (int k = 0; k <= 10; ++k)
            prod[k] = 0.0;
        double p0[5], p1[5], p2[5], tmp[9];
        const int s0 = perm[t][0], s1 = perm[t][1], s2 = perm[t][2];
        const int d0 = (s0 == 2) ? 4 : 3, d1 = (s1 == 2) ? 4 : 3, d2 = (s2 == 2) ? 4 : 3;
        const int o0 = 4 * s0, o1 = 4 * s1, o2 = 4 * s2; // column offsets 0, 4, 8
        PL_UNROLL
        for (int k = 0; k < 5; ++k) {
            p0[k] = (k <= d0) ? Az[0][o0 + d0 - k] : 0.0; // ascending
            p1[k] = (k <= d1) ? Az[1][o1 + d1 - k] : 0.0;
            p2[k] = (k <= d2) ? Az[2][o2 + d2 - k] : 0.0;
        }
        PL_UNROLL
        for (int k = 0; k < 9; ++k)
            tmp[k] = 0.0;
        PL_UNROLL
        for (int i = 0; i < 5; ++i)
            PL_UNROLL
            for (int j = 0; j < 5; ++j)
                if (i <= d0 && j <= d1)
                    tmp[i + j] += p0[i] * p1[j];
        PL_UNROLL
        for (int i = 0; i < 9; ++i)
            PL_UNROLL
            for (int j = 0; j < 5; ++j)
                if (i <= d0 + d1 && j <= d2)
                    prod[i + j] += tmp[i] * p2[j];
        PL_UNROLL
        for (int k = 0; k <= 10; ++k)
            c[k] += sgn[t] * prod[k];
    }
}

// one real root of the determinant -> (x, y) by back substitution -> E = normalised null-space combination
// (two steps, so that a caller can finish the first for all roots before it loads the null-space basis)
PL_HD void rel5_xy_at_root(const double (*Az)[13], double z, double &x, double &y) {
    const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2;
    double B[3][2], b[3];
    PL_UNROLL
    for (int i = 0; i < 3; ++i) {
        B[i][0] = Az[i][0] * z3 + Az[i][1] * z2 + Az[i][2] * z + Az[i][3];
        B[i][1] = Az[i][4] * z3 + Az[i][5] * z2 + Az[i][6] * z + Az[i][7];
        b[i] = Az[i][8] * z4 + Az[i][9] * z3 + Az[i][10] * z2 + Az[i][11] * z + Az[i][12];
    }
    const double dt = B[0][0] * B[1][1] - B[1][0] * B[0][1];
    const double idt = 1.0 / dt;
    double u0 = (B[1][1] * idt) * b[0] + (-B[0][1] * idt) * b[1];
    double u1 = (-B[1][0] * idt) * b[0] + (B[0][0] * idt) * b[1];
    if (fabs(B[2][0] * u0 + B[2][1] * u1 - b[2]) > 1e-6) {
        // least squares over the three rows via the 2x2 normal equations solved by a
        // pivoted Householder QR (reference: colPivHouseholderQr, relpose_5pt.cc:381)
        double Q[3][2] = {{B[0][0], B[0][1]}, {B[1][0], B[1][1]}, {B[2][0], B[2][1]}};
        double rhs[3] = {b[0], b[1], b[2]};
        const double n0 = Q[0][0] * Q[0][0] + Q[1][0] * Q[1][0] + Q[2][0] * Q[2][0];
        const double n1 = Q[0][1] * Q[0][1] + Q[1][1] * Q[1][1] + Q[2][1] * Q[2][1];
        const bool swapc = n1 > n0;
        PL_UNROLL
        for (int i = 0; i < 3; ++i)
            cswap(swapc, Q[i][0], Q[i][1]);
        double Rm[2][2] = {{0, 0}, {0, 0}};
        PL_UNROLL
        for (int k = 0; k < 2; ++k) {
            double tail = 0;
            PL_UNROLL
            for (int i = k + 1; i < 3; ++i)
                tail += Q[i][k] * Q[i][k];
            const double c0 = Q[k][k];
            double beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0)
                beta = -beta;
            double v[3] = {0, 0, 0};
            double tau = 0;
            if (tail > 2.2250738585072014e-308) {
                v[k] = 1.0;
                PL_UNROLL
                for (int i = k + 1; i < 3; ++i)
                    v[i] = Q[i][k] / (c0 - beta);
                tau = (beta - c0) / beta;
            } else {
                beta = c0;
            }
            Rm[k][k] = beta;
            PL_UNROLL
            for (int cc = k + 1; cc < 2; ++cc) {
                double t = 0;
                PL_UNROLL
                for (int i = k; i < 3; ++i)
                    t += v[i] * Q[i][cc];
                PL_UNROLL
                for (int i = k; i < 3; ++i)
                    Q[i][cc] -= tau * v[i] * t;
                Rm[k][cc] = Q[k][cc];
            }
            double t = 0;
            PL_UNROLL
            for (int i = k; i < 3; ++i)
                t += v[i] * rhs[i];
            PL_UNROLL
            for (int i = k; i < 3; ++i)
                rhs[i] -= tau * v[i] * t;
        }
        double w1 = rhs[1] / Rm[1][1];
        double w0 = (rhs[0] - Rm[0][1] * w1) / Rm[0][0];
        cswap(swapc, w0, w1);
        u0 = w0;
        u1 = w1;
    }
    x = -u0, y = -u1;
}
PL_HD void rel5_essential_from_xyz(const double *nb, double x, double y, double z, Mat3 &Eout) {
    const double inv_norm = 1.0 / sqrt(x * x + y * y + z * z + 1.0);
    PL_UNROLL
    for (int j = 0; j < 3; ++j)
        PL_UNROLL
        for (int i = 0; i < 3; ++i) {
            const int e = 3 * j + i;
            Eout.m[3 * i + j] = (nb[0 * 9 + e] * x + nb[1 * 9 + e] * y + nb[2 * 9 + e] * z + nb[3 * 9 + e]) * inv_norm;
        }
}
PL_HD void rel5_essential_at_root(const double *nb, const double (*Az)[13], double z, Mat3 &Eout) {
    double x, y;
    rel5_xy_at_root(Az, z, x, y);
    rel5_essential_from_xyz(nb, x, y, z, Eout);
}

// Returns the number of essential matrices (<= 10); E[i] row-major.
PL_HD int essential_5pt(const Vec3 *x1, const Vec3 *x2, Mat3 *Eout) {
    double nb[36], Az[3][13], c[11], roots[10];
    rel5_front(x1, x2, nb, Az);
    rel5_poly(Az, c);
    const int nroots = sturm_roots_deg10(c, roots);
    for (int s = 0; s < nroots; ++s)
        rel5_essential_at_root(nb, Az, roots[s], Eout[s]);
    return nroots;
}

// 5-point relative pose: up to 40 model records (pose + E=[t]xR(q)).  Returns the number of solutions; only
// the first `max_out` are written (the caller detects the overflow and retries with room for 40).
PL_HD int relpose_5pt_records(const Vec3 *x1, const Vec3 *x2, double *rec, int max_out = 40) {
    Mat3 E[10];
    const int ne = essential_5pt(x1, x2, E);
    int n = 0;
    for (int i = 0; i < ne; ++i) {
        PoseQT cand[4];
        const int nc = motion_from_essential<5>(E[i], x1, x2, cand);
        for (int k = 0; k < nc; ++k) {
            if (n < max_out)
                store_pose_model_q(rec + n * kModelStride, cand[k].q, cand[k].t, true);
            ++n;
        }
    }
    return n;
}

// =============================================================================== 7-point
PL_HD bool real_focal_check(const Mat3 &Fm) { // utils.cc:646-671
    const double *F = Fm.m;
#define FF(i, j) F[3 * (i) + (j)]
    float den, num;
    den = FF(0, 0) * FF(0, 1) * FF(2, 0) * FF(2, 2) - FF(0, 0) * FF(0, 2) * FF(2, 0) * FF(2, 1) +
          FF(0, 1) * FF(0, 1) * FF(2, 1) * FF(2, 2) - FF(0, 1) * FF(0, 2) * FF(2, 1) * FF(2, 1) +
          FF(1, 0) * FF(1, 1) * FF(2, 0) * FF(2, 2) - FF(1, 0) * FF(1, 2) * FF(2, 0) * FF(2, 1) +
          FF(1, 1) * FF(1, 1) * FF(2, 1) * FF(2, 2) - FF(1, 1) * FF(1, 2) * FF(2, 1) * FF(2, 1);
    num = -FF(2, 2) * (FF(0, 1) * FF(0, 2) * FF(2, 2) - FF(0, 2) * FF(0, 2) * FF(2, 1) +
                       FF(1, 1) * FF(1, 2) * FF(2, 2) - FF(1, 2) * FF(1, 2) * FF(2, 1));
    if (num * den < 0)
        return false;
    den = FF(0, 0) * FF(1, 0) * FF(0, 2) * FF(2, 2) - FF(0, 0) * FF(2, 0) * FF(0, 2) * FF(1, 2) +
          FF(1, 0) * FF(1, 0) * FF(1, 2) * FF(2, 2) - FF(1, 0) * FF(2, 0) * FF(1, 2) * FF(1, 2) +
          FF(0, 1) * FF(1, 1) * FF(0, 2) * FF(2, 2) - FF(0, 1) * FF(2, 1) * FF(0, 2) * FF(1, 2) +
          FF(1, 1) * FF(1, 1) * FF(1, 2) * FF(2, 2) - FF(1, 1) * FF(2, 1) * FF(1, 2) * FF(1, 2);
    num = -FF(2, 2) * (FF(1, 0) * FF(2, 0) * FF(2, 2) - FF(2, 0) * FF(2, 0) * FF(1, 2) +
                       FF(1, 1) * FF(2, 1) * FF(2, 2) - FF(2, 1) * FF(2, 1) * FF(1, 2));
#undef FF
    if (num * den < 0)
        return false;
    return true;
}

PL_HD int relpose_7pt(const Vec3 *x1, const Vec3 *x2, Mat3 *Fout) {
    double A[63];
    for (int i = 0; i < 7; ++i) {
        const double a[3] = {x1[i].x, x1[i].y, x1[i].z};
        for (int j = 0; j < 3; ++j) {
            A[i * 9 + 3 * j + 0] = a[j] * x2[i].x;
            A[i * 9 + 3 * j + 1] = a[j] * x2[i].y;
            A[i * 9 + 3 * j + 2] = a[j] * x2[i].z;
        }
    }
    double nb[18];
    complement_basis9_indexed<7>(A, nb);
    const double *n0 = nb, *n1 = nb + 9;
    // det(x F0 + F1), entries (i,j) <-> vec index 3j+i.  relpose_7pt.cc:22-37 writes the 48 monomials in the
    // lexicographic order of a symbolic expansion (factor 1 from vector entries 0..2, factor 2 from 3..5, factor 3
    // from 6..8, null vector 0 - the x part - before null vector 1), products left to right, coefficients summed
    // in that order.  The last bits of c3..c0 reach the roots, F, and through the rounding-level det(F) the SIGN
    // of every refined F (pl_svd3.h), so the order is kept: this loop nest enumerates it.
    double c[4] = {0, 0, 0, 0};
    PL_UNROLL
    for (int ra = 0; ra < 3; ++ra) {
    PL_UNROLL
        for (int ka = 0; ka < 2; ++ka) {
    PL_UNROLL
            for (int rb = 0; rb < 3; ++rb) {
                if (rb == ra)
                    continue;
                const int rc = 3 - ra - rb;
                const bool even = (ra == 0 && rb == 1) || (ra == 1 && rb == 2) || (ra == 2 && rb == 0);
    PL_UNROLL
                for (int kb = 0; kb < 2; ++kb) {
    PL_UNROLL
                    for (int kc = 0; kc < 2; ++kc) {
                        const double t = nb[9 * ka + ra] * nb[9 * kb + 3 + rb] * nb[9 * kc + 6 + rc];
                        const int deg = 3 - (ka + kb + kc);
                        c[deg] = even ? c[deg] + t : c[deg] - t;
                    }
                }
            }
        }
    }
    double roots[3];
    int nr;
    if (fabs(c[3]) < 1e-14) {
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
        nn = sqrt(nn);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i)
                Fout[s].m[3 * i + j] = f[3 * j + i] / nn;
    }
    return nr;
}

// <= 3 records; with `rfc` the real-focal-length filter of relative_pose.cc:393-398 is applied.
// The reference erases failing models back to front, which preserves the order of the survivors.
PL_HD int relpose_7pt_records(const Vec3 *x1, const Vec3 *x2, double *rec, bool rfc) {
    Mat3 F[3];
    const int nf = relpose_7pt(x1, x2, F);
    int n = 0;
    for (int i = 0; i < nf; ++i) {
        if (rfc && !real_focal_check(F[i]))
            continue;
        store_matrix_model(rec + (n++) * kModelStride, F[i]);
    }
    return n;
}

} // namespace pl
