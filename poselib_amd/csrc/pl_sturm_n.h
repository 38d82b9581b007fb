// poselib_amd - real roots of a polynomial of degree N by Sturm bisection, the reference's bisect_sturm<N> (misc/sturm.h:233-274:
// monic normalisation, chain :47-84, Cauchy bound :144-150, isolation by bisection :210-231 in recursion order - depth first, left
// half first -, Ridders' method then Newton on an isolating interval :153-208).  The degree-10 instance of the 5-point solver is
// pl_solver_rel.h's (tuned for the batched generator: flat isolation, leaves ranked afterwards); this is the plain form for any
// N <= 15 with the tolerance as a parameter - the shared-focal six-point solver calls bisect_sturm<15>(p, roots, 1e-12) on the
// characteristic polynomial of its action matrix (relpose_6pt_focal.cc:1069-1076).  One lane runs it; same operations, same order
// as oracle/src/solvers_rel.cc sturm_real_roots (tests/hostmath compares the two bit for bit).  On the device lane 0 of a sample's
// wavefront isolates the roots and one lane per leaf polishes them (sfocal.hip).
#pragma once
#include "pl_math.h"

namespace pl {

// The chain and the bisection's stack live in a caller-provided workspace of kSturmNWork(N) doubles (on the device: LDS of the
// sample's wavefront; on the host: a local array).
constexpr int kSturmNWork(int N) { return (N + 1) + N + 3 * (N - 1) + 3 + 3 * (N + 1) + 3 * (N + 2); }

template <int N> struct SturmN { // views into the workspace
    double *f;  // [N + 1] monic
    double *fp; // [N] derivative / N: monic of degree N - 1
    double *q0, *q1, *c; // [N - 1] each
    double *tail;        // tail0, tail1, last
    double *buf;         // [3][N + 1] remainder sequence under construction
    double *sa, *sb, *si; // [N + 2] each: deferred halves of the bisection (si: the packed counts, exact in a double)
    PL_HD explicit SturmN(double *ws) {
        f = ws, fp = f + (N + 1), q0 = fp + N, q1 = q0 + (N - 1), c = q1 + (N - 1), tail = c + (N - 1), buf = tail + 3;
        sa = buf + 3 * (N + 1), sb = sa + (N + 2), si = sb + (N + 2);
    }
};

template <int N> PL_HD double sturm_n_horner(const double *p, int deg, double x) {
    double v = x + p[deg - 1];
    for (int i = deg - 2; i >= 0; --i)
        v = x * v + p[i];
    return v;
}

template <int N> PL_HD void sturm_n_build(const SturmN<N> &S) { // sturm.h:47-84
    double *buf = S.buf;
    constexpr int W = N + 1;
    int hi = 0, lo = 1, rem = 2;
    for (int i = 0; i <= N; ++i)
        buf[i] = S.f[i];
    for (int i = 0; i < N; ++i)
        buf[W + i] = S.fp[i];
    for (int i = 0; i < N - 1; ++i) {
        const int dh = N - i, dl = N - 1 - i;
        const double a1 = buf[hi * W + dh] * buf[lo * W + dl];
        const double a0 = buf[hi * W + dh - 1] * buf[lo * W + dl] - buf[hi * W + dh] * buf[lo * W + dl - 1];
        buf[rem * W] = buf[hi * W] - a0 * buf[lo * W];
        for (int j = 1; j < dl; ++j)
            buf[rem * W + j] = buf[hi * W + j] - a1 * buf[lo * W + j - 1] - a0 * buf[lo * W + j];
        const double scale = -fabs(buf[rem * W + dl - 1]);
        const double inv = 1.0 / scale;
        for (int j = 0; j < dl; ++j)
            buf[rem * W + j] = buf[rem * W + j] * inv;
        S.q0[i] = a0;
        S.q1[i] = a1;
        S.c[i] = scale;
        const int t = hi;
        hi = lo;
        lo = rem;
        rem = t;
    }
    S.tail[0] = buf[hi * W];
    S.tail[1] = buf[hi * W + 1];
    S.tail[2] = buf[lo * W];
}

template <int N> PL_HD int sturm_n_variations(const SturmN<N> &S, double x) { // sturm.h:98-112
    double up2 = S.tail[2];
    double up1 = S.tail[0] + x * S.tail[1];
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

// The real roots of coef[0 .. N] (coef[N] the leading coefficient) in the reference's order, in two phases so that the second
// can run one leaf per lane on the device:
//   sturm_n_isolate    monic f, f' / N, the chain, the Cauchy bound, the bisection in recursion order (depth first, left half
//                      first): its LEAVES - intervals narrower than tol (the reference reports their right end as a root whatever
//                      the counts say) and intervals with exactly one sign variation - in the order the recursion reaches them
//   sturm_n_leaf_root  one leaf -> its root (Ridders + Newton on an isolating interval: nothing when the end points do not
//                      bracket a sign change)
// The roots of the leaves, in leaf order, cut off after N, are what the recursion emits (a leaf's polish does not touch the
// traversal).  ws: kSturmNWork(N) doubles; leaves: 2 * kSturmNLeaves(N) doubles (a, b per leaf); tiny: bit i = leaf i is narrow.
constexpr int kSturmNLeaves(int N) { return 2 * N; }
template <int N> PL_HD int sturm_n_isolate(const double *coef, double tol, double *ws, double *leaves, unsigned &tiny) {
    static_assert(N <= 15, "the sign-variation counts travel in 4 bits");
    tiny = 0;
    if (coef[N] == 0.0)
        return 0;
    const SturmN<N> S(ws);
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
    // variation, or it is narrower than tol), so at most N + 1 deferred halves are alive
    constexpr int kCap = N + 2, kLeaves = kSturmNLeaves(N);
    double a = -bound, b = bound;
    int sa = sa0, sb = sb0, depth = 0, sp = 0, nl = 0;
    for (;;) {
        bool descend = false;
        if (depth <= 300) { // MAX_STURM_RECURSION_DEPTH_LIMIT
            if (b - a < tol) {
                if (nl < kLeaves) {
                    leaves[2 * nl] = a, leaves[2 * nl + 1] = b;
                    tiny |= 1u << nl;
                    ++nl;
                }
            } else {
                const int k = sa - sb;
                if (k > 1) {
                    const double mid = (a + b) * 0.5;
                    const int sm = sturm_n_variations(S, mid);
                    if ((sm - sb >= 1 || b - mid < tol) && sp < kCap) {
                        S.sa[sp] = mid, S.sb[sp] = b, S.si[sp] = (double)((unsigned)sm | ((unsigned)sb << 4) | ((unsigned)(depth + 1) << 8));
                        ++sp;
                    }
                    b = mid;
                    sb = sm;
                    depth += 1;
                    descend = true;
                } else if (k == 1 && nl < kLeaves) {
                    leaves[2 * nl] = a, leaves[2 * nl + 1] = b;
                    ++nl;
                }
            }
        }
        if (descend)
            continue;
        if (sp == 0)
            break;
        --sp;
        a = S.sa[sp], b = S.sb[sp];
        const unsigned info = (unsigned)S.si[sp];
        sa = (int)(info & 0xfu);
        sb = (int)((info >> 4) & 0xfu);
        depth = (int)(info >> 8);
    }
    return nl;
}
// (ws: the workspace sturm_n_isolate left behind - the monic polynomial and its derivative are read from it)
template <int N> PL_HD int sturm_n_leaf_root(const double *ws, double la, double lb, bool is_tiny, double tol, double *root) {
    if (is_tiny) {
        *root = lb;
        return 1;
    }
    const SturmN<N> S(const_cast<double *>(ws));
    int n = 0;
    sturm_n_polish(S, la, lb, root, n, tol);
    return n;
}
template <int N> PL_HD int sturm_n_roots(const double *coef, double *roots, double tol, double *ws) {
    double leaves[2 * kSturmNLeaves(N)];
    unsigned tiny;
    const int nl = sturm_n_isolate<N>(coef, tol, ws, leaves, tiny);
    int n = 0;
    for (int i = 0; i < nl && n < N; ++i)
        n += sturm_n_leaf_root<N>(ws, leaves[2 * i], leaves[2 * i + 1], (tiny >> i) & 1u, tol, roots + n);
    return n;
}
template <int N> PL_HD int sturm_n_roots(const double *coef, double *roots, double tol) {
    double ws[kSturmNWork(N)];
    return sturm_n_roots<N>(coef, roots, tol, ws);
}

#ifdef __HIPCC__
#ifndef PL_WAVE_SYNC
#define PL_WAVE_SYNC()                                                                                                 \
    do {                                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                                         \
        __builtin_amdgcn_wave_barrier();                                                                               \
    } while (0)
#endif
// sturm_n_isolate by ONE WAVEFRONT.  Lane 0 builds the chain (serial: ~700 dependent operations); the bisection then runs LEVEL BY LEVEL
// with one lane per live interval: a visit of the recursion depends on its interval alone (narrow -> leaf; more than one sign
// variation -> bisect, the left half always, the right half when it holds a sign variation or is narrow; exactly one -> leaf; none ->
// nothing), so the intervals of a level are visited together - one evaluation of the chain per level and lane instead of one per
// visit, 10 - 25 levels instead of 50 - 100 serial evaluations.  The leaves are disjoint intervals: their rank by left end is the
// recursion's order (depth first, left half first).  Same midpoints, same counts, same leaves as the serial routine; a polynomial
// with a non-finite bound (garbage input) takes the serial routine.  ws / leaves as for sturm_n_isolate, plus kSturmNWaveWork(N)
// doubles of LDS for the level exchange.
constexpr int kSturmNWaveWork(int N) { return 4 * 64 + 3 * 64; } // next level's intervals (a, b, counts, -) | leaves as found (a, b, narrow)
template <int N>
__device__ __forceinline__ int sturm_n_isolate_wave(const double *coef, double tol, double *ws, double *leaves, unsigned &tiny, double *xw, int lane) {
    const SturmN<N> S(ws);
    double *na = xw, *nb = xw + 64, *ni = xw + 128, *ua = xw + 256, *ub = xw + 320, *ut = xw + 384;
    int state = 0; // lane 0: 0 = no roots, 1 = level-wise bisection, 2 = done serially
    double bound = 0;
    int sa0 = 0, sb0 = 0, nl_serial = 0;
    tiny = 0;
    if (lane == 0) {
        if (coef[N] != 0.0) {
            const double lead_inv = 1.0 / coef[N];
            for (int i = 0; i < N; ++i)
                S.f[i] = coef[i] * lead_inv;
            S.f[N] = 1.0;
            for (int i = 0; i < N - 1; ++i)
                S.fp[i] = S.f[i + 1] * ((i + 1) / (double)N);
            S.fp[N - 1] = 1.0;
            sturm_n_build(S);
            for (int i = 0; i < N; ++i)
                bound = fmax(bound, fabs(S.f[i]));
            bound = 1.0 + bound;
            if (!isfinite(bound)) {
                nl_serial = sturm_n_isolate<N>(coef, tol, ws, leaves, tiny);
                state = 2;
            } else {
                sa0 = sturm_n_variations(S, -bound), sb0 = sturm_n_variations(S, bound);
                state = (sa0 - sb0 != 0) ? 1 : 0;
            }
        }
        na[0] = bound, ni[0] = (double)state, ni[1] = (double)sa0, ni[2] = (double)sb0, ni[3] = (double)nl_serial, ni[4] = (double)tiny;
    }
    PL_WAVE_SYNC();
    state = (int)ni[0];
    if (state == 0)
        return 0;
    if (state == 2) {
        tiny = (unsigned)ni[4];
        return (int)ni[3];
    }
    bound = na[0], sa0 = (int)ni[1], sb0 = (int)ni[2];
    PL_WAVE_SYNC();
    // live interval of this lane
    bool alive = lane == 0;
    double a = -bound, b = bound;
    int sa = sa0, sb = sb0;
    int nfound = 0; // leaves found so far (uniform)
    for (int depth = 0; depth <= 301; ++depth) { // (depth > 300: the recursion returns without a visit)
        const bool visit = alive && depth <= 300;
        const bool narrow = visit && (b - a < tol);
        const int k = sa - sb;
        const bool leaf = visit && (narrow || k == 1);
        const bool split = visit && !narrow && k > 1;
        double mid = 0;
        int sm = 0;
        if (split) {
            mid = (a + b) * 0.5;
            sm = sturm_n_variations(S, mid);
        }
        const bool keep_right = split && (sm - sb >= 1 || b - mid < tol);
        const uint64_t lmask = __builtin_amdgcn_ballot_w64(leaf);
        if (leaf) {
            const uint32_t at = (uint32_t)nfound + __builtin_amdgcn_mbcnt_hi((uint32_t)(lmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lmask, 0u));
            if (at < 64u)
                ua[at] = a, ub[at] = b, ut[at] = narrow ? 1.0 : 0.0;
        }
        nfound += (int)__popcll(lmask);
        // children of the level: a splitting lane contributes its left half and, when kept, its right half
        const uint64_t smask = __builtin_amdgcn_ballot_w64(split), rmask = __builtin_amdgcn_ballot_w64(keep_right);
        if (!smask)
            break;
        if (split) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(smask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smask, 0u)) +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rmask, 0u));
            if (below < 64u)
                na[below] = a, nb[below] = mid, ni[below] = (double)((unsigned)sa | ((unsigned)sm << 4));
            if (keep_right && below + 1u < 64u)
                na[below + 1] = mid, nb[below + 1] = b, ni[below + 1] = (double)((unsigned)sm | ((unsigned)sb << 4));
        }
        const int total = (int)__popcll(smask) + (int)__popcll(rmask);
        PL_WAVE_SYNC();
        alive = lane < total && lane < 64;
        if (alive) {
            a = na[lane], b = nb[lane];
            const unsigned info = (unsigned)ni[lane];
            sa = (int)(info & 0xfu), sb = (int)(info >> 4);
        }
        PL_WAVE_SYNC();
    }
    PL_WAVE_SYNC();
    // the recursion's order: rank by left end (disjoint intervals); the first kSturmNLeaves(N) are kept
    const int nf = nfound < 64 ? nfound : 64;
    unsigned mine_tiny = 0;
    if (lane < nf) {
        const double la = ua[lane];
        int rank = 0;
        for (int j = 0; j < nf; ++j) // (equal left ends - a midpoint that rounds onto its interval's end - in the order found: by depth)
            rank += (ua[j] < la || (ua[j] == la && j < lane)) ? 1 : 0;
        if (rank < kSturmNLeaves(N)) {
            leaves[2 * rank] = la, leaves[2 * rank + 1] = ub[lane];
            mine_tiny = ut[lane] != 0.0 ? (1u << rank) : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mine_tiny |= (unsigned)__shfl_xor((int)mine_tiny, off, 64);
    tiny = mine_tiny;
    PL_WAVE_SYNC();
    return nf < kSturmNLeaves(N) ? nf : kSturmNLeaves(N);
}
#endif // __HIPCC__

} // namespace pl
