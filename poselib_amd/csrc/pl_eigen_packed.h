// poselib_amd - real eigenvalues of FOUR small matrices by one wavefront (round 5): pl_real_eigenvalues / pl_balance_pow2 of
// pl_solver_p35pf.h / pl_solver_6ptf.h, 16 lanes per matrix.
//
// pl_eigen_wave.h gives a wavefront ONE matrix: the three inner loops of the algorithm run over <= 15 lanes and every scalar of the
// iteration (shifts, reflectors, the deflation scans) is computed by all 64 lanes alike - 41 % / 63 % of the focal estimators' solver
// kernels (profiles/r05_focal_batch.md), whose vector ALUs are 70 % busy with such instructions.  Here a wavefront works on four
// matrices at once: lane = 16 x matrix + (column | row), the scalars of a matrix live in the 16 lanes of its group, the control flow
// is the UNION of the four iterations - every loop of the serial routine becomes a wave-uniform loop over the widest range any of
// the four needs, a matrix takes part in a step under a predicate.  A matrix element still sees exactly the operations of the serial
// routine in the serial routine's order (predicates only switch whole steps off), so the eigenvalues are the same bits.
//
// The routines are written once, over a context X that supplies the lane-parallel primitives:
//   X::A(i, j), hv(i), wr(i), wi(i), out(i)   the group's matrix (row-major) and workspace
//   X::lanes(lo, hi, pred, f)                 f(i) for i in [lo, hi] if pred - one lane per i on the device, a loop on the host
//   X::lane0(pred, f)                         f() once per matrix if pred
//   X::sync()                                 orders the LDS accesses of the wavefront's lanes
//   X::any(pred)                              pred of ANY matrix of the wavefront
//   X::umax(v) / umin(v)                      maximum / minimum of a per-matrix integer over the wavefront's matrices
//   X::div3 / div5                            the independent divisions of a reflector, one quotient per lane
// EigFlatHost (one matrix, plain loops) is the form tests/hostmath runs against the serial routines - the transformation of the
// control flow is checked on the host, bit for bit (tests/test_hostmath_vs_oracle.py); EigWave4 is the device form.
#pragma once
#include "pl_math.h"
#include <cstring>

namespace pl {

template <int n, class X> PL_HD void pl_balance_pow2_packed(X &cx, bool act) {
    bool done = !act; // (per matrix)
    for (int sweep = 0; sweep < 64 && cx.any(!done); ++sweep) {
        const bool in_sweep = !done;
        done = true;
        for (int i = 0; i < n; ++i) {
            double c = 0, r = 0;
            for (int j = 0; j < n; ++j)
                if (j != i) {
                    c += fabs(cx.A(j, i));
                    r += fabs(cx.A(i, j));
                }
            bool apply = in_sweep && !(c == 0 || r == 0);
            double g = r / 2.0, f = 1.0;
            const double s = c + r;
            if (apply) {
                while (c < g) {
                    f *= 2.0;
                    c *= 4.0;
                }
                g = r * 2.0;
                while (c >= g) {
                    f /= 2.0;
                    c /= 4.0;
                }
            }
            apply = apply && (c + r) / f < 0.95 * s;
            if (cx.any(apply)) {
                if (apply)
                    done = false;
                g = 1.0 / f;
                cx.sync();
                cx.lanes(0, n - 1, apply, [&](int j) { cx.A(i, j) *= g; });
                cx.sync();
                cx.lanes(0, n - 1, apply, [&](int j) { cx.A(j, i) *= f; });
                cx.sync();
            }
        }
        if (!in_sweep)
            done = true;
    }
}

// returns the number of eigenvalues that count as real (ascending in out(0 ...)); act: the group holds a matrix
template <int n, class X> PL_HD int pl_real_eigenvalues_packed(X &cx, bool act, double tol) {
    auto cl = [](int i) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }; // (a step that is switched off may form any index)
#define PL_PA(i, j) cx.A(cl(i), cl(j))
    cx.sync();
    // ---- Householder reduction to Hessenberg form
    for (int k = 0; k + 2 < n; ++k) {
        double tail = 0;
        for (int r = k + 2; r < n; ++r)
            tail += PL_PA(r, k) * PL_PA(r, k);
        const bool go = act && !(tail <= 1e-300);
        if (!cx.any(go))
            continue;
        const double c0 = PL_PA(k + 1, k);
        double beta = sqrt(c0 * c0 + tail);
        if (c0 >= 0)
            beta = -beta;
        const double den = c0 - beta;
        cx.lanes(0, n - 1, go, [&](int i) { cx.hv(i) = i <= k ? 0.0 : i == k + 1 ? 1.0 : PL_PA(i, k) / den; });
        const double tau = (beta - c0) / beta;
        cx.sync();
        cx.lanes(0, n - 1, go, [&](int c) {
            double t = 0;
            for (int r = k + 1; r < n; ++r)
                t += cx.hv(r) * PL_PA(r, c);
            for (int r = k + 1; r < n; ++r)
                PL_PA(r, c) -= tau * cx.hv(r) * t;
        });
        cx.sync();
        cx.lanes(0, n - 1, go, [&](int r) {
            double t = 0;
            for (int c = k + 1; c < n; ++c)
                t += PL_PA(r, c) * cx.hv(c);
            for (int c = k + 1; c < n; ++c)
                PL_PA(r, c) -= tau * t * cx.hv(c);
        });
        cx.sync();
        cx.lanes(k + 1, n - 1, go, [&](int i) { PL_PA(i, k) = i == k + 1 ? beta : 0.0; });
        cx.sync();
    }
    cx.lanes(0, n - 1, act, [&](int i) { cx.wr(i) = 0.0, cx.wi(i) = 0.0; });
    const double eps = 2.220446049250313e-16;
    double anorm = 0;
    for (int i = 0; i < n; ++i)
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < n; ++j)
            anorm += fabs(PL_PA(i, j));
    // ---- Francis double-shift QR iteration: one pass of the loop below = one pass of the serial routine's do-loop, per matrix
    int nn = n - 1, its = 0;
    bool failed = false, running = act;
    double t = 0, p = 0, q = 0, r = 0, w = 0, x = 0, y = 0, z = 0;
    while (cx.any(running)) {
        cx.sync();
        // the lowest small subdiagonal element at or below nn
        int l = 0;
        {
            bool found = false;
            const int jhi = cx.umax(running ? nn : 0);
            for (int j = jhi; j >= 1; --j) {
                double s = fabs(PL_PA(j - 1, j - 1)) + fabs(PL_PA(j, j));
                if (s == 0)
                    s = anorm;
                const bool hit = running && !found && j <= nn && fabs(PL_PA(j, j - 1)) <= eps * s;
                if (cx.any(hit)) {
                    cx.lane0(hit, [&] { PL_PA(j, j - 1) = 0; });
                    if (hit)
                        l = j, found = true;
                }
            }
        }
        cx.sync();
        if (running)
            x = PL_PA(nn, nn);
        const bool one = running && l == nn, two = running && l == nn - 1, sweep = running && !one && !two;
        if (cx.any(one))
            cx.lane0(one, [&] { cx.wr(cl(nn)) = x + t, cx.wi(cl(nn)) = 0; });
        if (cx.any(two || sweep)) {
            const double y1 = PL_PA(nn - 1, nn - 1), w1 = PL_PA(nn, nn - 1) * PL_PA(nn - 1, nn);
            if (two || sweep)
                y = y1, w = w1;
        }
        if (cx.any(two)) {
            const double pb = 0.5 * (y - x);
            const double qb = pb * pb + w;
            const double zs = sqrt(fabs(qb));
            const double xb = x + t;
            const double zb = pb + (pb >= 0 ? fabs(zs) : -fabs(zs));
            double w2 = xb + zb;
            if (zb != 0)
                w2 = xb - w / zb;
            cx.lane0(two, [&] {
                if (qb >= 0) {
                    cx.wr(cl(nn - 1)) = xb + zb, cx.wr(cl(nn)) = w2;
                    cx.wi(cl(nn - 1)) = 0, cx.wi(cl(nn)) = 0;
                } else {
                    cx.wr(cl(nn - 1)) = xb + pb, cx.wr(cl(nn)) = xb + pb;
                    cx.wi(cl(nn - 1)) = zs, cx.wi(cl(nn)) = -zs;
                }
            });
        }
        if (cx.any(sweep)) {
            if (sweep && its == 60)
                failed = true;
            const bool go = sweep && !failed;
            const bool exc = go && (its == 10 || its == 20); // exceptional shift
            if (cx.any(exc)) {
                if (exc)
                    t += x;
                cx.sync();
                cx.lanes(0, nn, exc, [&](int i) { PL_PA(i, i) -= x; });
                cx.sync();
                const double s = fabs(PL_PA(nn, nn - 1)) + fabs(PL_PA(nn - 1, nn - 2));
                if (exc) {
                    y = x = 0.75 * s;
                    w = -0.4375 * s * s;
                }
            }
            if (go)
                ++its;
            // two consecutive small subdiagonal elements
            int m = 0;
            {
                bool found = false;
                const int mhi = cx.umax(go ? nn - 2 : 0), mlo = cx.umin(go ? l : n);
                for (int mm = mhi; mm >= mlo; --mm) {
                    const bool on = go && !found && mm <= nn - 2 && mm >= l;
                    const double z1 = PL_PA(mm, mm);
                    const double r1 = x - z1, s1 = y - z1;
                    const double p1 = (r1 * s1 - w) / PL_PA(mm + 1, mm) + PL_PA(mm, mm + 1);
                    const double q1 = PL_PA(mm + 1, mm + 1) - z1 - r1 - s1;
                    const double r2 = PL_PA(mm + 2, mm + 1);
                    const double s2 = fabs(p1) + fabs(q1) + fabs(r2);
                    double pq, qq, rq;
                    cx.div3(p1, q1, r2, s2, pq, qq, rq);
                    const double u = fabs(PL_PA(mm, mm - 1)) * (fabs(qq) + fabs(rq));
                    const double v = fabs(pq) * (fabs(PL_PA(mm - 1, mm - 1)) + fabs(z1) + fabs(PL_PA(mm + 1, mm + 1)));
                    if (on) {
                        z = z1, p = pq, q = qq, r = rq;
                        if (mm == l || u <= eps * v)
                            m = mm, found = true;
                    }
                }
            }
            cx.sync();
            cx.lanes(m + 2, nn, go, [&](int i) {
                PL_PA(i, i - 2) = 0;
                if (i != m + 2)
                    PL_PA(i, i - 3) = 0;
            });
            cx.sync();
            // the double QR step on rows l .. nn and columns m .. nn
            const int klo = cx.umin(go ? m : n), khi = cx.umax(go ? nn - 1 : -1);
            for (int k = klo; k <= khi; ++k) {
                const bool on = go && k >= m && k <= nn - 1;
                if (cx.any(on && k != m)) {
                    const double p1 = PL_PA(k, k - 1), q1 = PL_PA(k + 1, k - 1), r1 = (k != nn - 1) ? PL_PA(k + 2, k - 1) : 0.0;
                    const double x1 = fabs(p1) + fabs(q1) + fabs(r1);
                    double pq, qq, rq;
                    cx.div3(p1, q1, r1, x1, pq, qq, rq);
                    if (on && k != m) {
                        x = x1;
                        if (x1 != 0)
                            p = pq, q = qq, r = rq;
                        else
                            p = p1, q = q1, r = r1;
                    }
                }
                const double sq = sqrt(p * p + q * q + r * r);
                const double s = p >= 0 ? sq : -sq;
                const bool upd = on && s != 0;
                if (cx.any(upd)) {
                    cx.sync();
                    cx.lane0(upd, [&] {
                        if (k == m) {
                            if (l != m)
                                PL_PA(k, k - 1) = -PL_PA(k, k - 1);
                        } else {
                            PL_PA(k, k - 1) = -s * x;
                        }
                    });
                    const double p2 = p + s;
                    double x2, y2, z2, q2, r2;
                    cx.div5(p2, q, r, s, x2, y2, z2, q2, r2); // x = p / s, y = q / s, z = r / s; q /= p, r /= p
                    if (upd)
                        p = p2, x = x2, y = y2, z = z2, q = q2, r = r2;
                    cx.sync();
                    cx.lanes(k, nn, upd, [&](int j) { // the reflector on rows k .. k + 2: column j
                        double pp = PL_PA(k, j) + q * PL_PA(k + 1, j);
                        if (k != nn - 1) {
                            pp += r * PL_PA(k + 2, j);
                            PL_PA(k + 2, j) -= pp * z;
                        }
                        PL_PA(k + 1, j) -= pp * y;
                        PL_PA(k, j) -= pp * x;
                    });
                    cx.sync();
                    const int mmin = nn < k + 3 ? nn : k + 3;
                    cx.lanes(l, mmin, upd, [&](int i) { // on columns k .. k + 2: row i
                        double pp = x * PL_PA(i, k) + y * PL_PA(i, k + 1);
                        if (k != nn - 1) {
                            pp += z * PL_PA(i, k + 2);
                            PL_PA(i, k + 2) -= pp * r;
                        }
                        PL_PA(i, k + 1) -= pp * q;
                        PL_PA(i, k) -= pp;
                    });
                    cx.sync();
                }
            }
        }
        if (one)
            nn -= 1, its = 0;
        if (two)
            nn -= 2, its = 0;
        running = running && !failed && nn >= 0;
    }
    cx.sync();
    int m = 0; // (every lane of the group walks the same list; one lane writes it)
    for (int i = 0; i < n; ++i) {
        const bool real = act && !failed && fabs(cx.wi(i)) <= tol * (1.0 + fabs(cx.wr(i)));
        if (cx.any(real)) { // insertion into the ascending list
            const double v = cx.wr(i);
            cx.sync();
            cx.lane0(real, [&] {
                int j = m;
                while (j > 0 && cx.out(j - 1) > v) {
                    cx.out(j) = cx.out(j - 1);
                    --j;
                }
                cx.out(j) = v;
            });
            if (real)
                ++m;
            cx.sync();
        }
    }
#undef PL_PA
    return m;
}

// ---- host form: ONE matrix, the lane loops as loops (tests/hostmath: bit for bit against the serial routines) ----
template <int n> struct EigFlatHost {
    double *a; // n * n matrix, then hv | wr | wi | out (n doubles each)
    double &A(int i, int j) { return a[i * n + j]; }
    double &hv(int i) { return a[n * n + i]; }
    double &wr(int i) { return a[n * n + n + i]; }
    double &wi(int i) { return a[n * n + 2 * n + i]; }
    double &out(int i) { return a[n * n + 3 * n + i]; }
    template <class F> void lanes(int lo, int hi, bool pred, F f) {
        if (pred)
            for (int i = lo < 0 ? 0 : lo; i <= hi && i < n; ++i)
                f(i);
    }
    template <class F> void lane0(bool pred, F f) {
        if (pred)
            f();
    }
    void sync() {}
    bool any(bool b) { return b; }
    int umax(int v) { return v; }
    int umin(int v) { return v; }
    void div3(double a0, double a1, double a2, double d, double &q0, double &q1, double &q2) { q0 = a0 / d, q1 = a1 / d, q2 = a2 / d; }
    void div5(double p, double q, double r, double s, double &x, double &y, double &z, double &q2, double &r2) {
        x = p / s, y = q / s, z = r / s, q2 = q / p, r2 = r / p;
    }
};

#if defined(__HIPCC__)
#ifndef PL_WAVE_SYNC
#define PL_WAVE_SYNC()                                                                                                 \
    do {                                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                                         \
        __builtin_amdgcn_wave_barrier();                                                                               \
    } while (0)
#endif
// lane I of every row of 16 lanes to all lanes of its row (v_mov_b32_dpp row_newbcast; every lane of the wavefront must be active)
template <int I> __device__ __forceinline__ double eig_row_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + I, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + I, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// ---- device form: lane = 16 x group + gl; the group's matrix and workspace in LDS at `a` ----
template <int n> struct EigWave4 {
    double *a;
    int gl;   // lane & 15
    int lane; // 0 .. 63
    __device__ __forceinline__ double &A(int i, int j) { return a[i * n + j]; }
    __device__ __forceinline__ double &hv(int i) { return a[n * n + i]; }
    __device__ __forceinline__ double &wr(int i) { return a[n * n + n + i]; }
    __device__ __forceinline__ double &wi(int i) { return a[n * n + 2 * n + i]; }
    __device__ __forceinline__ double &out(int i) { return a[n * n + 3 * n + i]; }
    template <class F> __device__ __forceinline__ void lanes(int lo, int hi, bool pred, F f) {
        if (pred && gl >= lo && gl <= hi && gl < n)
            f(gl);
    }
    template <class F> __device__ __forceinline__ void lane0(bool pred, F f) {
        if (pred && gl == 0)
            f();
    }
    __device__ __forceinline__ void sync() { PL_WAVE_SYNC(); }
    __device__ __forceinline__ bool any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; }
    __device__ __forceinline__ int umax(int v) {
        const int a0 = __builtin_amdgcn_readlane(v, 0), a1 = __builtin_amdgcn_readlane(v, 16), a2 = __builtin_amdgcn_readlane(v, 32),
                  a3 = __builtin_amdgcn_readlane(v, 48);
        return max(max(a0, a1), max(a2, a3));
    }
    __device__ __forceinline__ int umin(int v) {
        const int a0 = __builtin_amdgcn_readlane(v, 0), a1 = __builtin_amdgcn_readlane(v, 16), a2 = __builtin_amdgcn_readlane(v, 32),
                  a3 = __builtin_amdgcn_readlane(v, 48);
        return min(min(a0, a1), min(a2, a3));
    }
    // one fp64 division is ~35 instructions: lane i of the group forms quotient i, the others read it.  The same IEEE operation on
    // the same operands in another lane: the same bits.
    __device__ __forceinline__ void div3(double a0, double a1, double a2, double d, double &q0, double &q1, double &q2) {
        const double quo = (gl == 0 ? a0 : gl == 1 ? a1 : a2) / d;
        q0 = eig_row_bcast<0>(quo), q1 = eig_row_bcast<1>(quo), q2 = eig_row_bcast<2>(quo);
    }
    __device__ __forceinline__ void div5(double p, double q, double r, double s, double &x, double &y, double &z, double &q2, double &r2) {
        const double num = gl == 0 ? p : (gl == 1 || gl == 3) ? q : r, den = gl < 3 ? s : p;
        const double quo = num / den;
        x = eig_row_bcast<0>(quo), y = eig_row_bcast<1>(quo), z = eig_row_bcast<2>(quo);
        q2 = eig_row_bcast<3>(quo), r2 = eig_row_bcast<4>(quo);
    }
};
#endif

} // namespace pl
