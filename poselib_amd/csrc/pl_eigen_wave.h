// poselib_amd - eigenvalues of a small matrix by ONE WAVEFRONT (device only): pl_general_eigenvalues of pl_solver_p35pf.h with the
// matrix of one sample in LDS and the 64 lanes working on it together.
//
// The serial routine runs one sample per lane; 16 - 32 samples share a wavefront, every lane somewhere else in its own iteration
// (the wavefront executes the union of their paths), every access to the matrix a round trip to LDS: 0.45 ms per batch of 1001
// action matrices of P3.5Pf.  Here the control flow is the sample's own (wave-uniform: scalar branches, no divergence), the
// scalars of the iteration (shifts, reflectors) are computed by every lane alike from broadcast reads, and the three inner loops of
// the algorithm - a reflector applied to its rows over the columns j, to its columns over the rows i, the Householder updates of
// the Hessenberg reduction - run one column / row per lane.  Every matrix element sees the operations of the serial routine in the
// serial routine's order (an update of element (i, j) never depends on which lane performs it), so the eigenvalues are the same
// bits (tests/test_zz_gpu_focal.py: the solver's and the estimator's results against the oracle's).
//
// LDS per wavefront: eig_wave_doubles(n) = n * n (matrix, row-major) + 3 n (Householder vector, wr, wi).
#pragma once
#include "pl_math.h"

namespace pl {

constexpr int eig_wave_doubles(int n) { return n * n + 3 * n; }

// orders the LDS accesses of the lanes of one wavefront (the hardware executes a wavefront's LDS instructions in order; this keeps the
// compiler from moving or caching accesses across the phases of the algorithm)
#define PL_WAVE_SYNC()                                                                                                 \
    do {                                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                                         \
        __builtin_amdgcn_wave_barrier();                                                                               \
    } while (0)

// a / b, c / d, e / f ... at once: the divisions of a reflector are independent of one another, and an fp64 division is ~35
// instructions that every lane would issue alike - lane i forms quotient i, the others read it (v_readlane).  The same IEEE
// operation on the same operands in another lane: the same bits.
__device__ __forceinline__ double eig_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// pl_general_eigenvalues<n> (pl_solver_p35pf.h): Householder reduction to Hessenberg form, Francis double-shift QR iteration.
// a: the matrix (LDS, row-major, destroyed), followed by 3 n doubles of workspace; real and imaginary parts are left in
// a[n * n + n ...] and a[n * n + 2 n ...], in the order the deflation leaves them (NaN where the iteration did not converge).
template <int n> __device__ void pl_general_eigenvalues_wave(double *a, int lane) {
    double *const hv = a + n * n, *const wr = hv + n, *const wi = wr + n;
#define PL_A(i, j) a[(i) * n + (j)]
    PL_WAVE_SYNC();
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
        if (lane < n) // the Householder vector: lane r its entry
            hv[lane] = lane <= k ? 0.0 : lane == k + 1 ? 1.0 : PL_A(lane, k) / (c0 - beta);
        const double tau = (beta - c0) / beta;
        PL_WAVE_SYNC();
        if (lane < n) { // column c = lane
            const int c = lane;
            double t = 0;
            for (int r = k + 1; r < n; ++r)
                t += hv[r] * PL_A(r, c);
            for (int r = k + 1; r < n; ++r)
                PL_A(r, c) -= tau * hv[r] * t;
        }
        PL_WAVE_SYNC();
        if (lane < n) { // row r = lane
            const int r = lane;
            double t = 0;
            for (int c = k + 1; c < n; ++c)
                t += PL_A(r, c) * hv[c];
            for (int c = k + 1; c < n; ++c)
                PL_A(r, c) -= tau * t * hv[c];
        }
        PL_WAVE_SYNC();
        if (lane == k + 1)
            PL_A(k + 1, k) = beta;
        if (lane >= k + 2 && lane < n)
            PL_A(lane, k) = 0;
        PL_WAVE_SYNC();
    }
    if (lane < n)
        wr[lane] = wi[lane] = 0.0;
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
            PL_WAVE_SYNC();
            for (l = nn; l >= 1; --l) {
                s = fabs(PL_A(l - 1, l - 1)) + fabs(PL_A(l, l));
                if (s == 0)
                    s = anorm;
                if (fabs(PL_A(l, l - 1)) <= eps * s) {
                    PL_WAVE_SYNC();
                    if (lane == 0)
                        PL_A(l, l - 1) = 0;
                    PL_WAVE_SYNC();
                    break;
                }
            }
            x = PL_A(nn, nn);
            if (l == nn) {
                if (lane == 0) {
                    wr[nn] = x + t;
                    wi[nn] = 0;
                }
                --nn;
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
                        double w1 = x + z, w2 = x + z;
                        if (z != 0)
                            w2 = x - w / z;
                        if (lane == 0) {
                            wr[nn - 1] = w1, wr[nn] = w2;
                            wi[nn - 1] = wi[nn] = 0;
                        }
                    } else {
                        if (lane == 0) {
                            wr[nn - 1] = wr[nn] = x + p;
                            wi[nn - 1] = z;
                            wi[nn] = -z;
                        }
                    }
                    nn -= 2;
                } else {
                    if (its == 60) { // no convergence: what is left counts as NaN (it passes the reference's filter as such)
                        PL_WAVE_SYNC();
                        if (lane <= nn) {
                            wr[lane] = __builtin_nan("");
                            wi[lane] = 0;
                        }
                        PL_WAVE_SYNC();
                        return;
                    }
                    if (its == 10 || its == 20) {
                        t += x;
                        PL_WAVE_SYNC();
                        if (lane <= nn)
                            PL_A(lane, lane) -= x;
                        PL_WAVE_SYNC();
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
                        {
                            const double quo = (lane == 0 ? p : lane == 1 ? q : r) / s; // p /= s, q /= s, r /= s
                            p = eig_bcast(quo, 0), q = eig_bcast(quo, 1), r = eig_bcast(quo, 2);
                        }
                        if (m == l)
                            break;
                        const double u = fabs(PL_A(m, m - 1)) * (fabs(q) + fabs(r));
                        const double v = fabs(p) * (fabs(PL_A(m - 1, m - 1)) + fabs(z) + fabs(PL_A(m + 1, m + 1)));
                        if (u <= eps * v)
                            break;
                    }
                    PL_WAVE_SYNC();
                    if (lane >= m + 2 && lane <= nn) { // i = lane
                        PL_A(lane, lane - 2) = 0;
                        if (lane != m + 2)
                            PL_A(lane, lane - 3) = 0;
                    }
                    PL_WAVE_SYNC();
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = PL_A(k, k - 1);
                            q = PL_A(k + 1, k - 1);
                            r = (k != nn - 1) ? PL_A(k + 2, k - 1) : 0.0;
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0) {
                                const double quo = (lane == 0 ? p : lane == 1 ? q : r) / x; // p /= x, q /= x, r /= x
                                p = eig_bcast(quo, 0), q = eig_bcast(quo, 1), r = eig_bcast(quo, 2);
                            }
                        }
                        const double sq = sqrt(p * p + q * q + r * r);
                        if ((s = (p >= 0 ? sq : -sq)) != 0) {
                            PL_WAVE_SYNC();
                            if (lane == 0) {
                                if (k == m) {
                                    if (l != m)
                                        PL_A(k, k - 1) = -PL_A(k, k - 1);
                                } else {
                                    PL_A(k, k - 1) = -s * x;
                                }
                            }
                            p += s;
                            {   // x = p / s, y = q / s, z = r / s; q /= p, r /= p
                                const double num = lane == 0 ? p : (lane == 1 || lane == 3) ? q : r, den = lane < 3 ? s : p;
                                const double quo = num / den;
                                x = eig_bcast(quo, 0), y = eig_bcast(quo, 1), z = eig_bcast(quo, 2);
                                q = eig_bcast(quo, 3), r = eig_bcast(quo, 4);
                            }
                            PL_WAVE_SYNC();
                            if (lane >= k && lane <= nn) { // the reflector on rows k .. k + 2: column j = lane
                                const int j = lane;
                                double pp = PL_A(k, j) + q * PL_A(k + 1, j);
                                if (k != nn - 1) {
                                    pp += r * PL_A(k + 2, j);
                                    PL_A(k + 2, j) -= pp * z;
                                }
                                PL_A(k + 1, j) -= pp * y;
                                PL_A(k, j) -= pp * x;
                            }
                            PL_WAVE_SYNC();
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            if (lane >= l && lane <= mmin) { // on columns k .. k + 2: row i = lane
                                const int i = lane;
                                double pp = x * PL_A(i, k) + y * PL_A(i, k + 1);
                                if (k != nn - 1) {
                                    pp += z * PL_A(i, k + 2);
                                    PL_A(i, k + 2) -= pp * r;
                                }
                                PL_A(i, k + 1) -= pp * q;
                                PL_A(i, k) -= pp;
                            }
                            PL_WAVE_SYNC();
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
#undef PL_A
    PL_WAVE_SYNC();
}

} // namespace pl
