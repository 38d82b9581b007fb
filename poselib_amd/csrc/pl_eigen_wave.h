// poselib_amd - real eigenvalues of a small matrix by ONE WAVEFRONT (device only): pl_real_eigenvalues / pl_balance_pow2 of
// pl_solver_p35pf.h / pl_solver_6ptf.h with the matrix of one sample in LDS and the 64 lanes working on it together.
//
// The serial routines run one sample per lane; 16 - 32 samples share a wavefront, every lane somewhere else in its own iteration
// (the wavefront executes the union of their paths), every access to the matrix a round trip to LDS: 2.1 ms per batch for the
// 15 x 15 companion matrices of the shared-focal solver, 0.45 ms for the 10 x 10 action matrices of P3.5Pf.  Here the control flow
// is the sample's own (wave-uniform: scalar branches, no divergence), the scalars of the iteration (shifts, reflectors) are
// computed by every lane alike from broadcast reads, and the three inner loops of the algorithm - a reflector applied to its rows
// over the columns j, to its columns over the rows i, the Householder updates of the Hessenberg reduction - run one column / row per
// lane.  Every matrix element sees the operations of the serial routine in the serial routine's order (an update of element (i, j)
// never depends on which lane performs it), so the eigenvalues are the same bits (tests/test_zz_gpu_focal.py,
// tests/test_zz_gpu_shared_focal.py: the estimators' results against the oracle's).
//
// LDS per wavefront: kEigWaveDoubles(n) = n * n (matrix, row-major) + 4 n (Householder vector, wr, wi, out).
#pragma once
#include "pl_math.h"

namespace pl {

constexpr int eig_wave_doubles(int n) { return n * n + 4 * n; }

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

// pl_balance_pow2<n> (pl_solver_6ptf.h): Parlett-Reinsch balancing without the permutation step
template <int n> __device__ void pl_balance_pow2_wave(double *a, int lane) {
    bool done = false;
    for (int sweep = 0; sweep < 64 && !done; ++sweep) {
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
                PL_WAVE_SYNC();
                if (lane < n)
                    a[i * n + lane] *= g;
                PL_WAVE_SYNC();
                if (lane < n)
                    a[lane * n + i] *= f;
                PL_WAVE_SYNC();
            }
        }
    }
}

// pl_real_eigenvalues<n> (pl_solver_p35pf.h): Householder reduction to Hessenberg form, Francis double-shift QR iteration.
// a: the matrix (LDS, row-major, destroyed), followed by 4 n doubles of workspace; the eigenvalues that count as real are left,
// ascending, in a[n * n + 3 n ...]; returns their number (every lane the same).
template <int n> __device__ int pl_real_eigenvalues_wave(double *a, double tol, int lane) {
    double *const hv = a + n * n, *const wr = hv + n, *const wi = wr + n, *const out = wi + n;
#define PL_A(i, j) a[(i) * n + (j)]
    PL_WAVE_SYNC();
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
                    if (its == 60)
                        return 0;
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
    int m = 0; // (every lane walks the same list; lane 0 writes it)
    for (int i = 0; i < n; ++i)
        if (fabs(wi[i]) <= tol * (1.0 + fabs(wr[i]))) { // insertion into the ascending list
            int j = m++;
            const double v = wr[i];
            PL_WAVE_SYNC();
            if (lane == 0) {
                while (j > 0 && out[j - 1] > v) {
                    out[j] = out[j - 1];
                    --j;
                }
                out[j] = v;
            }
            PL_WAVE_SYNC();
        }
    return m;
}

// six_companion (pl_solver_6ptf.h) by one wavefront: the row reduction of the w^2 part (six steps of Gaussian elimination with complete
// pivoting over the remaining rows), the 10 x 10 system with 15 right-hand sides (LU with partial pivoting, back substitution) and the
// 15 x 15 companion matrix.  Cw (300: the equations, destroyed), A (100), B (150), fv (16) and T (225, the result) are plain LDS arrays
// of the wavefront.  The pivot searches keep the serial routine's tie rule (the first of equal candidates in its scan order), the
// factors of a step are formed one per lane, the row updates run over the lanes - every element sees the serial operations in the
// serial order.  As one lane per sample (matrices in LDS, 16 samples per wavefront): 360 k cycles = 81 % of k_sfocal_setup.
__device__ inline bool six_companion_wave(double *Cw, double *T, double *A, double *B, double *fv, int lane) {
#define PL_C(k, r, col) Cw[(k) * 100 + (r) * 10 + (col)]
#define PL_LA(r, col) A[(r) * 10 + (col)]
#define PL_RB(r, col) B[(r) * 15 + (col)]
    PL_WAVE_SYNC();
    for (int k = 0; k < 6; ++k) {
        // the largest |C2(r, col)|, r = 1 + k .. 9, the first in (r, col) order among equals; nothing > 0: degenerate
        double best = 0;
        int bl = 0x7fffffff;
        {
            const int l0 = (1 + k) * 10 + lane, l1 = l0 + 64;
            if (l0 < 100) {
                const double v = fabs(Cw[200 + l0]);
                if (v > best)
                    best = v, bl = l0;
            }
            if (l1 < 100) {
                const double v = fabs(Cw[200 + l1]);
                if (v > best)
                    best = v, bl = l1;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(best, off, 64);
                const int ol = __shfl_xor(bl, off, 64);
                if (ov > best || (ov == best && ol < bl))
                    best = ov, bl = ol;
            }
        }
        if (bl == 0x7fffffff)
            return false;
        const int pr = bl / 10, pc = bl - 10 * pr;
        if (pr != 1 + k) {
            if (lane < 30) {
                const int m = lane / 10, col = lane - 10 * m;
                const double t = PL_C(m, 1 + k, col);
                PL_C(m, 1 + k, col) = PL_C(m, pr, col);
                PL_C(m, pr, col) = t;
            }
            PL_WAVE_SYNC();
        }
        if (lane >= 2 + k && lane < 10)
            fv[lane] = PL_C(2, lane, pc) / PL_C(2, 1 + k, pc);
        PL_WAVE_SYNC();
        for (int idx = lane; idx < (8 - k) * 30; idx += 64) {
            const int r = 2 + k + idx / 30, e = idx % 30, m = e / 10, col = e - 10 * m;
            const double f = fv[r];
            if (f != 0)
                PL_C(m, r, col) -= f * PL_C(m, 1 + k, col);
        }
        PL_WAVE_SYNC();
        if (lane >= 2 + k && lane < 10 && fv[lane] != 0)
            PL_C(2, lane, pc) = 0;
        PL_WAVE_SYNC();
    }
    for (int e = lane; e < 100; e += 64) { // A = L^T
        const int col = e / 10, c = e - 10 * col;
        PL_LA(col, c) = c < 6 ? PL_C(2, 1 + c, col) : c < 9 ? PL_C(1, 7 + (c - 6), col) : PL_C(0, 0, col);
    }
    for (int e = lane; e < 150; e += 64) { // B = (right-hand rows)^T
        const int col = e / 15, c = e - 15 * col;
        PL_RB(col, c) = c < 6 ? PL_C(0, 1 + c, col) : c < 12 ? PL_C(1, 1 + (c - 6), col) : PL_C(0, 7 + (c - 12), col);
    }
    PL_WAVE_SYNC();
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
            PL_WAVE_SYNC();
            if (lane < 10) {
                const double t = PL_LA(k, lane);
                PL_LA(k, lane) = PL_LA(pr, lane);
                PL_LA(pr, lane) = t;
            } else if (lane < 25) {
                const int col = lane - 10;
                const double t = PL_RB(k, col);
                PL_RB(k, col) = PL_RB(pr, col);
                PL_RB(pr, col) = t;
            }
        }
        PL_WAVE_SYNC();
        if (lane > k && lane < 10)
            fv[lane] = PL_LA(lane, k) / PL_LA(k, k);
        PL_WAVE_SYNC();
        const int wid = 24 - k; // columns k + 1 .. 9 of A, then the 15 of B
        for (int idx = lane; idx < (9 - k) * wid; idx += 64) {
            const int r = k + 1 + idx / wid, c = idx % wid;
            const double f = fv[r];
            if (f != 0) {
                if (c < 9 - k)
                    PL_LA(r, k + 1 + c) -= f * PL_LA(k, k + 1 + c);
                else
                    PL_RB(r, c - (9 - k)) -= f * PL_RB(k, c - (9 - k));
            }
        }
        PL_WAVE_SYNC();
    }
    if (lane < 15) { // back substitution: a right-hand side per lane
        const int col = lane;
        for (int r = 9; r >= 0; --r) {
            double s = PL_RB(r, col);
            for (int m = r + 1; m < 10; ++m)
                s -= PL_LA(r, m) * PL_RB(m, col);
            PL_RB(r, col) = s / PL_LA(r, r);
        }
    }
    PL_WAVE_SYNC();
    for (int e = lane; e < 225; e += 64) {
        const int row = e / 15, c = e - 15 * row;
        double v = 0.0;
        if (row < 6) {
            v = c == 9 + row ? 1.0 : 0.0;
        } else if (row < 9) {
            if (c < 9)
                v = -PL_RB(c, 12 + (row - 6));
        } else {
            const int i = row - 9;
            if (c >= 9) {
                v = -PL_RB(c - 9, 6 + i);
            } else {
                double s = -PL_RB(c, i);
                for (int j = 0; j < 3; ++j)
                    s += PL_RB(6 + j, 6 + i) * PL_RB(c, 12 + j);
                v = s;
            }
        }
        T[e] = v;
    }
    PL_WAVE_SYNC();
#undef PL_C
#undef PL_LA
#undef PL_RB
    return true;
}

} // namespace pl
