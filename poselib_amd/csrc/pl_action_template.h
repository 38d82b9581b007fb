// poselib_amd - the machinery the two focal-length minimal solvers share: they are ACTION-MATRIX solvers in the reference
// (solvers/p35pf.cc, solvers/relpose_6pt_focal.cc) - coefficients of fixed polynomial equations written into a sparse matrix
// [C0 | C1], X = C0^-1 C1, a few rows of X are the non-trivial rows of the matrix of "multiply by one unknown" on a basis of
// the quotient ring.  The equations and layouts are tables (pl_focal_templates.h, generated from the equations by
// scripts/gen_focal_templates.py); this file evaluates them and does the dense linear algebra in the operation order of
// the reference's build (Eigen's decompositions as oracle/eigen_shim restates them - what oracle/_ref runs), so that the
// solutions are the reference's to the last bit:
//
//   template_coefficient     one coefficient = its products added in the listed order
//   lu_*                     LU with partial pivoting (first maximum wins), right-looking; forward substitution = the same
//                            row operations on the right-hand sides; back substitution with ascending column index
//   colpiv_qr_solve          least squares through Householder QR with column pivoting (p35pf.cc:34)
//   householder_qr_solve     square system through unpivoted Householder QR (relpose_6pt_focal.cc:46)
//   danilevsky_*             characteristic polynomial (misc/sturm.h:287-326)
//
// Serial forms (PL_HD): the host statement of the solvers (tests/hostmath against the oracle, bit for bit) and the per-root
// systems on the device (one lane per root).  Wave forms (device): one wavefront per sample, the matrix in LDS, a column (LU)
// or a column / row (Danilevsky) per lane; every element sees the serial operations in the serial order.
#pragma once
#include "pl_focal_templates.h"
#include "pl_math.h"

namespace pl {

static constexpr double kTemplateMultiplier[8] = {1, -1, 2, -2, 3, -3, 6, -6};

// x^3 correctly rounded (x^2 = h + l exactly, x^3 = h x + l x in double-double).  The reference's six-point formulas call
// std::pow(d, 3): glibc's result is this number except for 8 of 10^4 arguments (one unit in the last place);
// oracle/src/solvers_focal.cc has both (orc_set_exact_cubes) and the device tests compare with this one bit for bit.
PL_HD double exact_cube(double x) {
    const double h = x * x, l = __builtin_fma(x, x, -h);
    const double p = h * x, e = __builtin_fma(h, x, -p);
    return p + (e + l * x);
}

// coefficient k of a template: t = m; t *= factor, factor by factor; the products added from the left.  POWERS (six-point
// solver): a repeated index is ONE factor (d * d, or the cube), as the reference's formulas write pow(d, 2) / pow(d, 3).
template <bool POWERS, class Arr> PL_HD double template_coefficient(const Arr &d, const uint16_t *start, const uint32_t *packed, int k) {
    double sum = 0;
    const int e0 = start[k], e1 = start[k + 1];
    for (int e = e0; e < e1; ++e) {
        const uint32_t w = packed[e];
        const int a = (w >> 8) & 255, b = (w >> 16) & 255, c = (int)(w >> 24);
        double t = kTemplateMultiplier[w & 7];
        const double da = d[a], db = d[b];
        if (!POWERS) {
            t = t * da;
            t = t * db;
            if (c != 255)
                t = t * d[c];
        } else if (c == 255) {
            t = a == b ? t * (da * da) : (t * da) * db;
        } else {
            const double dc = d[c];
            if (a == b && b == c)
                t = t * exact_cube(da);
            else if (a == b)
                t = (t * (da * da)) * dc;
            else if (b == c)
                t = (t * da) * (db * db);
            else
                t = ((t * da) * db) * dc;
        }
        sum = e == e0 ? t : sum + t;
    }
    return sum;
}

// ---- serial LU.  C: ROWS x COLS row-major at row stride S (the first ROWS columns C0, the rest the right-hand sides), destroyed.
// Afterwards rows ROWS - TAIL .. ROWS - 1 of the right-hand-side columns hold those rows of C0^-1 C1.
template <int ROWS, int COLS, int S, int TAIL> PL_HD void lu_solve_tail(double *C) {
    for (int k = 0; k < ROWS; ++k) {
        int piv = k;
        double best = fabs(C[k * S + k]);
        for (int i = k + 1; i < ROWS; ++i)
            if (fabs(C[i * S + k]) > best) {
                best = fabs(C[i * S + k]);
                piv = i;
            }
        if (piv != k)
            for (int j = 0; j < COLS; ++j) {
                const double t = C[k * S + j];
                C[k * S + j] = C[piv * S + j];
                C[piv * S + j] = t;
            }
        const double pk = C[k * S + k];
        if (pk != 0)
            for (int i = k + 1; i < ROWS; ++i)
                C[i * S + k] /= pk;
        for (int i = k + 1; i < ROWS; ++i) {
            const double f = C[i * S + k];
            for (int j = k + 1; j < COLS; ++j)
                C[i * S + j] -= f * C[k * S + j];
        }
    }
    for (int c = ROWS; c < COLS; ++c)
        for (int i = ROWS - 1; i >= ROWS - TAIL; --i) {
            double s = C[i * S + c];
            for (int j = i + 1; j < ROWS; ++j)
                s -= C[i * S + j] * C[j * S + c];
            C[i * S + c] = s / C[i * S + i];
        }
}

// ---- x = argmin |A x - b|, A ROWS x COLS column-major (destroyed), b (destroyed): Householder QR with column pivoting, the
// column norms recomputed at every step, the reflectors applied to b as they are formed
template <int ROWS, int COLS> PL_HD void colpiv_qr_solve(double *A, double *b, double *x) {
    int colperm[COLS];
    double R[COLS * COLS], v[ROWS], w[COLS];
    for (int j = 0; j < COLS; ++j)
        colperm[j] = j;
    for (int k = 0; k < COLS; ++k) {
        int best = k;
        double bn = -1;
        for (int j = k; j < COLS; ++j) {
            double n = 0;
            for (int i = k; i < ROWS; ++i)
                n += A[j * ROWS + i] * A[j * ROWS + i];
            if (n > bn) {
                bn = n;
                best = j;
            }
        }
        if (best != k) {
            for (int i = 0; i < ROWS; ++i) {
                const double t = A[k * ROWS + i];
                A[k * ROWS + i] = A[best * ROWS + i];
                A[best * ROWS + i] = t;
            }
            for (int i = 0; i < k; ++i) {
                const double t = R[i * COLS + k];
                R[i * COLS + k] = R[i * COLS + best];
                R[i * COLS + best] = t;
            }
            const int t = colperm[k];
            colperm[k] = colperm[best];
            colperm[best] = t;
        }
        double tail = 0;
        for (int i = k + 1; i < ROWS; ++i)
            tail += A[k * ROWS + i] * A[k * ROWS + i];
        const double c0 = A[k * ROWS + k];
        double beta = sqrt(c0 * c0 + tail);
        if (c0 >= 0)
            beta = -beta;
        for (int i = 0; i < ROWS; ++i)
            v[i] = 0;
        double tau = 0;
        if (tail > 2.2250738585072014e-308) {
            v[k] = 1;
            for (int i = k + 1; i < ROWS; ++i)
                v[i] = A[k * ROWS + i] / (c0 - beta);
            tau = (beta - c0) / beta;
        } else {
            beta = c0;
        }
        R[k * COLS + k] = beta;
        for (int cc = k + 1; cc < COLS; ++cc) {
            double t = 0;
            for (int i = k; i < ROWS; ++i)
                t += v[i] * A[cc * ROWS + i];
            for (int i = k; i < ROWS; ++i)
                A[cc * ROWS + i] -= tau * v[i] * t;
            R[k * COLS + cc] = A[cc * ROWS + k];
        }
        double t = 0;
        for (int i = k; i < ROWS; ++i)
            t += v[i] * b[i];
        for (int i = k; i < ROWS; ++i)
            b[i] -= tau * v[i] * t;
    }
    for (int i = COLS - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < COLS; ++j)
            s -= R[i * COLS + j] * w[j];
        w[i] = s / R[i * COLS + i];
    }
    for (int j = 0; j < COLS; ++j)
        x[colperm[j]] = w[j];
}

// ---- x = A^-1 b, A n x n column-major (destroyed), b (destroyed): unpivoted Householder QR, Q^T b reflector by reflector, back
// substitution
template <int n> PL_HD void householder_qr_solve(double *A, double *b, double *x) {
    double tau[n];
    for (int k = 0; k < n; ++k) {
        double tail_sq = 0;
        for (int r = k + 1; r < n; ++r)
            tail_sq += A[k * n + r] * A[k * n + r];
        const double c0 = A[k * n + k];
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0;
            for (int r = k + 1; r < n; ++r)
                A[k * n + r] = 0;
        } else {
            double beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0)
                beta = -beta;
            for (int r = k + 1; r < n; ++r)
                A[k * n + r] = A[k * n + r] / (c0 - beta);
            tau[k] = (beta - c0) / beta;
            A[k * n + k] = beta;
        }
        if (tau[k] != 0)
            for (int c = k + 1; c < n; ++c) {
                double t = 0;
                for (int r = k + 1; r < n; ++r)
                    t += A[k * n + r] * A[c * n + r];
                t += A[c * n + k];
                A[c * n + k] -= tau[k] * t;
                for (int r = k + 1; r < n; ++r)
                    A[c * n + r] -= tau[k] * A[k * n + r] * t;
            }
    }
    for (int k = 0; k < n; ++k) {
        if (tau[k] == 0)
            continue;
        double t = 0;
        for (int r = k + 1; r < n; ++r)
            t += A[k * n + r] * b[r];
        t += b[k];
        b[k] -= tau[k] * t;
        for (int r = k + 1; r < n; ++r)
            b[r] -= tau[k] * A[k * n + r] * t;
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k)
            s -= A[k * n + i] * x[k];
        x[i] = s / A[i * n + i];
    }
}

// ---- characteristic polynomial by Danilevsky's method with pivoting: A n x n row-major (destroyed), p[0 .. n] (p[n] = 1)
template <int n> PL_HD void danilevsky_charpoly(double *A, double *p) {
    double v[n], vinv[n], acol[n], prod[n];
    for (int i = n - 1; i > 0; --i) {
        int piv_ind = i - 1;
        double piv = fabs(A[i * n + i - 1]);
        for (int j = 0; j < i - 1; ++j)
            if (fabs(A[i * n + j]) > piv) {
                piv = fabs(A[i * n + j]);
                piv_ind = j;
            }
        if (piv_ind != i - 1) {
            for (int c = 0; c < n; ++c) {
                const double t = A[(i - 1) * n + c];
                A[(i - 1) * n + c] = A[piv_ind * n + c];
                A[piv_ind * n + c] = t;
            }
            for (int r = 0; r < n; ++r) {
                const double t = A[r * n + i - 1];
                A[r * n + i - 1] = A[r * n + piv_ind];
                A[r * n + piv_ind] = t;
            }
        }
        piv = A[i * n + i - 1];
        for (int c = 0; c < n; ++c)
            v[c] = A[i * n + c];
        for (int c = 0; c < n; ++c) {
            double s = v[0] * A[c];
            for (int k = 1; k < n; ++k)
                s += v[k] * A[k * n + c];
            prod[c] = s;
        }
        for (int c = 0; c < n; ++c)
            A[(i - 1) * n + c] = prod[c];
        for (int c = 0; c < n; ++c)
            vinv[c] = -1.0 * v[c];
        vinv[i - 1] = 1;
        for (int c = 0; c < n; ++c)
            vinv[c] = vinv[c] / piv;
        vinv[i - 1] -= 1;
        for (int r = 0; r < n; ++r)
            acol[r] = A[r * n + i - 1];
        for (int j = 0; j <= i; ++j)
            for (int c = 0; c < n; ++c)
                A[j * n + c] = A[j * n + c] + acol[j] * vinv[c];
        for (int c = 0; c < n; ++c)
            A[i * n + c] = 0;
        A[i * n + i - 1] = 1;
    }
    p[n] = 1;
    for (int i = 0; i < n; ++i)
        p[i] = -A[n - i - 1];
}

#ifdef __HIPCC__
// ===================================================================================================== wave forms (device)
#ifndef PL_WAVE_SYNC
#define PL_WAVE_SYNC()                                                                                                 \
    do {                                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                                         \
        __builtin_amdgcn_wave_barrier();                                                                               \
    } while (0)
#endif

// The coefficients of a sample by its wavefront: lane l evaluates coefficients l, l + 64, ...  d: the data (LDS), out: LDS.
template <bool POWERS, int NCOEF>
__device__ __forceinline__ void template_coefficients_wave(const double *d, const uint16_t *start, const uint32_t *packed, double *out,
                                                           int lane) {
    for (int k = lane; k < NCOEF; k += 64)
        out[k] = template_coefficient<POWERS>(d, start, packed, k);
}

// [C0 | C1] from the coefficients: C (LDS) ROWS x COLS row-major at row stride S; lane c writes column c
template <int ROWS, int COLS, int S>
__device__ __forceinline__ void template_fill_wave(const double *coef, const uint16_t *colstart, const uint8_t *row, const uint16_t *which,
                                                   double *C, int lane) {
    for (int e = lane; e < ROWS * S; e += 64)
        C[e] = 0.0;
    PL_WAVE_SYNC();
    if (lane < COLS)
        for (int e = colstart[lane]; e < colstart[lane + 1]; ++e)
            C[row[e] * S + lane] = coef[which[e]];
    PL_WAVE_SYNC();
}

// lu_solve_tail by one wavefront: lane c holds column c (COLS <= 64).  The pivot search is done by every lane alike (broadcast
// reads of the pivot column), the factors of a step one per lane (lane i: row i), the updates one column per lane.
template <int ROWS, int COLS, int S, int TAIL> __device__ __forceinline__ void lu_solve_tail_wave(double *C, int lane) {
    static_assert(COLS <= 64 && ROWS <= 64, "a column / a row per lane");
    for (int k = 0; k < ROWS; ++k) {
        int piv = k;
        double best = fabs(C[k * S + k]);
        for (int i = k + 1; i < ROWS; ++i) {
            const double v = fabs(C[i * S + k]);
            if (v > best) {
                best = v;
                piv = i;
            }
        }
        if (piv != k) { // (wave-uniform)
            PL_WAVE_SYNC();
            if (lane < COLS) {
                const double t = C[k * S + lane];
                C[k * S + lane] = C[piv * S + lane];
                C[piv * S + lane] = t;
            }
            PL_WAVE_SYNC();
        }
        const double pk = C[k * S + k];
        if (pk != 0) {
            PL_WAVE_SYNC();
            if (lane > k && lane < ROWS)
                C[lane * S + k] /= pk;
        }
        PL_WAVE_SYNC();
        if (lane > k && lane < COLS) {
            const double ck = C[k * S + lane];
            for (int i = k + 1; i < ROWS; ++i)
                C[i * S + lane] -= C[i * S + k] * ck;
        }
        PL_WAVE_SYNC();
    }
    if (lane >= ROWS && lane < COLS)
        for (int i = ROWS - 1; i >= ROWS - TAIL; --i) {
            double s = C[i * S + lane];
            for (int j = i + 1; j < ROWS; ++j)
                s -= C[i * S + j] * C[j * S + lane];
            C[i * S + lane] = s / C[i * S + i];
        }
    PL_WAVE_SYNC();
}

// danilevsky_charpoly by one wavefront: A (LDS) n x n row-major, ws: 2 n doubles of LDS; lane c = column c (products, updates)
template <int n> __device__ __forceinline__ void danilevsky_charpoly_wave(double *A, double *ws, double *p, int lane) {
    double *v = ws, *acol = ws + n;
    for (int i = n - 1; i > 0; --i) {
        int piv_ind = i - 1;
        double piv = fabs(A[i * n + i - 1]);
        for (int j = 0; j < i - 1; ++j) {
            const double a = fabs(A[i * n + j]);
            if (a > piv) {
                piv = a;
                piv_ind = j;
            }
        }
        if (piv_ind != i - 1) { // (uniform)
            PL_WAVE_SYNC();
            if (lane < n) {
                const double t = A[(i - 1) * n + lane];
                A[(i - 1) * n + lane] = A[piv_ind * n + lane];
                A[piv_ind * n + lane] = t;
            }
            PL_WAVE_SYNC();
            if (lane < n) {
                const double t = A[lane * n + i - 1];
                A[lane * n + i - 1] = A[lane * n + piv_ind];
                A[lane * n + piv_ind] = t;
            }
            PL_WAVE_SYNC();
        }
        piv = A[i * n + i - 1];
        if (lane < n)
            v[lane] = A[i * n + lane];
        PL_WAVE_SYNC();
        double prod = 0;
        if (lane < n) {
            prod = v[0] * A[lane];
            for (int k = 1; k < n; ++k)
                prod += v[k] * A[k * n + lane];
        }
        PL_WAVE_SYNC();
        double vinv = 0;
        if (lane < n) {
            A[(i - 1) * n + lane] = prod;
            vinv = -1.0 * v[lane];
            if (lane == i - 1)
                vinv = 1;
            vinv = vinv / piv;
            if (lane == i - 1)
                vinv -= 1;
        }
        PL_WAVE_SYNC();
        if (lane < n)
            acol[lane] = A[lane * n + i - 1];
        PL_WAVE_SYNC();
        if (lane < n) {
            for (int j = 0; j <= i; ++j)
                A[j * n + lane] = A[j * n + lane] + acol[j] * vinv;
            A[i * n + lane] = lane == i - 1 ? 1.0 : 0.0;
        }
        PL_WAVE_SYNC();
    }
    if (lane < n)
        p[lane] = -A[n - lane - 1];
    if (lane == 0)
        p[n] = 1;
    PL_WAVE_SYNC();
}
#endif // __HIPCC__

} // namespace pl
