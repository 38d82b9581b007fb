// poselib_amd - pl_null_vector<n> (pl_solver_p35pf.h: null vector of a singular n x n matrix by Gaussian elimination with complete
// pivoting) by 16 LANES PER MATRIX, four matrices per wavefront (round 5).
//
// The serial routine runs one matrix per lane on a working copy in LDS: ~2500 LDS round trips per matrix, 8 - 10 lanes of 64 at work - the
// root stages of the two focal-length solvers were 19 % / 30 % of the estimators' device time.  Here lane j of a group holds COLUMN j in
// registers.  Rows and columns are not moved: the routine keeps the POSITION every original row / column has in the serial routine's
// swapped matrix (rpos, cpos) and works on "rows / columns at positions >= k":
//   pivot        every lane scans its column, a butterfly over the group takes the largest magnitude; among equal magnitudes the
//                smallest (row position, column position) - the first the serial scan meets - and never a zero (the serial `>`)
//   elimination  the pivot column goes to all lanes, lane r forms the factor of row r (one division per lane instead of n - 1 - k
//                in a row), the factors come back by row_newbcast, every lane updates its column
//   back substitution in position order, the products gathered from the lanes in the serial order of the sum
// Every element sees the serial routine's operations on the serial routine's operands, so the null vector is the same bits - checked
// on the host: the routine is written once over a context of lane primitives; NullFlatHost (one matrix, lane loops as loops) shadows
// every pl_null_vector call of tests/hostmath (PL_EIG_SHADOW_CHECK), NullWave4 is the device form.
//
// Context X:  B(r, j) element (original row r, original column j) - on the device j is the lane's own column;
//             cpos(j), y(j), tmp(j), tmp2(j) per-column scalars;  lanes(f): f(j) for every column j;
//             best_pivot(k, rpos, v, pr, pc): the pivot of step k;  factors(pc, pr, f): f[r] = B(r, pc) / B(pr, pc);
//             from_lane(fn, src): fn(src) - on the device fn of the own lane, shuffled from lane src.
#pragma once
#include "pl_math.h"

namespace pl {

template <int n, class X> PL_HD void pl_null_vector_packed(X &cx, bool act) {
    int rpos[n], row_at[n], col_at[n]; // position of original row r; original row / column at position p  (uniform per matrix)
    for (int i = 0; i < n; ++i)
        rpos[i] = i, row_at[i] = i, col_at[i] = i;
    cx.lanes([&](int j) { cx.cpos(j) = j; });
    bool running = act;
    for (int k = 0; k < n - 1; ++k) {
        if (!cx.any(running))
            break;
        double best;
        int pr, pc;
        cx.best_pivot(k, rpos, best, pr, pc);
        if (!(best > 0)) // (the serial `if (best == 0) break;`: the scan's `>` never takes a NaN either)
            running = false;
        if (cx.any(running)) {
            const bool go = running;
            // the serial routine swaps rows k <-> pr and columns k <-> pc: here only the positions move
            if (go) {
                const int rk = row_at[k], ppos = rpos[pr];
                for (int r = 0; r < n; ++r)
                    rpos[r] = r == pr ? k : (r == rk ? ppos : rpos[r]);
                for (int p = 0; p < n; ++p)
                    row_at[p] = p == k ? pr : (p == ppos ? rk : row_at[p]);
            }
            {
                const int ck = col_at[k];
                int cpp = 0; // position of column pc
                for (int p = 0; p < n; ++p)
                    cpp = col_at[p] == pc ? p : cpp;
                if (go) {
                    cx.lanes([&](int j) { cx.cpos(j) = j == pc ? k : (j == ck ? cpp : cx.cpos(j)); });
                    for (int p = 0; p < n; ++p)
                        col_at[p] = p == k ? pc : (p == cpp ? ck : col_at[p]);
                }
            }
            double f[n];
            cx.factors(pc, pr, f);
            cx.lanes([&](int j) {
                if (go && cx.cpos(j) >= k) {
                    double prow = cx.B(0, j); // B(pr, j)
                    for (int r = 1; r < n; ++r)
                        prow = r == pr ? cx.B(r, j) : prow;
                    for (int r = 0; r < n; ++r)
                        if (rpos[r] > k)
                            cx.B(r, j) -= f[r] * prow;
                }
            });
        }
    }
    // back substitution in position order: y[n - 1] = 1, y[i] = -(sum_{j > i} B[i][j] y[j]) / B[i][i]
    cx.lanes([&](int j) { cx.y(j) = cx.cpos(j) == n - 1 ? 1.0 : 0.0; });
    if (cx.any(act))
        for (int i = n - 2; i >= 0; --i) {
            const int ri = row_at[i];
            cx.lanes([&](int j) { // the row at position i: its entry in column j, and the term of the sum column j contributes
                double b = cx.B(0, j);
                for (int r = 1; r < n; ++r)
                    b = r == ri ? cx.B(r, j) : b;
                cx.tmp(j) = b;
                cx.tmp2(j) = b * cx.y(j);
            });
            double s = 0;
            for (int p = i + 1; p < n; ++p)
                s += cx.from_lane([&](int j) { return cx.tmp2(j); }, col_at[p]);
            const double d = cx.from_lane([&](int j) { return cx.tmp(j); }, col_at[i]);
            const double yi = -s / d;
            const int ci = col_at[i];
            cx.lanes([&](int j) { cx.y(j) = j == ci ? yi : cx.y(j); });
        }
    // (the serial routine's v[colperm[i]] = y[i]: column j's entry is the y of its position - cx.y(j))
}

// ---- host form: ONE matrix, lane loops as loops (tests/hostmath) ----
template <int n> struct NullFlatHost {
    double b[n][n]; // [original row][original column]
    int cp[n];
    double yv[n], t1[n], t2[n];
    double &B(int r, int j) { return b[r][j]; }
    int &cpos(int j) { return cp[j]; }
    double &y(int j) { return yv[j]; }
    double &tmp(int j) { return t1[j]; }
    double &tmp2(int j) { return t2[j]; }
    template <class F> void lanes(F f) {
        for (int j = 0; j < n; ++j)
            f(j);
    }
    bool any(bool v) { return v; }
    void best_pivot(int k, const int *rpos, double &best, int &pr, int &pc) {
        best = 0, pr = 0, pc = 0;
        int brp = 1 << 20, bcp = 1 << 20;
        for (int j = 0; j < n; ++j) {
            if (cp[j] < k)
                continue;
            for (int r = 0; r < n; ++r) {
                if (rpos[r] < k)
                    continue;
                const double v = fabs(b[r][j]);
                if (v > best || (v == best && v > 0 && (rpos[r] < brp || (rpos[r] == brp && cp[j] < bcp))))
                    best = v, pr = r, pc = j, brp = rpos[r], bcp = cp[j];
            }
        }
    }
    void factors(int pc, int pr, double *f) {
        for (int r = 0; r < n; ++r)
            f[r] = b[r][pc] / b[pr][pc];
    }
    template <class F> double from_lane(F fn, int src) { return fn(src); }
};

#if defined(__HIPCC__)
// lane I of every row of 16 lanes to all lanes of its row (v_mov_b32_dpp row_newbcast; every lane of the wavefront must be active)
template <int I> __device__ __forceinline__ double null_row_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + I, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + I, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int n, int I> struct NullBcastAll {
    static __device__ __forceinline__ void run(double q, double *f) {
        f[I] = null_row_bcast<I>(q);
        NullBcastAll<n, I + 1>::run(q, f);
    }
};
template <int n> struct NullBcastAll<n, n> {
    static __device__ __forceinline__ void run(double, double *) {}
};
// ---- device form: lane = 16 x group + gl; lane gl < n holds column gl of its group's matrix in registers ----
template <int n> struct NullWave4 {
    double c[n]; // this lane's column, by original row
    int cp;
    double yv, t1, t2;
    int gl;   // lane & 15
    int lane; // 0 .. 63
    __device__ __forceinline__ double &B(int r, int) { return c[r]; }
    __device__ __forceinline__ int &cpos(int) { return cp; }
    __device__ __forceinline__ double &y(int) { return yv; }
    __device__ __forceinline__ double &tmp(int) { return t1; }
    __device__ __forceinline__ double &tmp2(int) { return t2; }
    template <class F> __device__ __forceinline__ void lanes(F f) {
        if (gl < n)
            f(gl);
    }
    __device__ __forceinline__ bool any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }
    // value of lane `src` of the own group (src uniform within the group)
    __device__ __forceinline__ double shfl_group(double v, int src) {
        const int idx = ((lane & 48) + src) << 2;
        const int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
        return __hiloint2double(hi, lo);
    }
    template <class F> __device__ __forceinline__ double from_lane(F fn, int src) { return shfl_group(fn(gl), src); }
    __device__ __forceinline__ void best_pivot(int k, const int *rpos, double &best, int &pr, int &pc) {
        // own column: largest magnitude over the rows at positions >= k, the smallest row position among equals; zeros never
        double bv = 0;
        int brp = 1 << 20, br = 0;
        const bool mine = gl < n && cp >= k;
#pragma unroll
        for (int r = 0; r < n; ++r) {
            const double v = fabs(c[r]);
            const bool take = mine && rpos[r] >= k && (v > bv || (v == bv && v > 0 && rpos[r] < brp));
            bv = take ? v : bv, brp = take ? rpos[r] : brp, br = take ? r : br;
        }
        int bcp = mine ? cp : (1 << 20), bc = gl;
        // butterfly over the 16 lanes of the group
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const int idx = (lane ^ off) << 2;
            const double ov = __hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(bv)), __builtin_amdgcn_ds_bpermute(idx, __double2loint(bv)));
            const int orp = __builtin_amdgcn_ds_bpermute(idx, brp), ocp = __builtin_amdgcn_ds_bpermute(idx, bcp);
            const int orr = __builtin_amdgcn_ds_bpermute(idx, br), oc = __builtin_amdgcn_ds_bpermute(idx, bc);
            const bool take = ov > bv || (ov == bv && ov > 0 && (orp < brp || (orp == brp && ocp < bcp)));
            bv = take ? ov : bv, brp = take ? orp : brp, bcp = take ? ocp : bcp, br = take ? orr : br, bc = take ? oc : bc;
        }
        best = bv, pr = br, pc = bc;
    }
    // f[r] = B(r, pc) / B(pr, pc): the pivot column to every lane, one division per lane (lane r: row r), the quotients back
    __device__ __forceinline__ void factors(int pc, int pr, double *f) {
        double num = 0, den = 1;
#pragma unroll
        for (int r = 0; r < n; ++r) {
            const double v = shfl_group(c[r], pc);
            num = r == gl ? v : num;
            den = r == pr ? v : den;
        }
        const double q = num / den;
        NullBcastAll<n, 0>::run(q, f);
    }
};
#endif

} // namespace pl
