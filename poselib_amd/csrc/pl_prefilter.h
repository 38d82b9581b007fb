// poselib_amd — conservative fp32 pre-filters of the four MSAC scores.
//
// A pre-filter may only answer "this correspondence is CERTAINLY NOT an inlier of this model" (per the reference's
// own fp64 arithmetic, utils.cc:36-65 / :204-239 / :300-329).  Everything it cannot exclude is evaluated with the
// exact fp64 expression (pl_score.h), so counts, inlier sets and scores are unchanged; the filter only removes
// work.  The predicates are plain inline functions so that the scoring kernels (kernels.hip) and the test-only
// host build (tests/hostmath) run the very same IEEE operations; tests/test_hostmath_vs_oracle.py checks the
// "never drops an inlier" property against the oracle on adversarial inputs.
//
// Notation: u = 2^-24.  Inputs are rounded to fp32 (relative error u each); a dot product of n terms evaluated
// with FMAs carries an error below (n+2) u * sum |terms|.  All bounds below are stated for the fp64 values the
// reference computes; their own rounding (2^-53) is absorbed by the factor-2 margins.
//
//  * reprojection (absolute pose), model (R, t), point (x, y; X):  z = R X + t,  inlier => z2 > 0 and
//    |z0 - x z2| < thr z2 (same in y).  With S = |X|_2 + max|t_i| every fp32 z^_i is within 8u S of z_i and
//    a^ = fl(z^0 - x^ z^2) within 16u (1 + |x|) S of z0 - x z2;  W = 32u (1 + max|x|,|y| + thr) S covers that plus
//    the error of thr^ z^2, hence   max(|a^0|, |a^1|) > fma(thr^, z^2, W)  or  z^2 < -W   proves an outlier.
//  * homography, model H, point (a; b):  h = H (a,1),  inlier => |h0 - b0 h2| < thr |h2| (same with b1, h1).
//    With hm >= max|H_ij|, na = 1 + |a0| + |a1|, nb = 1 + |b0| + |b1| + thr:  |h^_i - h_i| <= 5u hm na and
//    a^ = fl(h^0 - b^0 h^2) is within 9u hm na nb of h0 - b0 h2;  W = 32u hm na nb, test
//    max(|a^0|, |a^1|) > fma(thr^, |h^2|, W).
//  * Sampson (fundamental / essential), model F, point (a; b):  inlier => C^2 < thr2 (Cx + Cy) with
//    C = b^T F a,  Cx + Cy = (Fa)_0^2 + (Fa)_1^2 + (F^T b)_0^2 + (F^T b)_1^2.  With fm >= max|F_ij|,
//    na = 1 + |a0| + |a1|, nb = 1 + |b0| + |b1|:  every (Fa)^_i is within e_a = 16u fm na, every (F^T b)^_j within
//    e_b = 16u fm nb (the bound is 5u) and C^ within e_C = 32u fm na nb (bound 10u) of the exact value.  By
//    (|E| + e)^2 <= (1 + d) E^2 + (1 + 1/d) e^2  with d = 1/64:
//        D_up = (1 + d) sum E^_k^2 + 130 (e_a^2 + e_b^2)  >=  Cx + Cy,      |C| >= |C^| - e_C,
//    so  |C^| > e_C  and  (|C^| - e_C)^2 > thr2 (1 + 64u) D_up  proves an outlier (the 64u absorbs the roundings
//    of the test itself).  The kernels evaluate a slightly weaker sufficient condition that needs ONE comparison and
//    no absolute value: for every a, b >= 0 and h in (0, 1),  (a - b)^2 >= (1 - h) a^2 - (1 / h - 1) b^2,  so with h = 1/16
//        C^^2  >  T1 * sum E^_k^2  +  gf^2 * w                                  (gf = 16u fm per model, w per point)
//        T1 = (16/15) (1 + d) thr2 (1 + 96u),    w = (16/15) (1 + 96u) (thr2 (1 + 64u) 130 (na^2 + nb^2) + 60 (na nb)^2)
//    implies the condition above (e_C^2 = 4 gf^2 (na nb)^2; the right-hand side is non-negative, so the inequality also
//    forces |C^| > 3.8 e_C > e_C); the extra 32u covers the roundings of C^ C^, gf^2 w and the final FMA.  The acceptance
//    band is 4 % wider than with the two-comparison form (h trades that against the weight of e_C, which matters for
//    un-normalised pixel coordinates); 20 instead of 24 operations per pair.
//  * reprojection, fp16 / MFMA form (k_score_mfma): v_mfma_f32_32x32x8_f16 evaluates, with fp32 accumulation,
//        z^_c = rn16(R_c) (X_hi + X_lo) + rn16(t_c)          X_hi = rn16(X), X_lo = rn16(X - X_hi)
//        B^   = rn16(thr R_2) (X_hi + X_lo) + up16(thr t_2 + g) + up16(w)
//    |R_ck| <= 1 gives |rn16(R_ck) - R_ck| <= 2^-12 and |rn16(t_c) - t_c| <= 2^-12 |t_c|; the hi/lo pair leaves
//    2^-22 |X| (+ 3e-8 once the low part is subnormal); products are exact in fp32 and eight accumulations add
//    <= 2^-21 (|X|_1 + |t_c|):  |z^_c - z_c| <= 2^-12 (|X|_1 + |t_c|) (1 + 2^-8) + 1e-7,  hence
//    |a^ - a| <= (1 + |x|) of that for a^ = fl(z^0 - x z^2) (fp32, exact x), and B^ >= thr z_2 - 2^-11 thr |X|_1
//    + g + w.  With  g = G max|t_c| + c,  w = G |X|_1 + 1.3e-4,  G = 2^-11 (1 + max|x|,|y| + thr),  c = 2.5e-4 (1 +
//    max|x|,|y| + thr) + 6e-5 (every factor rounded up; the absolute parts also cover fp16 subnormal inputs - X_lo
//    below |X| = 0.25, small t, small thr R_2 - being flushed to zero by the matrix pipe) the slack is twice the
//    error of a^ plus the error of B^ itself, so
//    max(|a^0|, |a^1|) > B^  proves an outlier; the "behind the camera" test is left to the exact pass.  Points or
//    translations beyond 3e4 (fp16 range) and rotation rows that are not unit-bounded get an infinite slack
//    (always evaluated exactly), NaN models -inf (never).
//  * models with a NaN entry have no inliers at all (see store_shadow); models or thresholds outside the range in
//    which fp32 keeps its relative accuracy (max-abs entry outside [1e-18, 1e18]) get an infinite slack, i.e. every
//    point is evaluated exactly.
#pragma once
#include "pl_math.h"

#include <cmath>

namespace pl {

constexpr float kPfU = 5.9604644775390625e-08f; // 2^-24

struct PrefilterArgs {
    float thr;     // sqrt(thr2), rounded up
    float gx;      // absolute pose: 32u (1 + max|x|,|y| + thr), rounded up
    float thr2_up; // Sampson: thr2 (1 + 64u), rounded up
    int enabled;   // 0: exact evaluation of every point
    float g16;     // absolute pose, fp16 / MFMA form of the filter: 2^-11 (1 + max|x|,|y| + thr), rounded up
    float c16;     //   and the absolute part (2e-7 + 2.5e-4) (1 + max|x|,|y| + thr)
    float t1;      // Sampson, one-comparison form: (16/15) (1 + 1/64) thr2 (1 + 96u), rounded up
    float w252;    //   and (16/15) (1 + 96u) 60, rounded up (factor of (na nb)^2 in the per-point term w)
};

PL_HD float pf_up(float v) { return v * 1.000001f + 1e-30f; } // pads a non-negative bound upwards

// Host side: the kernel parameters for a problem (every value rounded UP).  est: pl_score.h Estimator;
// xy_absmax: max(|x|, |y|) over the 2-D points (absolute pose only).  enabled = 0 when the threshold lies outside
// the range fp32 can carry.
inline PrefilterArgs make_prefilter_args(int est, double thr2, float xy_absmax) {
    PrefilterArgs a;
    a.thr = a.gx = a.thr2_up = a.g16 = a.c16 = a.t1 = a.w252 = 0.f;
    a.enabled = 0;
    if (!(thr2 >= 1e-30 && thr2 <= 1e30))
        return a;
    if (est == 0 /* EST_ABS */ && !(xy_absmax <= 3.0e38f))
        return a;
    const float inf = __builtin_huge_valf();
    const double thr = sqrt(thr2);
    const double u = 5.9604644775390625e-08;
    a.thr = nextafterf((float)thr, inf);
    a.thr2_up = nextafterf((float)(thr2 * (1.0 + 64.0 * u)), inf);
    a.t1 = nextafterf((float)((16.0 / 15.0) * (1.0 + 1.0 / 64.0) * thr2 * (1.0 + 96.0 * u)), inf);
    a.w252 = nextafterf((float)((16.0 / 15.0) * (1.0 + 96.0 * u) * 60.0), inf);
    if (est == 0) {
        a.gx = nextafterf((float)(32.0 * u * (1.0 + (double)xy_absmax + thr)), inf);
        a.g16 = nextafterf((float)(4.8828125e-4 * (1.0 + (double)xy_absmax + thr)), inf); // 2^-11
        // 2e-7: fp32 accumulation; 2.5e-4: four fp16 inputs per row (X_lo, t) may be subnormal, i.e. below 6.1e-5 - the
        // bound holds even if the matrix pipe flushes them to zero
        a.c16 = nextafterf((float)((2e-7 + 2.5e-4) * (1.0 + (double)xy_absmax + thr)), inf);
    }
    a.enabled = 1;
    return a;
}

// ---- per-point bound terms (computed once per kernel, kept in registers) ------------------------------------
PL_HD float pf_point_abs(double X, double Y, double Z, float gx) { // gx * upper bound of |X|_2
    return pf_up(gx * pf_up((float)sqrt(X * X + Y * Y + Z * Z)));
}
PL_HD void pf_point_two_view(double a0, double a1, double b0, double b1, float thr, float &nanb, float &nsq,
                             float &nanb_thr) {
    const float na = pf_up((float)(1.0 + fabs(a0) + fabs(a1)));
    const float nb = pf_up((float)(1.0 + fabs(b0) + fabs(b1)));
    nanb = pf_up(na * nb);                          // Sampson: e_C = 2 gf * na * nb
    nsq = pf_up(130.f * pf_up(na * na + nb * nb));  // Sampson: 130 (e_a^2 + e_b^2) = gf^2 * nsq
    nanb_thr = pf_up(na * pf_up(nb + thr));         // homography: W = gh * na * (nb + thr)
}
// Sampson, one-comparison form: the per-point term w (header comment), every factor rounded up
PL_HD float pf_point_sampson_w(float nanb, float nsq, const PrefilterArgs &pf) {
    // (16/15) (1 + 96u) thr2_up nsq <= t1 nsq  (t1 carries the larger factor (1 + 1/64) as well)
    return pf_up(pf_up(pf.t1 * nsq) + pf_up(pf.w252 * pf_up(nanb * nanb)));
}

// ---- per-model terms: shadow[12] = padded max|t_i| (absolute pose), shadow[13] = NaN flag, shadow[14] = padded
// max-abs matrix entry or +inf (pl_math.h store_shadow / model_scale_f32) ----------------------------------------

// ---- the three predicates: true = certainly not an inlier -----------------------------------------------------
// r: fp32 shadow (9 matrix entries row-major, t at 9..11)
PL_HD bool pf_abs_outlier(const float *r, float gt /* gx * tmax */, float thr, float x, float y, float X, float Y,
                          float Z, float fw) {
    const float z0 = fmaf(r[0], X, fmaf(r[1], Y, fmaf(r[2], Z, r[9])));
    const float z1 = fmaf(r[3], X, fmaf(r[4], Y, fmaf(r[5], Z, r[10])));
    const float z2 = fmaf(r[6], X, fmaf(r[7], Y, fmaf(r[8], Z, r[11])));
    const float a0 = fmaf(-x, z2, z0);
    const float a1 = fmaf(-y, z2, z1);
    const float W = fw + gt;
    const float B = fmaf(thr, z2, W);
    return (fmaxf(fabsf(a0), fabsf(a1)) > B) | (z2 < -W);
}

PL_HD bool pf_hom_outlier(const float *r, float gh /* 32u * hm */, float thr, float a0, float a1, float b0, float b1,
                          float nanb_thr) {
    const float h0 = fmaf(r[0], a0, fmaf(r[1], a1, r[2]));
    const float h1 = fmaf(r[3], a0, fmaf(r[4], a1, r[5]));
    const float h2 = fmaf(r[6], a0, fmaf(r[7], a1, r[8]));
    const float e0 = fmaf(-b0, h2, h0);
    const float e1 = fmaf(-b1, h2, h1);
    const float W = gh * nanb_thr;
    const float B = fmaf(thr, fabsf(h2), W);
    return fmaxf(fabsf(e0), fabsf(e1)) > B;
}

// w: pf_point_sampson_w of the correspondence; t1: PrefilterArgs.t1
PL_HD bool pf_sampson_outlier(const float *r, float gf /* 16u * fm */, float t1, float a0, float a1, float b0,
                              float b1, float w) {
    const float Ea0 = fmaf(r[0], a0, fmaf(r[1], a1, r[2]));
    const float Ea1 = fmaf(r[3], a0, fmaf(r[4], a1, r[5]));
    const float Ea2 = fmaf(r[6], a0, fmaf(r[7], a1, r[8]));
    const float Eb0 = fmaf(r[0], b0, fmaf(r[3], b1, r[6]));
    const float Eb1 = fmaf(r[1], b0, fmaf(r[4], b1, r[7]));
    const float C = fmaf(b0, Ea0, fmaf(b1, Ea1, Ea2));
    const float S = fmaf(Eb1, Eb1, fmaf(Eb0, Eb0, fmaf(Ea1, Ea1, Ea0 * Ea0)));
    return C * C > fmaf(t1, S, (gf * gf) * w);
}

} // namespace pl
