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
//  * reprojection, fp16 / MFMA form (k_score_mfma): round 6 tests THREE half-planes of a triangle around the inlier disc and
//    carries the slack differently - see "Round 6" at pf16_abs_model below, which builds on this derivation of the round-2 form
//    (kept for reference): the test was four half-planes per pair,
//        F(-+, a) = thr z_2 -+ (z_a - p z_2) + slack >= 0        a = 0, 1;  p = x (a = 0) or y (a = 1)
//    and each is LINEAR in sixteen numbers of the correspondence, so v_mfma_f32_32x32x16_f16 evaluates, with fp32
//    accumulation,
//        F^ = rn16(thr R_2 -+ R_a) (X_hi + X_lo) + up16(thr t_2 -+ t_a + g) + up16(w)
//             +- rn16(R_2) ((p X)_hi + (p X)_lo) +- rn16(t_2) rn16(p)
//    (X_hi = rn16(X), X_lo = rn16(X - X_hi); the products p X are formed in fp64 and split the same way; p itself enters
//    with its high part only).  |R_ck| <= 1 and thr <= 1 give coefficients bounded by 1 + thr, rounded with relative error
//    2^-11 (the unit roundoff of fp16; round 2 wrote 2^-12 here and had no margin left for the terms that follow - ADVICE r2);
//    the splits leave 2^-22 (+ 6.1e-5 absolute once a part is subnormal - also when the matrix pipe flushes it); products
//    are exact in fp32 and sixteen accumulations add <= 2^-19 of the sum of magnitudes:
//        |F^ - F| <= (2^-11 + 2^-19 + 2^-21) (1 + thr + |p|) |X|_1  +  2^-11 (2 |p| + (1 + thr)) max|t_c| (1 + 2^-8)  +  absolute terms
//    (the 2 |p|: rounded t_2 and the dropped low part of p).  With  g = G max|t_c| + c,  w = G |X|_1 + 1.3e-4,
//    G = 2^-11 (1 + 2^-6) (1 + max|x|,|y| + thr),  c = 4e-4 (1 + max|x|,|y| + thr) + 6e-5 (every factor rounded up): the
//    factor 1 + 2^-6 of G pays for the accumulation and split terms of the X part (2^-8 + 2^-10 of the leading term) whatever
//    the field of view and the threshold; the translation part needs 2 |p| + 1 + thr <= 2 (1 + max|p| + thr) - i.e. G covers
//    it with a factor of two to spare -, and the absolute parts cover the subnormal operands: six low parts and p per row,
//    and coefficients below 6.1e-5.  The constant is rounded towards +inf, so
//    F^ >= F:  a NEGATIVE F^ proves |z_a - p z_2| > thr z_2, i.e. an outlier (z_2 <= 0: not an inlier either way).  The
//    kernel ORs the four sign bits.  Points or translations beyond 3e4 (fp16 range; also |p| |X|_1) and rotation rows that
//    are not unit-bounded get an infinite slack (always evaluated exactly), NaN models -inf (never).
//  * Sampson, fp16 / MFMA form (k_score_mfma2; two-view problems whose coordinates are bounded by 8, i.e. normalised image
//    points): the matrix pipe evaluates the two FORMS of the test directly, for 32 models x 32 correspondences per
//    instruction (v_mfma_f32_32x32x16_f16, fp32 accumulation):
//        C~ = sum_ij F~_ij phi~_ij          F~ = F cF, cF the power of two with max|F~_ij| in [2^9, 2^10),  phi~_ij = 2^8 b_i a_j
//        S~ = sum_k g~_k psi~_k             g~ = 2^-7 coefficients of  a^T G1 a + b^T G2 b,  G1 = F~_0 F~_0^T + F~_1 F~_1^T (rows),
//                                           G2 likewise from the first two columns;  psi~ = 2^8 (a0^2, a0 a1, a1^2, a0, a1, 1, b0^2, ...)
//    so C~ = 2^8 cF C and S~ = 2 cF^2 (Cx + Cy):  C^2 > T (Cx + Cy)  <=>  C~^2 > 2^15 T S~.  C~ has to resolve a cancellation
//    (|C| << sum |F_ij phi_ij| near the epipolar line): both operands are split into fp16 high / low parts and three of the
//    four partial products are kept (27 of the 32 k slots: hi*hi, hi*lo, lo*hi); S~ is a sum of squares and runs on the
//    high parts only (11 slots + one slot that adds the slack).  Error bounds, in the scaled units, per correspondence only
//    (|F~_ij| <= 2^10 whatever the model):
//      - split: |v - hi - lo| <= 2^-22 |v| (+ 2^-14 absolute once a part is below the fp16 normal range - that also covers a
//        matrix pipe that flushes subnormal inputs); the dropped lo*lo product is <= 2^-22 |F~ phi~|; products of two fp16
//        numbers are exact in fp32; 32 accumulations at <= 2^-23 each (truncation assumed, not round-to-nearest):
//            |C^ - C~| <= 2^-17 sum |F~ phi~| + 2^-14 (sum |F~| + sum |phi~|) <= E_C = 2.1 na nb + 0.6
//        (sum_ij |phi_ij| = na nb exactly, na = 1 + |a0| + |a1|, nb = 1 + |b0| + |b1|)
//      - S^: both factors rounded to fp16 (2^-11 each), 12 accumulations:
//            |S^ - S~| <= 2^-9.9 sum |g~ psi~| + 2^-14 (sum |g~| + sum |psi~|) <= E_S = 4401 (na^2 + nb^2) + 24
//        (|g~_k| <= 2^15, sum over the 11 monomials with their multiplicities <= 2^22 (na^2 + nb^2))
//    With (|C^| - E_C)^2 >= (15/16) C^^2 - 15 E_C^2:   C^^2 > t16 (S^ + w),  w >= E_S + 16 E_C^2 / t16,
//    t16 = (16/15) 2^15 thr2 (1 + 256u)  implies  C~^2 > 2^15 thr2 (1 + 256u) S~, i.e. a certain outlier (the 256u absorb the
//    fp32 roundings of C^ C^ and of the final FMA and the reference's own fp64 roundings).  w rides in the twelfth k slot
//    (model side 2^14, correspondence side w 2^-14 rounded UP to fp16, a normal number), so the kernel's tail per pair is
//    one multiplication and one FMA whose SIGN is the verdict.  Model side of that slot = -inf: NaN model (no inliers);
//    +inf: matrix outside [1e-18, 1e18] (every point evaluated exactly).  Correspondences with a coordinate beyond 8 (or
//    NaN) carry zero operands and the largest finite slack: never excluded.
//  * homography, fp16 / MFMA form (k_score_mfmah, round 3; coordinates bounded by 8, thr <= 8): an inlier satisfies
//    |h_c - b_c h_2| < thr |h_2| for c = 0, 1 (h = H (a, 1)).  |h_2| has no fixed sign over an image - for 55 % of the
//    4-point hypotheses of BASELINE config 3 the vanishing line crosses the correspondences' bounding box, with 20 % of them
//    on the minority side - so the inlier region is the union of two opposite cones and no set of half-planes (the P3P form)
//    can bound it.  The matrix pipe delivers, per (hypothesis, correspondence), FOUR linear forms of the nine monomials
//    phi = 2^8 (a0, a1, 1, b0 a0, b0 a1, b0, b1 a0, b1 a1, b1)  of the correspondence and a slack - two chained
//    v_mfma_f32_32x32x16_f16 (32 k slots) for 8 hypotheses x 32 correspondences:
//        V_c = H~_c . phi_a - H~_2 . phi_(b_c a)      ( = 2^8 cH (h_c - b_c h_2),  cH the power of two with max|H~| in [2^9, 2^10) )
//        U   = thr H~_2 . phi_a                       ( = 2^8 cH thr h_2 )
//        S   = g w + wa                               ( g = 2^-16.5 max|H~| per hypothesis (stored x 4);  w = 2^8 na (nb + thr)
//                                                       (stored / 4), wa = 2^-5 na (nb + 1) + 3.8 per correspondence; rounded UP )
//    and the vector ALU keeps the sign of  (|U| + S) - max(|V_0|, |V_1|)  (4 instructions per pair: v_med3 with |.| modifiers,
//    add, subtract, v_alignbit).  The first cut of this form rounded every factor to ONE fp16 number: an error of 2^-10 of the
//    sum of magnitudes, i.e. a band of 16 thresholds at thr = 1 px / f = 1000 - and the hypotheses of a RANSAC run are not
//    random matrices: a homography through two or three inliers maps hundreds of correspondences to within a few pixels,
//    which all survived (894 survivors per hypothesis against 211 of the fp32 form; measured slower than the fp32 form).  So
//    both factors are split into fp16 high / low parts like the Sampson form's C~, and three of the four partial products are
//    kept: k slots 0..8 hi*hi, 9..17 lo(coefficient)*hi, 18..26 hi*lo(monomial), 27 / 28 the slack.  Error budget: the dropped
//    lo*lo product and the two splits <= 3 * 2^-22 |alpha phi|, products exact in fp32, 29 accumulations at <= 2^-23 each
//    (truncation assumed): 2^-17.86 of the sum of magnitudes; operands below the fp16 normal range cost <= 2^-14 of the
//    other factor each (also when the matrix pipe flushes them - low parts of small coefficients are subnormal):
//        |V^_c - V_c| <= 2^-17.86 2^8 Hm na nb'  +  2^-14 (2 + 2^-11) (2^8 na nb' + 6 * 2^10)     (nb' = 1 + max |b_c|)
//        |U^ - U|     <= 2^-17.86 2^8 thr Hm na  +  2^-14 (2 + 2^-11) (2^8 na + 3 * 2^10 thr)     (thr <= 8)
//    g w = 2^-16.5 2^8 Hm na (nb' + thr) covers the relative parts 2.5 times over (which also pays for the fp32 roundings of
//    the tail), wa >= 2^-5 (1 + 2^-10) na (nb' + 1) + 0.7503 + 3.0015 the absolute ones.  max |V^_c| > |U^| + S  then implies
//    max |V_c| > |U|, a certain outlier; h_2 = 0 (never an inlier: the reference divides by it) needs no special case.  At
//    thr = 1e-3 on normalised coordinates the band is 1.1 thresholds wide.  Model side of the slack slot = -inf: NaN model (no
//    inliers); +inf: matrix outside [1e-18, 1e18] (every point evaluated exactly).  Correspondences with a coordinate beyond
//    8 (or NaN) carry zero operands and the largest finite slack: never excluded.
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
    float g16;     // absolute pose, fp16 / MFMA form of the filter: 2^-11 (1 + 2^-6), rounded up (0: form not available)
    float c16;     //   and the absolute part (2e-7 + 5e-4) (1 + 1.3661 max|x|,|y| + thr)
    float t1;      // Sampson, one-comparison form: (16/15) (1 + 1/64) thr2 (1 + 96u), rounded up
    float w252;    //   and (16/15) (1 + 96u) 60, rounded up (factor of (na nb)^2 in the per-point term w)
    float t16;     // Sampson, fp16 / MFMA form: (16/15) 2^15 thr2 (1 + 256u), rounded up; 0 = that form is not available
    float h16;     // homography, fp16 / MFMA form: the threshold, rounded up; 0 = that form is not available
};

PL_HD float pf_up(float v) { return v * 1.000001f + 1e-30f; } // pads a non-negative bound upwards

// Host side: the kernel parameters for a problem (every value rounded UP).  est: pl_score.h Estimator;
// xy_absmax: max(|x|, |y|) over the 2-D points (absolute pose only).  enabled = 0 when the threshold lies outside
// the range fp32 can carry.
inline PrefilterArgs make_prefilter_args(int est, double thr2, float xy_absmax) {
    PrefilterArgs a;
    a.thr = a.gx = a.thr2_up = a.g16 = a.c16 = a.t1 = a.w252 = a.t16 = a.h16 = 0.f;
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
        // fp16 / MFMA form (round 6: three directions, p = c x + s y with |p| <= 1.3661 max(|x|, |y|); pf16_abs_model / _point)
        a.g16 = nextafterf((float)(4.8828125e-4 * 1.015625), inf); // 2^-11 (1 + 2^-6)
        // absolute part: operands below the fp16 normal range (also when the matrix pipe flushes them), fp32 accumulation
        a.c16 = nextafterf((float)((2e-7 + 5.0e-4) * (1.0 + 1.3661 * (double)xy_absmax + thr)), inf);
    }
    if (est == 1 || est == 2) {
        // xy_absmax of a two-view problem: max over all four coordinates (driver.cc make_problem; +inf when unknown).
        // thr2 range: t16 and 16 E_C^2 / t16 have to stay inside fp32 / fp16 with room to spare
        if (xy_absmax <= 8.0f && thr2 >= 1e-12 && thr2 <= 1e4)
            a.t16 = nextafterf((float)((16.0 / 15.0) * 32768.0 * thr2 * (1.0 + 256.0 * u)), inf);
    }
    if (est == 3) {
        // homography on the matrix cores: coordinates bounded by 8 (xy_absmax of a two-view problem, see above), thr <= 8
        // (the coefficient thr H~_2j has to stay inside fp16) and not absurdly small
        if (xy_absmax <= 8.0f && thr >= 1e-9 && thr <= 8.0)
            a.h16 = a.thr;
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

// ---- Sampson, fp16 / MFMA form: operands (header comment).  Portable fp16 conversions: the kernels, the operand builder
// of the hypotheses (pipeline.hip) and the test-only host build run the same integer code ------------------------------
PL_HD uint16_t pf_half_rn(float f) { // round to nearest even; overflow -> inf
#if defined(__HIP_DEVICE_COMPILE__)
    // the conversion instruction does exactly this (v_cvt_f16_f32: IEEE round-to-nearest-even, subnormal results kept);
    // the integer form below is what the host test build runs, checked bit for bit against IEEE half on 4e5 values
    const _Float16 h = (_Float16)f;
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
#endif
    uint32_t x;
    __builtin_memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u)
        return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) // >= 65520
        return (uint16_t)(sign | 0x7c00u);
    if (x < 0x38800000u) { // below 2^-14: multiples of 2^-24
        float af;
        __builtin_memcpy(&af, &x, 4);
        const float scaled = af * 16777216.0f;                // exact
        const float r = (scaled + 8388608.0f) - 8388608.0f;   // nearest-even integer, 0 <= scaled < 2^10
        return (uint16_t)(sign | (uint32_t)r);
    }
    x += 0x0fffu + ((x >> 13) & 1u);
    return (uint16_t)(sign | ((x - 0x38000000u) >> 13));
}
PL_HD float pf_half_to_float(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
#endif
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t x;
    if (e == 0) {
        const float v = (float)m * 5.9604644775390625e-08f;
        __builtin_memcpy(&x, &v, 4);
        x |= sign;
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    __builtin_memcpy(&f, &x, 4);
    return f;
}
PL_HD uint16_t pf_half_up(float v) { // v >= 0: the smallest fp16 >= v (+inf beyond 65504)
    uint16_t b = pf_half_rn(v);
    if (b < 0x7c00u && pf_half_to_float(b) < v)
        b = (uint16_t)(b + 1);
    return b;
}
PL_HD void pf_split16(double v, uint16_t &hi, uint16_t &lo) { // |v| < 65504
    hi = pf_half_rn((float)v);
    lo = pf_half_rn((float)(v - (double)pf_half_to_float(hi)));
}

struct Sampson16Operand { // one model or one correspondence: k slots of the three matrix instructions
    uint16_t c[32];       // C~: slots 0..8 / 9..17 / 18..26, 27..31 = 0
    uint16_t s[16];       // S~: slots 0..10, slack in 11, 12..15 = 0
};
constexpr double kS16PointMax = 8.0; // largest |coordinate| the fp16 operands carry

// model side.  F: row-major 3x3; nan_model: the record's NaN flag (pl_math.h store_shadow)
PL_HD void pf16_sampson_model(const double *F, bool nan_model, Sampson16Operand &o) {
    for (int k = 0; k < 32; ++k)
        o.c[k] = 0;
    for (int k = 0; k < 16; ++k)
        o.s[k] = 0;
    double fm = 0.0;
    bool bad = nan_model;
    for (int k = 0; k < 9; ++k) {
        const double a = fabs(F[k]);
        bad |= (a != a);
        fm = a > fm ? a : fm;
    }
    if (bad) {
        o.s[11] = 0xfc00u; // -inf: no inliers
        return;
    }
    if (!(fm >= 1e-18 && fm <= 1e18)) {
        o.s[11] = 0x7c00u; // +inf: every correspondence is evaluated exactly
        return;
    }
    int e;
    (void)frexp(fm, &e);                  // fm in [2^(e-1), 2^e)
    const double cF = ldexp(1.0, 10 - e); // max |F~| in [2^9, 2^10)
    double Ft[9];
    for (int k = 0; k < 9; ++k) {
        Ft[k] = F[k] * cF;
        uint16_t h, l;
        pf_split16(Ft[k], h, l);
        o.c[k] = h, o.c[9 + k] = h, o.c[18 + k] = l;
    }
    // G1 from rows 0, 1 (quadratic form in a), G2 from columns 0, 1 (in b)
    const double *r0 = Ft, *r1 = Ft + 3;
    const double c0[3] = {Ft[0], Ft[3], Ft[6]}, c1[3] = {Ft[1], Ft[4], Ft[7]};
    auto g1 = [&](int i, int j) { return r0[i] * r0[j] + r1[i] * r1[j]; };
    auto g2 = [&](int i, int j) { return c0[i] * c0[j] + c1[i] * c1[j]; };
    const double sc = 0.0078125; // 2^-7
    const double g[11] = {g1(0, 0) * sc, 2.0 * g1(0, 1) * sc, g1(1, 1) * sc, 2.0 * g1(0, 2) * sc, 2.0 * g1(1, 2) * sc,
                          (g1(2, 2) + g2(2, 2)) * sc,
                          g2(0, 0) * sc, 2.0 * g2(0, 1) * sc, g2(1, 1) * sc, 2.0 * g2(0, 2) * sc, 2.0 * g2(1, 2) * sc};
    for (int k = 0; k < 11; ++k)
        o.s[k] = pf_half_rn((float)g[k]);
    o.s[11] = 0x7400u; // 2^14
}

// correspondence side.  Returns false (zero operands, largest finite slack) for points the operands cannot carry.
PL_HD bool pf16_sampson_point(double a0, double a1, double b0, double b1, bool valid, float t16, Sampson16Operand &o) {
    for (int k = 0; k < 32; ++k)
        o.c[k] = 0;
    for (int k = 0; k < 16; ++k)
        o.s[k] = 0;
    const double m = fmax(fmax(fabs(a0), fabs(a1)), fmax(fabs(b0), fabs(b1)));
    const bool finite = (a0 == a0) & (a1 == a1) & (b0 == b0) & (b1 == b1);
    if (!(valid && finite && m <= kS16PointMax)) {
        o.s[11] = 0x7bffu; // 65504
        return false;
    }
    const double al[3] = {a0, a1, 1.0}, be[3] = {b0, b1, 1.0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            uint16_t h, l;
            pf_split16(be[i] * al[j] * 256.0, h, l);
            o.c[3 * i + j] = h, o.c[9 + 3 * i + j] = l, o.c[18 + 3 * i + j] = h;
        }
    const double psi[11] = {a0 * a0, a0 * a1, a1 * a1, a0, a1, 1.0, b0 * b0, b0 * b1, b1 * b1, b0, b1};
    for (int k = 0; k < 11; ++k)
        o.s[k] = pf_half_rn((float)(psi[k] * 256.0));
    const double na = 1.0 + fabs(a0) + fabs(a1), nb = 1.0 + fabs(b0) + fabs(b1);
    const double EC = 2.1 * na * nb + 0.6;
    const double ES = 4401.0 * (na * na + nb * nb) + 24.0;
    const double w = (ES + 16.0 * EC * EC / (double)t16) * 1.000001;
    o.s[11] = pf_half_up((float)(w * 6.103515625e-05) * 1.000001f); // 2^-14; >= 2^-14 * 8826: a normal fp16
    return true;
}

// The verdict as the kernel computes it, with one particular accumulation order (the bound holds for any order):
// true = certainly not an inlier.  Test-only host build and documentation of the device tail.
PL_HD bool pf16_sampson_outlier(const Sampson16Operand &model, const Sampson16Operand &point, float t16) {
    float C = 0.f, S = 0.f;
    for (int k = 0; k < 32; ++k)
        C = fmaf(pf_half_to_float(model.c[k]), pf_half_to_float(point.c[k]), C);
    for (int k = 0; k < 16; ++k)
        S = fmaf(pf_half_to_float(model.s[k]), pf_half_to_float(point.s[k]), S);
    const float d = fmaf(t16, S, -(C * C));
    uint32_t bits;
    __builtin_memcpy(&bits, &d, 4);
    return (bits >> 31) != 0u;
}

// ---- reprojection, fp16 / MFMA form: operands (header comment).  The kernels (k_shadow16, k_score_mfma) and the test-only
// host build run these very functions ---------------------------------------------------------------------------------
PL_HD uint16_t pf_half_toward_plus_inf(float v) { // any sign: the smallest fp16 >= v
    uint16_t b = pf_half_rn(v);
    if ((b & 0x7fffu) < 0x7c00u && pf_half_to_float(b) < v) {
        b = (b & 0x8000u) ? (uint16_t)(b - 1) : (uint16_t)(b + 1);
        if (b == 0x8000u)
            b = 0; // -0 -> +0
    }
    return b;
}
// Round 6: THREE half-planes instead of four, and the slack where it arises.  An inlier's residual vector
// a = (z_0 - x z_2, z_1 - y z_2) has |a|_2 < thr z_2 (utils.cc:36-65: the squared reprojection error against thr^2, z_2 > 0), so
// d . a < thr z_2 for EVERY direction |d|_2 <= 1: the three directions below (0, 120, 240 degrees; the second component rounded
// DOWN, |d| < 1) bound the inlier disc by a triangle instead of the axis-parallel square.  The triangle's area is 1.3 x the
// square's - 30 % more pairs reach the exact pass - but a pair costs the matrix pipe three single-row products (32 hypotheses
// per instruction) and the vector ALU one v_or3 and one v_alignbit instead of two products of two rows (16 hypotheses) and three
// vector instructions.  Per direction (c, s), with p = c x + s y, R' = c R_0 + s R_1 (entries bounded by |c| + |s| <= 1.3661; by
// 1 for orthonormal rows) and t' = c t_0 + s t_1, the exact form and what the matrix pipe accumulates in fp32 are
//     F  = (thr R_2 - R') . X + (thr t_2 - t') + p (R_2 . X) + p t_2                                   ( > 0 for an inlier )
//     F^ = rn16(thr R_2 - R') (X_hi + X_lo) + up16(thr t_2 - t' + g) + up16(w) + rn16(R_2) ((p X)_hi + (p X)_lo)
//          + rn16(t_2) rn16(p) + up16(Tm) up16(Pa)
// (the last product rides in the sixteenth k slot, unused until round 6).  Errors, with u16 = 2^-11 (fp16 unit roundoff), splits
// exact to 2^-22, sixteen fp32 accumulations <= 2^-19 of the sum of magnitudes, fp32 roundings of the coefficients <= 2^-22:
//     X part:            (u16 + 2^-19 + 2^-21) (1.3661 + thr + |p|) |X|_1       <=  w  = u16 (1 + 2^-6) (1.3661 + thr + pm) |X|_1 + 1.3e-4,
//                                                                                   pm >= |p| of every direction: |(x, y)|_2 rounded up
//     rn16(t_2) rn16(p): 2 u16 (1 + 2^-12) |t_2| |p|  (+ 2^-19 of it)            <=  Tm Pa,  Tm = 2 u16 (1 + 2^-6) max|t_c|,  Pa = |p|
//     constant:          rounded UP; its fp32 roundings and 2^-19 of |constant|  <=  g  = 2^-16 max|t_c| + c
// - round 2 .. 5 charged the translation part as G max|t_c| with G = u16 (1 + 2^-6) (1 + max|x|,|y| + thr), which covers
// 2 u16 |p| |t_2| only while 2 |p| <= 1 + max|x|,|y| + thr, i.e. not in the corners of a field of view beyond 90 degrees; the
// bilinear slot is exact in |p| and tighter everywhere else.  Operands below the fp16 normal range (the matrix pipe may flush
// them): X_hi / X_lo and (p X)_hi / (p X)_lo cost <= 6.1e-5 of their coefficient each (three + three per row), the constant
// and rn16(t_2) one unit each (x 1 resp. x |p|); coefficients below 6.1e-5 carry no rounding error, so the budget of the X part
// pays for them; together <= 6.1e-5 (8.2 + 3 thr + max|p|) <= c = 5e-4 (1 + 1.3661 max|x|,|y| + thr).  |p| < 2^-14 enters as
// rn16(p) = 0 and Pa = 2^-4 (Tm Pa >= |t_2| 2^-14 >= the dropped product); Tm and Pa are never below 2^-14.
// F^ >= F then: a NEGATIVE F^ of any direction proves an outlier.
constexpr int kAbs16Dirs = 3;
PL_HD float pf16_abs_dir_c(int k) { return k == 0 ? 1.f : -0.5f; }
PL_HD float pf16_abs_dir_s(int k) { return k == 0 ? 0.f : (k == 1 ? 0.8660254f : -0.8660254f); } // (0.8660254f < sqrt(3) / 2)
struct Abs16Model {      // rows of one hypothesis
    uint16_t d[kAbs16Dirs][8]; // first k block of direction k: (c_0, c_1, c_2, c_0, c_1, c_2, const, 1)
    uint16_t b1[8];            // second k block (every direction): (R_20, R_21, R_22, R_20, R_21, R_22, t_2, Tm)
};
// shadow: the record's fp32 shadow (R row-major at 0..8, t at 9..11, padded max|t_c| at 12, NaN flag at 13) or nullptr for
// a row that is not a hypothesis; c16: PrefilterArgs.c16
PL_HD void pf16_abs_model(const float *shadow, float c16, float thr, Abs16Model &o) {
    const float inf = __builtin_huge_valf();
    float R[9], t[3], slack, tm = 0.f;
    for (int i = 0; i < 9; ++i)
        R[i] = 0.f;
    t[0] = t[1] = t[2] = 0.f;
    if (!shadow) {
        slack = -inf; // never a candidate
    } else {
        float rmax = 0.f;
        for (int i = 0; i < 9; ++i)
            rmax = fmaxf(rmax, fabsf(shadow[i]));
        const float tmax = shadow[12];
        uint32_t nanflag;
        __builtin_memcpy(&nanflag, &shadow[13], 4);
        if (nanflag != 0u) {
            slack = -inf; // NaN model: no inliers
        } else if (!(tmax < 2.7e4f) || !(rmax <= 1.0001f)) { // (2.7e4: |thr t_2 - t'| <= 2.37 max|t_c| stays below 65504)
            slack = inf; // outside what fp16 carries: every point is evaluated exactly
        } else {
            for (int i = 0; i < 9; ++i)
                R[i] = shadow[i];
            for (int i = 0; i < 3; ++i)
                t[i] = shadow[9 + i];
            slack = fmaf(1.52587890625e-5f, tmax, c16) * 1.000001f + 6.2e-5f; // 2^-16 max|t_c| + c
            tm = fmaxf(9.918212890625e-4f * tmax * 1.000001f, 6.103515625e-5f); // 2^-10 (1 + 2^-6) max|t_c|, at least 2^-14
        }
    }
    const bool finite = slack != inf && slack != -inf;
    for (int k = 0; k < kAbs16Dirs; ++k) {
        const float c = pf16_abs_dir_c(k), s = pf16_abs_dir_s(k);
        uint16_t *row = o.d[k];
        for (int d = 0; d < 3; ++d)
            row[d] = row[3 + d] = pf_half_rn(fmaf(thr, R[6 + d], -fmaf(c, R[d], s * R[3 + d])));
        row[6] = finite ? pf_half_toward_plus_inf(fmaf(thr, t[2], -fmaf(c, t[0], s * t[1])) + slack) : (uint16_t)(slack > 0 ? 0x7c00u : 0xfc00u);
        row[7] = 0x3c00u; // 1.0
    }
    for (int d = 0; d < 3; ++d)
        o.b1[d] = o.b1[3 + d] = pf_half_rn(R[6 + d]);
    o.b1[6] = pf_half_rn(t[2]);
    o.b1[7] = finite ? pf_half_up(tm) : (uint16_t)0;
}
struct Abs16Point {
    uint16_t b0[8];             // (X_hi, X_lo, 1, w)
    uint16_t bp[kAbs16Dirs][8]; // direction k: ((p X)_hi, (p X)_lo, p, Pa),  p = c_k x + s_k y
};
// g16: PrefilterArgs.g16 = 2^-11 (1 + 2^-6), thr: PrefilterArgs.thr.  Returns false (zero operands, largest finite slack) for
// correspondences the operands cannot carry
PL_HD bool pf16_abs_point(double x, double y, double X, double Y, double Z, bool valid, float g16, float thr, Abs16Point &o) {
    const double n1 = fabs(X) + fabs(Y) + fabs(Z);
    const double pm = fabs(x) + fabs(y); // >= |p| of every direction
    const bool use = valid && n1 < 3.0e4 && pm * n1 < 3.0e4 && pm < 3.0e4;
    const double P[3] = {X, Y, Z};
    for (int d = 0; d < 3; ++d)
        pf_split16(use ? P[d] : 0.0, o.b0[d], o.b0[3 + d]);
    for (int k = 0; k < kAbs16Dirs; ++k) {
        const double p = use ? (double)pf16_abs_dir_c(k) * x + (double)pf16_abs_dir_s(k) * y : 0.0;
        for (int d = 0; d < 3; ++d)
            pf_split16(p * P[d], o.bp[k][d], o.bp[k][3 + d]);
        // the factor of t_2: below 2^-14 it enters as zero and the slack slot pays for the dropped product
        const bool tiny = !(fabs(p) >= 6.103515625e-5);
        o.bp[k][6] = tiny ? (uint16_t)0 : pf_half_rn((float)p);
        o.bp[k][7] = use ? pf_half_up(tiny ? 0.0625f : (float)fabs(p) * 1.000001f) : (uint16_t)0;
    }
    // the point's share of the slack, rounded up (out of range: the largest finite fp16, not +inf - hypotheses with
    // |t| >= 3e4 carry an infinite slack of their own, so 65504 exceeds every |a| a zero operand can produce; thr <= 1)
    const float p2 = pf_up((float)sqrt(x * x + y * y)); // >= |p| of every direction (|d| <= 1)
    const float wv = use ? fminf(pf_up(g16 * pf_up(1.3661f + thr + p2) * pf_up((float)n1)) + 1.3e-4f, 65504.f) : 65504.f;
    o.b0[6] = 0x3c00u;
    o.b0[7] = pf_half_up(wv);
    return use;
}
// The verdict as the kernel computes it (one particular accumulation order; `order` permutes it): true = certainly not an
// inlier.  Test-only host build and documentation of the device tail.
PL_HD bool pf16_abs_outlier(const Abs16Model &m, const Abs16Point &p, const int *order /* 16 slots or nullptr */) {
    uint32_t sign = 0;
    for (int a = 0; a < kAbs16Dirs; ++a) {
        float acc = 0.f;
        for (int q = 0; q < 16; ++q) {
            const int k = order ? order[q] : q;
            const uint16_t av = k < 8 ? m.d[a][k] : m.b1[k - 8];
            const uint16_t bv = k < 8 ? p.b0[k] : p.bp[a][k - 8];
            acc = fmaf(pf_half_to_float(av), pf_half_to_float(bv), acc);
        }
        uint32_t bits;
        __builtin_memcpy(&bits, &acc, 4);
        sign |= bits;
    }
    return (sign >> 31) != 0u;
}


// ---- homography, fp16 / MFMA form: operands (header comment).  Monomials m = (a0, a1, 1, b0 a0, b0 a1, b0, b1 a0, b1 a1, b1);
// k slots 0..8: coefficient_hi * m_hi, 9..17: coefficient_lo * m_hi, 18..26: coefficient_hi * m_lo, 27: slack (relative
// part), 28: slack (absolute part), 29..31 unused.  Rows of a hypothesis: V_0, V_1, U, S ------------------------------------
struct Hom16Model {
    uint16_t r[4][32];
};
struct Hom16Point {
    uint16_t k[32];
};
constexpr double kH16PointMax = 8.0;
// H: row-major 3x3; nan_model: the record's NaN flag; thr: PrefilterArgs.h16
PL_HD void pf16_hom_model(const double *H, bool nan_model, float thr, Hom16Model &o) {
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 32; ++k)
            o.r[r][k] = 0;
    double hm = 0.0;
    bool bad = nan_model;
    for (int k = 0; k < 9; ++k) {
        const double a = fabs(H[k]);
        bad |= (a != a);
        hm = a > hm ? a : hm;
    }
    if (bad) {
        o.r[3][28] = 0xfc00u; // -inf (times the correspondence's positive wa): no inliers
        return;
    }
    if (!(hm >= 1e-18 && hm <= 1e18)) {
        o.r[3][28] = 0x7c00u; // +inf: every correspondence is evaluated exactly
        return;
    }
    int e;
    (void)frexp(hm, &e);                  // hm in [2^(e-1), 2^e)
    const double cH = ldexp(1.0, 10 - e); // max |H~| in [2^9, 2^10)
    // coefficient of monomial m in row r (zero where the row does not use it)
    double coef[3][9];
    for (int m = 0; m < 9; ++m)
        coef[0][m] = coef[1][m] = coef[2][m] = 0.0;
    for (int j = 0; j < 3; ++j) {
        coef[0][j] = H[j] * cH;          // V_0 = H~_0 . (a, 1) - b0 H~_2 . (a, 1)
        coef[0][3 + j] = -(H[6 + j] * cH);
        coef[1][j] = H[3 + j] * cH;      // V_1 = H~_1 . (a, 1) - b1 H~_2 . (a, 1)
        coef[1][6 + j] = -(H[6 + j] * cH);
        coef[2][j] = (double)thr * (H[6 + j] * cH); // U = thr H~_2 . (a, 1)
    }
    for (int r = 0; r < 3; ++r)
        for (int m = 0; m < 9; ++m) {
            uint16_t h, l;
            pf_split16(coef[r][m], h, l);
            o.r[r][m] = h, o.r[r][9 + m] = l, o.r[r][18 + m] = h;
        }
    o.r[3][27] = pf_half_up((float)(hm * cH * 4.3158e-5) * 1.000001f); // 4 * 2^-16.5 (the correspondence carries w / 4)
    o.r[3][28] = 0x3c00u;                                              // 1.0
}
// returns false (zero operands, largest finite slack) for correspondences the operands cannot carry
PL_HD bool pf16_hom_point(double a0, double a1, double b0, double b1, bool valid, float thr, Hom16Point &o) {
    for (int k = 0; k < 32; ++k)
        o.k[k] = 0;
    const double m = fmax(fmax(fabs(a0), fabs(a1)), fmax(fabs(b0), fabs(b1)));
    const bool finite = (a0 == a0) & (a1 == a1) & (b0 == b0) & (b1 == b1);
    if (!(valid && finite && m <= kH16PointMax)) {
        o.k[28] = 0x7bffu; // 65504 (times the hypothesis' 1.0)
        return false;
    }
    const double al[3] = {a0, a1, 1.0}, be[3] = {1.0, b0, b1};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            uint16_t h, l;
            pf_split16(be[i] * al[j] * 256.0, h, l);
            o.k[3 * i + j] = h, o.k[9 + 3 * i + j] = h, o.k[18 + 3 * i + j] = l;
        }
    const double na = 1.0 + fabs(a0) + fabs(a1), nb = 1.0 + fmax(fabs(b0), fabs(b1));
    o.k[27] = pf_half_up((float)(64.0 * na * (nb + (double)thr) * 1.000001) * 1.000001f); // w / 4 <= 18496
    o.k[28] = pf_half_up((float)((0.03125 * na * (nb + 1.0) * 1.001 + 3.8) * 1.000001) * 1.000001f);
    return true;
}
// The verdict as the kernel computes it (one particular accumulation order; `order` permutes it): true = certainly not an
// inlier.  Test-only host build and documentation of the device tail.
PL_HD bool pf16_hom_outlier(const Hom16Model &m, const Hom16Point &p, const int *order /* 32 slots or nullptr */) {
    float acc[4];
    for (int r = 0; r < 4; ++r) {
        acc[r] = 0.f;
        for (int q = 0; q < 32; ++q) {
            const int k = order ? order[q] : q;
            acc[r] = fmaf(pf_half_to_float(m.r[r][k]), pf_half_to_float(p.k[k]), acc[r]);
        }
    }
    const float t = fmaxf(fabsf(acc[0]), fabsf(acc[1]));
    const float d = (fabsf(acc[2]) + acc[3]) - t;
    uint32_t bits;
    __builtin_memcpy(&bits, &d, 4);
    return (bits >> 31) != 0u;
}

} // namespace pl
