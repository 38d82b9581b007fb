// poselib_amd - relative pose of two views with one shared unknown focal length: the pieces of ransac_shared_focal_relpose
// (PoseLib/robust/ransac.cc:182-203 with SharedFocalRelativePoseEstimator, robust/estimators/relative_pose.{h:148-175, cc:154-203},
// refiner robust/optim/relative.h:488-592) that are shared by the kernels (sfocal.hip), the host driver (driver_sfocal.inc) and the
// test-only host build (tests/hostmath):
//   * the model - an ImagePair whose two cameras are the same SIMPLE_PINHOLE {f, 0, 0} - as the 8 doubles of FocalModel (q, t, f),
//   * F = K_inv E K_inv in the two association orders the reference uses,
//   * the Sampson residual of compute_sampson_msac_score / get_inliers (utils.cc:204-239, :401-419 - pl_score.h sampson_sq),
//   * the refiner's residual, Jacobian row and step,
//   * the traits that run pl_focal.h's loop template for this estimator.
// The minimal solver is pl_solver_6ptf.h (the reference's template solver restated).  The refiner's residual / Jacobian and the two forms of
// F follow the operation order of the reference's SharedFocalRelativePoseRefiner and estimator (PoseLib, BSD-3-Clause) on purpose:
// bit parity of the refined model with the reference's sources is the requirement, and it fixes the order of every operation.
#pragma once
#include "pl_focal.h"
#include "pl_refine.h"
#include "pl_score.h"

namespace pl {

constexpr int kSFocalSample = 6;
constexpr int kSFocalMaxModels = 60; // relpose_6pt_focal.cc:1101: 15 solutions x 4 poses

PL_HD Mat3 sfocal_essential(const FocalModel &m) {
    Quat q;
    q.w = m.q[0], q.x = m.q[1], q.y = m.q[2], q.z = m.q[3];
    return essential_from_motion(quat_to_rotmat(q), v3(m.t[0], m.t[1], m.t[2]));
}
// K_inv * (E * K_inv), K_inv = diag(1, 1, f): score_model, refine_model's pre-filter and the final get_inliers
// (relative_pose.cc:165-168, :179-182, ransac.cc:194-198)
PL_HD void sfocal_F_score(const FocalModel &m, double *F /* row-major */) {
    const Mat3 E = sfocal_essential(m);
    for (int i = 0; i < 9; ++i)
        F[i] = E.m[i];
    F[2] = F[2] * m.f, F[5] = F[5] * m.f, F[8] = F[8] * m.f;
    F[6] = m.f * F[6], F[7] = m.f * F[7], F[8] = m.f * F[8];
}
// K_inv * E * K_inv, left to right: the refiner (relative.h:499-500, :519-520)
PL_HD void sfocal_F_refine(const Mat3 &E, double f, double *F) {
    for (int i = 0; i < 9; ++i)
        F[i] = E.m[i];
    F[6] = f * F[6], F[7] = f * F[7], F[8] = f * F[8];
    F[2] = F[2] * f, F[5] = F[5] * f, F[8] = F[8] * f;
}

// ---- the refiner (6 parameters: rotation 3, translation tangent 2, focal 1).  Parameter block: q (0..3), t (4..6), the
// tangent basis (7..12, refreshed with the Jacobian - Refiner<EST_REL>::prepare_params), focal (13).
struct SFocalCtx {
    double F[9];     // K_inv E K_inv
    double D[9 * 6]; // d vec(F) / d params, column-major vec index m: D[m * 6 + c]
};
constexpr int kSFocalFocalSlot = 13;
PL_HD void sfocal_prepare(const double *p, SFocalCtx &c, bool jacobian) {
    Quat q;
    q.w = p[0], q.x = p[1], q.y = p[2], q.z = p[3];
    const Mat3 R = quat_to_rotmat(q);
    const Vec3 t = v3(p[4], p[5], p[6]);
    const Mat3 E = essential_from_motion(R, t);
    const double focal = p[kSFocalFocalSlot];
    sfocal_F_refine(E, focal, c.F);
    if (!jacobian)
        return;
    // relative.h:39-61 (deriv_essential_wrt_pose), then :527-540
    const Vec3 e0 = col(E, 0), e1 = col(E, 1), e2 = col(E, 2);
    const Vec3 zero = v3(0, 0, 0);
    const Vec3 blocks[3][3] = {{zero, -e2, e1}, {e2, zero, -e0}, {-e1, e0, zero}};
    const Vec3 tb0 = v3(p[7], p[8], p[9]), tb1 = v3(p[10], p[11], p[12]);
    for (int cb = 0; cb < 3; ++cb) {
        for (int k = 0; k < 3; ++k) {
            const Vec3 v = blocks[cb][k];
            c.D[(3 * cb + 0) * 6 + k] = v.x;
            c.D[(3 * cb + 1) * 6 + k] = v.y;
            c.D[(3 * cb + 2) * 6 + k] = v.z;
        }
        const Vec3 a = cross(tb0, col(R, cb)), b = cross(tb1, col(R, cb));
        c.D[(3 * cb + 0) * 6 + 3] = a.x, c.D[(3 * cb + 1) * 6 + 3] = a.y, c.D[(3 * cb + 2) * 6 + 3] = a.z;
        c.D[(3 * cb + 0) * 6 + 4] = b.x, c.D[(3 * cb + 1) * 6 + 4] = b.y, c.D[(3 * cb + 2) * 6 + 4] = b.z;
    }
    const double ff = focal * focal;
    const int once[4] = {2, 5, 6, 7};
    for (int m = 0; m < 4; ++m)
        for (int k = 0; k < 5; ++k)
            c.D[once[m] * 6 + k] *= focal;
    for (int k = 0; k < 5; ++k)
        c.D[8 * 6 + k] *= ff;
    const double df[9] = {0.0, 0.0, E(2, 0), 0.0, 0.0, E(2, 1), E(0, 2), E(1, 2), 2 * E(2, 2) * focal};
    for (int m = 0; m < 9; ++m)
        c.D[m * 6 + 5] = df[m];
}
PL_HD double sfocal_residual(const SFocalCtx &c, double a0, double a1, double b0, double b1) {
    return sampson_residual(c.F, a0, a1, b0, b1);
}
PL_HD double sfocal_jacobian(const SFocalCtx &c, double a0, double a1, double b0, double b1, double *J /* 6 */) {
    double dF[9];
    const double r = sampson_residual_grad(c.F, a0, a1, b0, b1, dF);
    for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int m = 0; m < 9; ++m)
            s += dF[m] * c.D[m * 6 + k];
        J[k] = s;
    }
    return r;
}
PL_HD void sfocal_step(const double *p, const double *dp, double *out) { // relative.h:577-585
    RefineCtx unused;
    Refiner<EST_REL>::step(p, unused, dp, out);
    out[kSFocalFocalSlot] = p[kSFocalFocalSlot] + dp[5];
}
// One correspondence's row of the normal equations: [w, w r, J0..J5]; false: weight zero (jacobian_accumulator.h:101-103)
constexpr int kSFocalRow = 8;
PL_HD bool sfocal_row(const SFocalCtx &c, const Loss &loss, double a0, double a1, double b0, double b1, double *row) {
    double J[6];
    const double r = sfocal_jacobian(c, a0, a1, b0, b1, J);
    const double w = 1.0 * loss_weight(loss, r * r);
    if (w == 0)
        return false;
    row[0] = w;
    row[1] = w * r;
    for (int k = 0; k < 6; ++k)
        row[2 + k] = J[k];
    return true;
}
// entry e of [lower triangle row-major (21) | Jtr (6)]: JtJ(i, j) += w (J_i J_j), Jtr(i) += (w r) J_i - one shape for both kinds
// (term = (tri ? w : 1) * (row[a] row[b]); 1.0 x is exact, the product commutes), so that the consumer loop of k_sfocal_lm is free
// of branches and of index arithmetic
constexpr int kSFocalEntries = 27;
struct SFocalEntry {
    int a, b;
    bool tri;
};
PL_HD SFocalEntry sfocal_entry_of(int e) {
    SFocalEntry en;
    if (e < 21) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= e)
            ++i;
        en.a = 2 + i, en.b = 2 + (e - i * (i + 1) / 2), en.tri = true;
    } else {
        en.a = 1, en.b = 2 + (e - 21), en.tri = false;
    }
    return en;
}
PL_HD double sfocal_entry_term(const double *row, const SFocalEntry &en) {
    const double t = row[en.a] * row[en.b];
    return (en.tri ? row[0] : 1.0) * t;
}
PL_HD double sfocal_entry_term(const double *row, int e) { return sfocal_entry_term(row, sfocal_entry_of(e)); }

// ---- the loop: score_model is the plain MSAC score of the Sampson error (relative_pose.cc:164-171) - the back end returns it
// whole (inliers' residuals and the outliers' thresholds added in correspondence order, utils.cc:226-236) ----
struct SharedFocalTraits {
    static constexpr int kSample = kSFocalSample, kMaxModels = kSFocalMaxModels;
    static double finish(double score, uint64_t, uint64_t, const FocalLoopOptions &, double) { return score; }
};

// ---- kernels (sfocal.hip) ----
struct SFocalGenArgs {
    const double *a[4]; // x1, y1, x2, y2
    uint32_t n;
    uint64_t seed, pos_base;
    const uint32_t *positions;
    const uint32_t *samples; // optional: num_iters x kSFocalSample explicit indices (PROSAC)
    uint32_t num_iters;
    FocalModel *models;   // [num_iters * kSFocalMaxModels]
    uint32_t *num_models; // [num_iters]
    FocalModel *host_models;   // optional mirrors in pinned host memory (same layout): only the models found are written
    uint32_t *host_num_models;
    double *stage;             // sfocal_stage_bytes(num_iters): workspace of the three generator kernels (sfocal.hip)
    const double *explicit_in; // optional: num_iters x 36 minimal problems [x1 6 x 3 | x2 6 x 3] instead of samples of the points
};
struct SFocalLMTask;
struct SFocalScoreArgs {
    const double *a[4];
    uint32_t n;
    const FocalModel *models;
    const uint32_t *num_models; // per group of kSFocalMaxModels slots; nullptr: every slot holds a model
    const SFocalLMTask *lm_tasks; // optional: slot s = the model k_sfocal_lm left in task s (a skipped task still holds its seed) instead of models
    uint32_t num_slots;
    double thr2;
    uint32_t *counts; // [num_slots]
    double *scores;   // [num_slots]
};
// one refinement (refine_model, relative_pose.cc:173-203; with prefilter_thr2 == 0 and a mask: the front-end's final bundle)
struct SFocalLMTask {
    const double *a[4];
    uint32_t n, pad;
    double params[kParamDoubles]; // in/out: q, t, -, focal at kSFocalFocalSlot
    LMOptions opt;
    double prefilter_thr2; // > 0: refine on the correspondences with Sampson error below it; skip when <= 6 of them
    const uint8_t *mask;   // prefilter_thr2 == 0: optional subset
    uint8_t *scratch;      // n bytes (prefilter)
    uint32_t iterations, skipped;
    double cost, initial_cost;
};

#if defined(__HIPCC__)
hipError_t launch_sfocal_generate_g(const SFocalGenArgs *args, uint32_t G, uint32_t max_iters, hipStream_t stream);
hipError_t launch_sfocal_score_g(const SFocalScoreArgs *args, uint32_t G, uint32_t max_slots, bool workgroup_per_model, hipStream_t stream);
hipError_t launch_sfocal_mask_g(const FocalMaskArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream); // (FocalMaskArgs.a[4] unused)
hipError_t launch_sfocal_generate(const SFocalGenArgs &g, hipStream_t stream);
hipError_t launch_sfocal_score(const SFocalScoreArgs &a, hipStream_t stream);
size_t sfocal_stage_bytes(uint32_t num_iters);
hipError_t launch_sfocal_solve(const double *in, uint32_t count, FocalModel *models, uint32_t *num_models, double *stage,
                               uint32_t stage_samples, hipStream_t stream);
hipError_t launch_sfocal_mask(const double *const *a, uint32_t n, const FocalModel &m, double thr2, uint8_t *mask, uint8_t *host_mask,
                              hipStream_t stream);
hipError_t launch_sfocal_lm(SFocalLMTask *tasks, uint32_t num_tasks, hipStream_t stream);
#endif

} // namespace pl
