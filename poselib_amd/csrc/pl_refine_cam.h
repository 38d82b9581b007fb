// poselib_amd — absolute-pose bundle adjustment that refines camera intrinsics along with the pose
// (BundleOptions::refine_focal_length / refine_principal_point / refine_extra_params).
//
// Reference semantics followed (citations relative to /root/reference/PoseLib):
//   robust/bundle.cc:93-118                  bundle_adjust: Camera::get_param_refinement_idx -> AbsolutePoseRefiner
//   robust/optim/absolute.h:49-171           Jacobian columns [rotation 3 | translation 3 | selected camera parameters],
//                                            step adds the increments to the selected parameters
//   misc/camera_models.cc:688-698, 739-747, 953-965, 1005-1021   d projection / d parameters (SIMPLE_PINHOLE, PINHOLE, OPENCV)
//   misc/camera_models.cc get_param_refinement_idx: focal indices, principal-point indices, extra indices in that order
//
// PL_HD like pl_refine.h: k_lm_cam (lm_cam.hip) and the host test build (tests/hostmath) run the same functions; the
// oracle's restatement (oracle/src/refine.cc AbsProblem with cam_idx) is pinned bit-exactly against the reference sources.
//
// A correspondence's contribution is one ROW of kCamRow doubles: [w, w r0, w r1, J0[0..13], J1[0..13]] with the columns
// in a fixed layout - 0..5 the pose, 6 + m the camera parameter m of the model (all of them, refined or not).  The normal
// equations of the K = 6 + M refined columns are sums over rows of
//     JtJ(i, j) += w (J0[c_i] J0[c_j] + J1[c_i] J1[c_j]),      Jtr(i) += J0[c_i] (w r0) + J1[c_i] (w r1)
// (jacobian_accumulator.h:82-97), c_i = i for the pose, 6 + idx[i - 6] for the camera: selecting columns by index on
// the consumer's side keeps every array index in the producer a compile-time constant.
#pragma once
#include "pl_refine.h"

namespace pl {

constexpr int kCamMaxParams = 8;                   // OPENCV: fx fy cx cy k1 k2 p1 p2
constexpr int kCamMaxK = 6 + kCamMaxParams;        // 14
constexpr int kCamRow = 3 + 2 * kCamMaxK;          // 31 doubles per correspondence
constexpr int kCamMaxEntries = kCamMaxK * (kCamMaxK + 1) / 2 + kCamMaxK; // 119

enum CamRefineFlags : int { CAM_REFINE_FOCAL = 1, CAM_REFINE_PRINCIPAL = 2, CAM_REFINE_EXTRA = 4 };

// Which parameters of the model the flags select, in the reference's order (focal, principal point, extra).  Returns M.
PL_HD int camera_refinement_idx(int model_id, int flags, int *idx) {
    int m = 0;
    const bool f = flags & CAM_REFINE_FOCAL, pp = flags & CAM_REFINE_PRINCIPAL, ex = flags & CAM_REFINE_EXTRA;
    switch (model_id) {
    case CAM_SIMPLE_PINHOLE:
        if (f)
            idx[m++] = 0;
        if (pp)
            idx[m++] = 1, idx[m++] = 2;
        break;
    case CAM_PINHOLE:
    case CAM_OPENCV:
        if (f)
            idx[m++] = 0, idx[m++] = 1;
        if (pp)
            idx[m++] = 2, idx[m++] = 3;
        if (ex && model_id == CAM_OPENCV)
            idx[m++] = 4, idx[m++] = 5, idx[m++] = 6, idx[m++] = 7;
        break;
    default:
        break;
    }
    return m;
}

// projection, d(xp)/dZ (2x3 row-major) and d(xp)/d(parameters) (Jc[0][m], Jc[1][m] for EVERY parameter m of the model)
PL_HD void camera_project_jac_params(const CameraParams &c, Vec3 Z, double &ox, double &oy, double *J, double (*Jc)[kCamMaxParams]) {
    PL_UNROLL
    for (int m = 0; m < kCamMaxParams; ++m)
        Jc[0][m] = Jc[1][m] = 0.0;
    switch (c.model_id) {
    case CAM_SIMPLE_PINHOLE:
    case CAM_PINHOLE: {
        const bool simple = c.model_id == CAM_SIMPLE_PINHOLE;
        const double fx = c.p[0], fy = simple ? c.p[0] : c.p[1];
        const double cx = simple ? c.p[1] : c.p[2], cy = simple ? c.p[2] : c.p[3];
        const double zi = 1.0 / Z.z;
        const double px = fx * Z.x * zi, py = fy * Z.y * zi;
        J[0] = fx * zi, J[1] = 0.0, J[2] = -px * zi;
        J[3] = 0.0, J[4] = fy * zi, J[5] = -py * zi;
        if (simple) {
            Jc[0][0] = Z.x * zi, Jc[1][0] = Z.y * zi;
            Jc[0][1] = 1.0, Jc[1][2] = 1.0;
        } else {
            Jc[0][0] = Z.x * zi, Jc[1][1] = Z.y * zi;
            Jc[0][2] = 1.0, Jc[1][3] = 1.0;
        }
        ox = px + cx;
        oy = py + cy;
        return;
    }
    case CAM_OPENCV: {
        const double u = Z.x / Z.z, v = Z.y / Z.z;
        double du, dv, Jd[4];
        opencv_distort(c.p[4], c.p[5], c.p[6], c.p[7], u, v, du, dv, Jd);
        const double P[6] = {1.0 / Z.z, 0.0, -u / Z.z, 0.0, 1.0 / Z.z, -v / Z.z};
        PL_UNROLL
        for (int a = 0; a < 2; ++a)
            PL_UNROLL
            for (int b = 0; b < 3; ++b)
                J[3 * a + b] = Jd[2 * a] * P[b] + Jd[2 * a + 1] * P[3 + b];
        PL_UNROLL
        for (int b = 0; b < 3; ++b) {
            J[b] *= c.p[0];
            J[3 + b] *= c.p[1];
        }
        const double u2 = u * u, uv = u * v, v2 = v * v, r2 = u * u + v * v;
        const double j0[4] = {r2 * u, r2 * r2 * u, 2.0 * uv, (r2 + 2.0 * u2)};
        const double j1[4] = {r2 * v, r2 * r2 * v, (r2 + 2.0 * v2), 2.0 * uv};
        Jc[0][0] = du, Jc[1][1] = dv;
        Jc[0][2] = 1.0, Jc[1][3] = 1.0;
        PL_UNROLL
        for (int k = 0; k < 4; ++k) {
            Jc[0][4 + k] = c.p[0] * j0[k];
            Jc[1][4 + k] = c.p[1] * j1[k];
        }
        ox = c.p[0] * du + c.p[2];
        oy = c.p[1] * dv + c.p[3];
        return;
    }
    default: { // identity camera: no parameters
        ox = Z.x / Z.z;
        oy = Z.y / Z.z;
        const double zi = 1.0 / Z.z;
        J[0] = zi, J[1] = 0.0, J[2] = -ox * zi;
        J[3] = 0.0, J[4] = zi, J[5] = -oy * zi;
    }
    }
}

// One correspondence's row.  false: no contribution (behind the camera, absolute.h:100-102, or weight zero,
// jacobian_accumulator.h:85-87) - the row is left untouched.  R: rotation of the current pose, row-major; t: p[4..6].
PL_HD bool abs_cam_row(const double *p, const double *R, const CameraParams &cam, const Loss &loss, double x, double y, double X,
                       double Y, double Z, double *row /* kCamRow */) {
    const Vec3 Zc = v3(R[0] * X + R[1] * Y + R[2] * Z + p[4], R[3] * X + R[4] * Y + R[5] * Z + p[5],
                       R[6] * X + R[7] * Y + R[8] * Z + p[6]);
    if (Zc.z < 0)
        return false;
    double px, py, Jp[6], Jc[2][kCamMaxParams];
    camera_project_jac_params(cam, Zc, px, py, Jp, Jc);
    const double r0 = px - x, r1 = py - y;
    const double w = 1.0 * loss_weight(loss, r0 * r0 + r1 * r1);
    if (w == 0)
        return false;
    row[0] = w;
    row[1] = w * r0;
    row[2] = w * r1;
    PL_UNROLL
    for (int a = 0; a < 2; ++a) {
        double *J = row + 3 + kCamMaxK * a;
        const double d0 = Jp[3 * a] * R[0] + Jp[3 * a + 1] * R[3] + Jp[3 * a + 2] * R[6];
        const double d1 = Jp[3 * a] * R[1] + Jp[3 * a + 1] * R[4] + Jp[3 * a + 2] * R[7];
        const double d2 = Jp[3 * a] * R[2] + Jp[3 * a + 1] * R[5] + Jp[3 * a + 2] * R[8];
        J[0] = -Z * d1 + Y * d2;
        J[1] = Z * d0 - X * d2;
        J[2] = -Y * d0 + X * d1;
        J[3] = d0;
        J[4] = d1;
        J[5] = d2;
        PL_UNROLL
        for (int m = 0; m < kCamMaxParams; ++m)
            J[6 + m] = Jc[a][m];
    }
    return true;
}

// Residual pass: the correspondence's robust cost term; false when it is skipped (behind the camera).
PL_HD bool abs_cam_cost(const double *p, const double *R, const CameraParams &cam, const Loss &loss, double x, double y, double X,
                        double Y, double Z, double &term) {
    const Vec3 Zc = v3(R[0] * X + R[1] * Y + R[2] * Z + p[4], R[3] * X + R[4] * Y + R[5] * Z + p[5],
                       R[6] * X + R[7] * Y + R[8] * Z + p[6]);
    if (Zc.z < 0)
        return false;
    double px, py;
    camera_project(cam, Zc, px, py);
    const double r0 = px - x, r1 = py - y;
    term = 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
    return true;
}

// Entry e of the packed normal equations [lower triangle row-major | Jtr] of K = 6 + M columns, as the four row offsets its
// term multiplies and whether the weight multiplies it:  term = (tri ? w : 1) * (row[a] row[b] + row[c] row[d])
//   triangle entry (i, j):  a, c = J0[c_i], J1[c_i];  b, d = J0[c_j], J1[c_j]      w (J0i J0j + J1i J1j)
//   gradient entry i:       a, c = J0[c_i], J1[c_i];  b, d = w r0, w r1            J0i (w r0) + J1i (w r1)   (1.0 x is exact)
// One shape for both kinds keeps the consumer loop of k_lm_cam free of branches: the LDS reads of several rows travel together.
struct CamEntry {
    int a, b, c, d;
    bool tri;
};
PL_HD CamEntry cam_entry_of(int e, int K, const int *idx) {
    const int T = K * (K + 1) / 2;
    int i, j;
    if (e < T) {
        i = 0;
        while ((i + 1) * (i + 2) / 2 <= e)
            ++i;
        j = e - i * (i + 1) / 2;
    } else {
        i = e - T;
        j = -1;
    }
    const int ci = (i < 6) ? i : 6 + idx[i - 6];
    CamEntry en;
    en.a = 3 + ci, en.c = 3 + kCamMaxK + ci;
    en.tri = j >= 0;
    if (en.tri) {
        const int cj = (j < 6) ? j : 6 + idx[j - 6];
        en.b = 3 + cj, en.d = 3 + kCamMaxK + cj;
    } else {
        en.b = 1, en.d = 2;
    }
    return en;
}
// ... and the entry's term of one row
PL_HD double cam_entry_term(const double *row, const CamEntry &en) {
    const double t = row[en.a] * row[en.b] + row[en.c] * row[en.d];
    return (en.tri ? row[0] : 1.0) * t;
}

// lm_solve / lm_update of pl_refine.h for a run-time K = 7..14 (thread 0 of the kernel; the fully unrolled Cholesky of
// every K keeps the factor in registers)
PL_HD void lm_solve_k(int K, LMControl &c, const double *normal, bool fresh, uint32_t jac_count) {
    switch (K) {
    case 7: lm_solve<7>(c, normal, fresh, jac_count); break;
    case 8: lm_solve<8>(c, normal, fresh, jac_count); break;
    case 9: lm_solve<9>(c, normal, fresh, jac_count); break;
    case 10: lm_solve<10>(c, normal, fresh, jac_count); break;
    case 11: lm_solve<11>(c, normal, fresh, jac_count); break;
    case 12: lm_solve<12>(c, normal, fresh, jac_count); break;
    case 13: lm_solve<13>(c, normal, fresh, jac_count); break;
    default: lm_solve<14>(c, normal, fresh, jac_count); break;
    }
}
PL_HD bool lm_update_k(int K, LMControl &c, const double *normal, double racc, uint32_t res_count) {
    switch (K) {
    case 7: return lm_update<7>(c, normal, racc, res_count);
    case 8: return lm_update<8>(c, normal, racc, res_count);
    case 9: return lm_update<9>(c, normal, racc, res_count);
    case 10: return lm_update<10>(c, normal, racc, res_count);
    case 11: return lm_update<11>(c, normal, racc, res_count);
    case 12: return lm_update<12>(c, normal, racc, res_count);
    case 13: return lm_update<13>(c, normal, racc, res_count);
    default: return lm_update<14>(c, normal, racc, res_count);
    }
}

// absolute.h:157-170: pose step as Refiner<EST_ABS>::step, then the selected parameters += their increments
PL_HD void abs_cam_step(const double *p, const CameraParams &cam, const double *dp, const int *idx, int M, double *out,
                        CameraParams &cam_out) {
    RefineCtx unused;
    Refiner<EST_ABS>::step(p, unused, dp, out);
    cam_out = cam;
    for (int m = 0; m < M; ++m)
        cam_out.p[idx[m]] += dp[6 + m];
}

} // namespace pl
