// poselib_amd — per-correspondence residuals of the four MSAC scores.
//
// Each function restates the scalar expression of the reference with the SAME association
// order (PoseLib/robust/utils.cc:36-65 reprojection, :158-201 Sampson + cheirality,
// :204-239 Sampson, :300-329 homography transfer), so for a bit-identical model the squared
// residual — and therefore the inlier decision — is bit-identical to the CPU path.
// The model arrives as a 16-double record (pl_math.h); its fields are wave-uniform and live in
// SGPRs inside the scoring kernel.
#pragma once
#include "pl_math.h"

namespace pl {

enum Estimator : int { EST_ABS = 0, EST_REL = 1, EST_FUND = 2, EST_HOM = 3 };

// ---- absolute pose: correspondence = (x, y, X, Y, Z) ---------------------------------------
// Returns the inlier decision; r2 is valid when the point is in front of the camera.
PL_HD bool reproj_inlier(const double *M, double x, double y, double X, double Y, double Z, double thr2, double &r2) {
    const double *R = M + kMatOff;
    const double z0 = R[0] * X + R[1] * Y + R[2] * Z + M[4];
    const double z1 = R[3] * X + R[4] * Y + R[5] * Z + M[5];
    const double z2 = R[6] * X + R[7] * Y + R[8] * Z + M[6];
    const double inv = 1.0 / z2;
    const double e0 = z0 * inv - x;
    const double e1 = z1 * inv - y;
    r2 = e0 * e0 + e1 * e1;
    // utils.cc:51-52 skips z2 <= 0; a NaN z2 is not skipped there but then r2 is NaN and the
    // comparison below is false as well.
    return (z2 > 0.0) & (r2 < thr2);
}

// ---- two-view: correspondence = (x1, y1, x2, y2) --------------------------------------------
PL_HD double sampson_sq(const double *E /*row-major 3x3*/, double a0, double a1, double b0, double b1) {
    const double Ea0 = E[0] * a0 + E[1] * a1 + E[2];
    const double Ea1 = E[3] * a0 + E[4] * a1 + E[5];
    const double Ea2 = E[6] * a0 + E[7] * a1 + E[8];
    const double Eb0 = E[0] * b0 + E[3] * b1 + E[6];
    const double Eb1 = E[1] * b0 + E[4] * b1 + E[7];
    const double C = b0 * Ea0 + b1 * Ea1 + Ea2;
    const double Cx = Ea0 * Ea0 + Ea1 * Ea1;
    const double Cy = Eb0 * Eb0 + Eb1 * Eb1;
    return C * C / (Cx + Cy);
}

// relative pose: Sampson below threshold AND positive depth in both views (min depth 0.01)
// u1 = bearing(a0, a1), u2 = bearing(b0, b1): they depend on the correspondence only, so a scorer that sees the same
// correspondence for many models passes them in (same function, same bits)
PL_HD bool sampson_pose_inlier_b(const double *M, double a0, double a1, double b0, double b1, Vec3 u1, Vec3 u2,
                                 double thr2, double &r2) {
    r2 = sampson_sq(M + kMatOff, a0, a1, b0, b1);
    if (!(r2 < thr2))
        return false;
    Quat q;
    q.w = M[0], q.x = M[1], q.y = M[2], q.z = M[3];
    return check_cheirality(q, v3(M[4], M[5], M[6]), u1, u2, 0.01);
}
PL_HD bool sampson_pose_inlier(const double *M, double a0, double a1, double b0, double b1, double thr2, double &r2) {
    r2 = sampson_sq(M + kMatOff, a0, a1, b0, b1);
    if (!(r2 < thr2))
        return false;
    Quat q;
    q.w = M[0], q.x = M[1], q.y = M[2], q.z = M[3];
    return check_cheirality(q, v3(M[4], M[5], M[6]), bearing(a0, a1), bearing(b0, b1), 0.01);
}

PL_HD bool sampson_inlier(const double *M, double a0, double a1, double b0, double b1, double thr2, double &r2) {
    r2 = sampson_sq(M + kMatOff, a0, a1, b0, b1);
    return r2 < thr2;
}

PL_HD bool homography_inlier(const double *M, double a0, double a1, double b0, double b1, double thr2, double &r2) {
    const double *H = M + kMatOff;
    const double h0 = H[0] * a0 + H[1] * a1 + H[2];
    const double h1 = H[3] * a0 + H[4] * a1 + H[5];
    const double inv = 1.0 / (H[6] * a0 + H[7] * a1 + H[8]);
    const double e0 = h0 * inv - b0;
    const double e1 = h1 * inv - b1;
    r2 = e0 * e0 + e1 * e1;
    return r2 < thr2;
}

// ---- final inlier masks (different arithmetic form for absolute pose: utils.cc:374-384) -----
PL_HD bool reproj_mask(const double *M, double x, double y, double X, double Y, double Z, double thr2) {
    const double *R = M + kMatOff;
    const double z0 = R[0] * X + R[1] * Y + R[2] * Z + M[4];
    const double z1 = R[3] * X + R[4] * Y + R[5] * Z + M[5];
    const double z2 = R[6] * X + R[7] * Y + R[8] * Z + M[6];
    const double e0 = z0 / z2 - x;
    const double e1 = z1 / z2 - y;
    const double r2 = e0 * e0 + e1 * e1;
    return (r2 < thr2) & (z2 > 0.0);
}

} // namespace pl
