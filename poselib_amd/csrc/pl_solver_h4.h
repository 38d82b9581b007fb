// poselib_amd — 4-point homography for one lane.
//
// Follows PoseLib/solvers/homography_4pt.cc:36-128: orientation pre-check on the four bearings
// (:38-55), then the closed-form affine-core-affine construction (:57-118; Cai et al., PAMI'25 —
// no DLT / SVD), Frobenius normalisation and the |det| >= 1e-8 post-check (:121-125).
#pragma once
#include "pl_math.h"

namespace pl {

PL_HD int homography_4pt(const Vec3 *x1, const Vec3 *x2, Mat3 &H, bool check_orientation) {
    if (check_orientation) {
        Vec3 p = cross(x1[0], x1[1]), q = cross(x2[0], x2[1]);
        if (dot(p, x1[2]) * dot(q, x2[2]) < 0)
            return 0;
        if (dot(p, x1[3]) * dot(q, x2[3]) < 0)
            return 0;
        p = cross(x1[2], x1[3]);
        q = cross(x2[2], x2[3]);
        if (dot(p, x1[0]) * dot(q, x2[0]) < 0)
            return 0;
        if (dot(p, x1[1]) * dot(q, x2[1]) < 0)
            return 0;
    }
    double ax[4], ay[4], bx[4], by[4];
    for (int i = 0; i < 4; ++i) {
        ax[i] = x1[i].x / x1[i].z;
        ay[i] = x1[i].y / x1[i].z;
        bx[i] = x2[i].x / x2[i].z;
        by[i] = x2[i].y / x2[i].z;
    }
    // affine frames anchored at point 0 in each plane
    const double n1x = ax[1] - ax[0], p1x = ax[2] - ax[0], q1x = ax[3] - ax[0];
    const double n1y = ay[1] - ay[0], p1y = ay[2] - ay[0], q1y = ay[3] - ay[0];
    const double fA1 = n1x * p1y - n1y * p1x;
    const double Q3x = p1y * q1x - p1x * q1y;
    const double Q3y = n1x * q1y - n1y * q1x;
    const double n2x = bx[1] - bx[0], p2x = bx[2] - bx[0], q2x = bx[3] - bx[0];
    const double n2y = by[1] - by[0], p2y = by[2] - by[0], q2y = by[3] - by[0];
    const double fA2 = n2x * p2y - n2y * p2x;
    const double Q4x = p2y * q2x - p2x * q2y;
    const double Q4y = n2x * q2y - n2y * q2x;
    // diagonal-plus-last-row core
    const double tt1 = fA1 - Q3x - Q3y;
    const double C11 = Q3y * Q4x * tt1;
    const double C22 = Q3x * Q4y * tt1;
    const double C33 = Q3x * Q3y * (fA2 - Q4x - Q4y);
    const double C31 = C11 - C33;
    const double C32 = C22 - C33;
    const double tt3 = bx[0] * C33;
    const double tt4 = by[0] * C33;
    const double H11 = bx[1] * C11 - tt3;
    const double H12 = bx[2] * C22 - tt3;
    const double H21 = by[1] * C11 - tt4;
    const double H22 = by[2] * C22 - tt4;
    double h[9];
    h[0] = H11 * p1y - H12 * n1y;
    h[1] = H12 * n1x - H11 * p1x;
    h[3] = H21 * p1y - H22 * n1y;
    h[4] = H22 * n1x - H21 * p1x;
    h[6] = C31 * p1y - C32 * n1y;
    h[7] = C32 * n1x - C31 * p1x;
    h[2] = tt3 * fA1 - h[0] * ax[0] - h[1] * ay[0];
    h[5] = tt4 * fA1 - h[3] * ax[0] - h[4] * ay[0];
    h[8] = C33 * fA1 - h[6] * ax[0] - h[7] * ay[0];
    // Frobenius norm, summed in the column-major order of the 3x3 (h is row-major)
    double s = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            s += h[3 * i + j] * h[3 * i + j];
    const double nrm = sqrt(s);
    for (int i = 0; i < 9; ++i)
        H.m[i] = h[i] / nrm;
    if (fabs(det3(H)) < 1e-8)
        return 0;
    return 1;
}

} // namespace pl
