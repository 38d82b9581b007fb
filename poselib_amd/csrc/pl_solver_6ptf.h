// poselib_amd - relative pose of two views that share ONE unknown focal length, from six correspondences.
//
// The reference's solver restated (PoseLib/solvers/relpose_6pt_focal.cc; interface solvers/relpose_6pt_focal.h:12-13: unit bearings
// with the principal point at the origin) so that it returns the reference's solutions in the reference's order, to the last bit
// of oracle/_ref's build (up to the cubes, below):
//   1. null space of the six epipolar constraints from a full-pivoting Householder QR, F = N0 + x N1 + y N2 (:1086-1094)
//                                                                                                       six_nullspace
//   2. with w = 1 / f^2, Q = diag(1, 1, w): 2 F Q F^T Q F - tr(F Q F^T Q) F = 0 and det F = 0 - 280 coefficients, the 31 x 46
//      template [C0 | C1], the last eight rows of C0^-1 C1 (:54-1043)            pl_action_template.h + pl_focal_templates.h
//   3. the 15 x 15 action matrix of the multiplication by y (:1045-1053), its characteristic polynomial by Danilevsky's method
//      (misc/sturm.h:287-326) and the real roots by Sturm bisection of degree 15, tolerance 1e-12 (:1069-1076)
//   4. per root x and w from a 7 x 7 system (:11-52), w < 1e-8 dropped, focal = sqrt(1 / w), E = K F K, motion_from_essential on
//      the bearings at that focal length (:1104-1141)                                     six_root_xw, six_solution_poses
// The cubes d^3 of step 2: the reference's formulas call std::pow(d, 3), whose glibc result is the correctly rounded cube except
// for 8 of 10^4 arguments; exact_cube (pl_action_template.h) is the correctly rounded one, and the oracle has a switch for it.
// On the device: step 1 one lane per sample (k_sfocal_setup), steps 2 - 3 one wavefront per sample with the matrices in LDS, step 4
// one lane per root, then one lane per (root, pose candidate) (sfocal.hip).  relpose_6pt_shared_focal() below is the serial
// statement: tests/hostmath holds it against the oracle bit for bit.
#pragma once
#include "pl_action_template.h"
#include "pl_solver_p35pf.h" // complement_basis_indexed
#include "pl_solver_rel.h"   // motion_from_essential_emit
#include "pl_sturm_n.h"

namespace pl {

constexpr int kSixMaxModels = 60; // 15 solutions x 4 poses (relpose_6pt_focal.cc:1101)

// Step 1: nb = the last three columns of Q (9 x 3 column-major)
PL_HD void six_nullspace(const Vec3 *x1, const Vec3 *x2, double *nb /* 27 */) {
    double A[54];
    for (int i = 0; i < 6; ++i) {
        const double a[3] = {x1[i].x, x1[i].y, x1[i].z};
        for (int j = 0; j < 3; ++j) {
            A[i * 9 + 3 * j + 0] = a[j] * x2[i].x;
            A[i * 9 + 3 * j + 1] = a[j] * x2[i].y;
            A[i * 9 + 3 * j + 2] = a[j] * x2[i].z;
        }
    }
    complement_basis_indexed<9, 6>(A, nb);
}

// the action matrix takes its rows from RR = [-(rows 23 .. 30 of C0^-1 C1); I] (relpose_6pt_focal.cc:1045-1053)
static constexpr int8_t kSixActionRow[15] = {15, 11, 0, 1, 2, 12, 3, 16, 4, 5, 17, 6, 18, 19, 7};
// entry (i, j) of the action matrix; X(r, j): row 23 + r of column j of C0^-1 C1
template <class Tail> PL_HD double six_action_entry(const Tail &X, int i, int j) {
    const int r = kSixActionRow[i];
    return r < 8 ? -X(r, j) : (r - 8 == j ? 1.0 : 0.0);
}

// Step 4, first half: x and w that belong to the root y = z0 of the action matrix AM (15 x 15 row-major)
static constexpr int8_t kSixReduced[8] = {2, 3, 4, 6, 8, 9, 11, 14};
template <class Arr> PL_HD void six_root_xw(const Arr &AM, double z0, double &x, double &w) {
    const double z1 = z0 * z0, z2 = z1 * z0;
    double AA[64]; // column-major 8 x 8
    for (int r = 0; r < 8; ++r) {
        const int row = kSixReduced[r] * 15;
        AA[0 * 8 + r] = AM[row + 2];
        AA[1 * 8 + r] = AM[row + 6];
        AA[2 * 8 + r] = z0 * AM[row + 4] + AM[row + 5];
        AA[3 * 8 + r] = AM[row + 1] + z0 * AM[row + 3];
        AA[4 * 8 + r] = AM[row + 14];
        AA[5 * 8 + r] = z0 * AM[row + 11] + AM[row + 13];
        AA[6 * 8 + r] = z1 * AM[row + 9] + z0 * AM[row + 10] + AM[row + 12];
        AA[7 * 8 + r] = AM[row + 0] + z0 * AM[row + 7] + z1 * AM[row + 8];
    }
    AA[0 * 8 + 0] = AA[0 * 8 + 0] - z0;
    AA[1 * 8 + 3] = AA[1 * 8 + 3] - z0;
    AA[2 * 8 + 2] = AA[2 * 8 + 2] - z1;
    AA[3 * 8 + 1] = AA[3 * 8 + 1] - z1;
    AA[4 * 8 + 7] = AA[4 * 8 + 7] - z0;
    AA[5 * 8 + 6] = AA[5 * 8 + 6] - z1;
    AA[6 * 8 + 5] = AA[6 * 8 + 5] - z2;
    AA[7 * 8 + 4] = AA[7 * 8 + 4] - z2;
    double B[49], rhs[7], s[7];
    for (int c = 0; c < 7; ++c)
        for (int r = 0; r < 7; ++r)
            B[c * 7 + r] = AA[c * 8 + r];
    for (int r = 0; r < 7; ++r)
        rhs[r] = -AA[7 * 8 + r];
    householder_qr_solve<7>(B, rhs, s);
    x = s[3];
    w = s[6];
}

// Step 4, second half: the essential matrix and the bearings at the focal length of one solution (x, y, w)
PL_HD void six_solution_essential(const Vec3 *x1, const Vec3 *x2, const double *nb, double sx, double sy, double sw, Mat3 &E, Vec3 *u1,
                                  Vec3 *u2, double &focal_out) {
    const double focal = sqrt(1.0 / sw);
    double Fv[9], nrm = 0;
    for (int e = 0; e < 9; ++e) {
        Fv[e] = nb[e] + sx * nb[9 + e] + sy * nb[18 + e];
        nrm += Fv[e] * Fv[e];
    }
    nrm = sqrt(nrm);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double ki = i < 2 ? focal : 1.0, kj = j < 2 ? focal : 1.0;
            E(i, j) = ki * ((Fv[3 * j + i] / nrm) * kj);
        }
    for (int i = 0; i < 6; ++i) {
        u1[i] = normalized(v3(x1[i].x / focal, x1[i].y / focal, x1[i].z));
        u2[i] = normalized(v3(x2[i].x / focal, x2[i].y / focal, x2[i].z));
    }
    focal_out = focal;
}
// the poses of one solution: emit(q, t, focal) for every one of them (<= 4), in the reference's order
template <class Emit>
PL_HD void six_solution_poses(const Vec3 *x1, const Vec3 *x2, const double *nb, double sx, double sy, double sw, Emit &&emit) {
    Mat3 E;
    Vec3 u1[6], u2[6];
    double focal;
    six_solution_essential(x1, x2, nb, sx, sy, sw, E, u1, u2, focal);
    motion_from_essential_emit<6>(E, u1, u2, [&](Quat q, Vec3 t) { emit(q, t, focal); });
}

// The whole solver, serially (host; tests/hostmath).  x1, x2: six unit bearings.  emit(q, t, focal) is called for every model, in
// the reference's order.  Returns their number.
template <class Emit> PL_HD int relpose_6pt_shared_focal(const Vec3 *x1, const Vec3 *x2, Emit &&emit) {
    double nb[27];
    six_nullspace(x1, x2, nb);
    constexpr int S = 46;
    double coef[kSixCoeffs], C[31 * S];
    for (int k = 0; k < kSixCoeffs; ++k)
        coef[k] = template_coefficient<true>(nb, kSixTermStart, kSixTermPacked, k);
    for (int e = 0; e < 31 * S; ++e)
        C[e] = 0.0;
    for (int c = 0; c < kSixCols; ++c)
        for (int e = kSixColStart[c]; e < kSixColStart[c + 1]; ++e)
            C[kSixEntryRow[e] * S + c] = coef[kSixEntryCoeff[e]];
    lu_solve_tail<31, 46, S, 8>(C);
    double AM[225], AMp[225], poly[16], ev[15];
    auto tail = [&](int r, int j) { return C[(23 + r) * S + 31 + j]; };
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j)
            AMp[i * 15 + j] = AM[i * 15 + j] = six_action_entry(tail, i, j);
    danilevsky_charpoly<15>(AMp, poly);
    const int nroots = sturm_n_roots<15>(poly, ev, 1e-12);
    int n = 0;
    for (int s = 0; s < nroots; ++s) {
        double x, w;
        six_root_xw(AM, ev[s], x, w);
        if (w < 1e-8) // relpose_6pt_focal.cc:1105
            continue;
        six_solution_poses(x1, x2, nb, x, ev[s], w, [&](Quat q, Vec3 t, double focal) {
            emit(q, t, focal);
            ++n;
        });
    }
    return n;
}

} // namespace pl
