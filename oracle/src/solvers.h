// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Declarations of the minimal solvers restated from the reference (see the .cc files for
// per-function file:line citations).
#pragma once
#include "vecmath.h"

namespace orc {

// univariate
int quadratic_real_roots(double a, double b, double c, double r[2]);
bool cubic_one_real_root(double c2, double c1, double c0, double &root);
int cubic_real_roots(double c2, double c1, double c0, double r[3]);

// Real roots of a degree-10 polynomial c[0] + c[1] z + ... + c[10] z^10 by Sturm bracketing.
int sturm_real_roots_deg10(const double c[11], double roots[10], double tol = 1e-10);
// generic-degree variant (N <= 16) used by tests
int sturm_real_roots(const double *c, int N, double *roots, double tol = 1e-10);

// absolute pose: unit bearings x, 3-D points X  ->  <= 4 poses
int p3p(const V3 x[3], const V3 X[3], Pose out[4]);
// P3.5Pf (solvers_focal.cc restates solvers/p35pf.cc; interface solvers/p35pf.h:39-54): image points relative to the principal
// point, at most 10 (pose, focal) solutions in the reference's order
int p35pf(const V2 x[4], const V3 X[4], Pose out[10], double focals[10]);

// relative pose (unit bearings)
// shared unknown focal length from six correspondences (solvers_focal.cc; interface of solvers/relpose_6pt_focal.h:12-13)
int relpose_6pt_shared_focal(const V3 x1[6], const V3 x2[6], Pose out[60], double focals[60]);
// the cubes of the six-point coefficients: correctly rounded like the device (default) or std::pow(d, 3) like the reference
void set_exact_cubes(bool on);
int essential_5pt(const V3 x1[5], const V3 x2[5], M3 E[10]);
int relpose_5pt(const V3 x1[5], const V3 x2[5], Pose out[40]);
int relpose_7pt(const V3 x1[7], const V3 x2[7], M3 F[3]);
int homography_4pt(const V3 x1[4], const V3 x2[4], M3 *H, bool check_cheirality = true);

// essential-matrix helpers
M3 essential_from_motion(const Pose &p);
bool check_cheirality(const Pose &p, const V3 &x1, const V3 &x2, double min_depth = 0.0);
int motion_from_essential(const M3 &E, const V3 *x1, const V3 *x2, int npts, Pose *out);

// Orthonormal basis of the orthogonal complement of span(columns of A) where A is rows x cols
// (column-major, rows >= cols), following the algorithm of Eigen's
// fullPivHouseholderQr().matrixQ().rightCols(rows - cols).  `basis` is rows x (rows-cols),
// column-major.
void householder_complement(const double *A, int rows, int cols, double *basis);

} // namespace orc
