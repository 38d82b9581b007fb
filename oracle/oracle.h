/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement of the PoseLib LO-RANSAC hot path (see oracle/README.md).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (poselib_amd/) never links, imports or calls it.
 *
 * Conventions (same as the reference's std::vector<Eigen::Vector2d/3d>::data()):
 *   points are contiguous AoS doubles (N x 2 or N x 3); poses are {q[4] (w,x,y,z), t[3]};
 *   3x3 matrices are COLUMN-major (Eigen::Matrix3d layout).
 */
#ifndef ORACLE_H_
#define ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t max_iterations, min_iterations;
    double dyn_num_trials_mult, success_prob;
    uint64_t seed;
    int32_t progressive_sampling;
    int32_t score_initial_model;
    uint64_t max_prosac_iterations;
} orc_ransac_opt;

typedef struct {
    uint64_t max_iterations;
    int32_t loss_type; /* 0 TRIVIAL 1 TRUNCATED 2 HUBER 3 CAUCHY 4 TRUNCATED_CAUCHY 5 TRUNCATED_LE_ZACH */
    int32_t lambda_update; /* 0 NIELSEN 1 FIXED_FACTOR */
    int32_t damping;       /* 0 LEVENBERG 1 MARQUARDT */
    int32_t refine_flags;  /* bit 0 refine_focal_length, bit 1 refine_principal_point, bit 2 refine_extra_params (types.h:92-94) */
    double loss_scale, gradient_tol, step_tol, relative_cost_tol, initial_lambda, min_lambda, max_lambda, lambda_factor;
} orc_bundle_opt;

typedef struct {
    orc_ransac_opt ransac;
    orc_bundle_opt bundle;
    double max_error;
    int32_t real_focal_check; /* fundamental only */
    int32_t estimate_focal_length; /* absolute pose: AbsolutePoseOptions::estimate_focal_length (robust.cc:47-54) */
    double min_fov;                /* absolute pose: AbsolutePoseOptions::min_fov, degrees (types.h:126; 5.0) */
} orc_robust_opt;

typedef struct {
    uint64_t refinements, iterations, num_inliers;
    double inlier_ratio, model_score;
    uint64_t hypotheses; /* minimal models scored in the main loop (metric numerator) */
    double seconds;      /* wall time of the ransac_* call */
} orc_stats;

typedef struct {
    int32_t model_id; /* -1 NULL, 0 SIMPLE_PINHOLE, 1 PINHOLE, 4 OPENCV */
    int32_t width, height, num_params;
    double params[12];
} orc_camera;

typedef struct {
    uint64_t iterations;
    double initial_cost, cost, lambda, nu;
    uint64_t invalid_steps;
    double step_norm, grad_norm;
} orc_bundle_stats;

/* ---- sampler / loop control ---- */
void orc_sampler_draw(uint64_t seed, uint64_t N, uint64_t K, uint64_t n_samples, int32_t prosac,
                      uint64_t max_prosac_iterations, uint64_t *out_idx /* n_samples*K */, uint64_t *state_after);
int32_t orc_random_int(uint64_t *state);
double orc_all_inlier_probability(uint64_t inliers, uint64_t N, uint64_t K);
uint64_t orc_dynamic_max_iter(uint64_t inliers, uint64_t N, uint64_t K, double log_fail, double mult, uint64_t min_it,
                              uint64_t max_it);
/* tests/ransac_test.cc MockEstimator: one dummy model per iteration, fixed inlier count, score 0 */
void orc_mock_ransac(uint64_t num_data, uint64_t sample_sz, uint64_t inlier_count, const orc_ransac_opt *opt,
                     orc_stats *out);

/* ---- solvers ---- */
int orc_solve_cubic_single_real(double c2, double c1, double c0, double *root);
int orc_solve_cubic_real(double c2, double c1, double c0, double *roots);
int orc_sturm_roots(const double *coeffs, int degree, double *roots);
int orc_sturm_roots_tol(const double *coeffs, int degree, double tol, double *roots);
int orc_p3p(const double *x /*3x3*/, const double *X /*3x3*/, double *poses /*4x7*/);
int orc_p35pf(const double *x /*4x2, principal point at the origin*/, const double *X /*4x3*/, double *poses /*10x7*/,
              double *focals /*10*/);
int orc_relpose_6pt_shared_focal(const double *x1 /*6x3 unit bearings*/, const double *x2, double *poses /*60x7*/,
                                 double *focals /*60*/);
/* the cubes in the six-point solver's coefficients: 1 = correctly rounded (default; what the device computes), 0 = std::pow(d, 3)
   as the reference calls it (bit parity with oracle/_ref on the same host); process-wide, test use only */
void orc_set_exact_cubes(int on);
int orc_essential_5pt(const double *x1, const double *x2, double *E /*10x9 col-major each*/);
int orc_relpose_5pt(const double *x1, const double *x2, double *poses /*40x7*/);
int orc_relpose_7pt(const double *x1, const double *x2, double *F /*3x9*/);
int orc_homography_4pt(const double *x1, const double *x2, double *H /*9*/, int check_cheirality);
void orc_nullspace(const double *A, int rows, int cols, double *basis);

/* ---- scoring ---- */
double orc_score_reproj(const double *pose7, const double *x, const double *X, size_t n, double sq_thr, uint64_t *cnt);
double orc_score_sampson_pose(const double *pose7, const double *x1, const double *x2, size_t n, double sq_thr,
                              uint64_t *cnt);
double orc_score_sampson_F(const double *F9, const double *x1, const double *x2, size_t n, double sq_thr, uint64_t *cnt);
double orc_score_homography(const double *H9, const double *x1, const double *x2, size_t n, double sq_thr,
                            uint64_t *cnt);
void orc_inliers_reproj(const double *pose7, const double *x, const double *X, size_t n, double sq_thr, uint8_t *mask);
void orc_inliers_sampson_pose(const double *pose7, const double *x1, const double *x2, size_t n, double sq_thr,
                              uint8_t *mask);
void orc_inliers_sampson_F(const double *F9, const double *x1, const double *x2, size_t n, double sq_thr, uint8_t *mask);
void orc_inliers_homography(const double *H9, const double *x1, const double *x2, size_t n, double sq_thr,
                            uint8_t *mask);
double orc_normalize_points(double *x1, double *x2, size_t n, double *T1, double *T2, int normalize_scale,
                            int normalize_centroid, int shared_scale);
void orc_unproject(const orc_camera *cam, const double *xp, size_t n, double *out /* n x 2 */);

/* ---- refinement ---- */
void orc_bundle_adjust(const double *x, const double *X, size_t n, const orc_camera *cam, double *pose7,
                       const orc_bundle_opt *opt, orc_bundle_stats *st);
/* the same with the camera in / out: the parameters opt->refine_flags names are refined along with the pose */
void orc_bundle_adjust_camera(const double *x, const double *X, size_t n, orc_camera *cam, double *pose7,
                              const orc_bundle_opt *opt, orc_bundle_stats *st);
void orc_refine_relpose(const double *x1, const double *x2, size_t n, double *pose7, const orc_bundle_opt *opt,
                        orc_bundle_stats *st);
void orc_refine_homography(const double *x1, const double *x2, size_t n, double *H9, const orc_bundle_opt *opt,
                           orc_bundle_stats *st);
void orc_svd3(const double *A9_rowmajor, double *U9, double *s3, double *V9);
void orc_refine_fundamental(const double *x1, const double *x2, size_t n, double *F9, const orc_bundle_opt *opt,
                            orc_bundle_stats *st);

/* ---- RANSAC entry points (robust/ransac.h) and front-ends (robust.h) ---- */
void orc_ransac_pnp(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7,
                    uint8_t *inliers, orc_stats *st);
void orc_ransac_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                        uint8_t *inliers, orc_stats *st);
void orc_ransac_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                            uint8_t *inliers, orc_stats *st);
void orc_ransac_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                           uint8_t *inliers, orc_stats *st);

/* robust/ransac.h ransac_pnpf: pose + focal length (SIMPLE_PINHOLE, principal point at the origin) */
/* shared unknown focal length, two views (robust.h:84-90, ransac.h:71-73, bundle.h:108-111): pose7 / focal in-out */
void orc_refine_shared_focal_relpose(const double *x1, const double *x2, size_t n, double *pose7, double *focal,
                                     const orc_bundle_opt *opt, orc_bundle_stats *st);
void orc_ransac_shared_focal_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                                     double *focal, uint8_t *inliers, orc_stats *st);
void orc_estimate_shared_focal_relative_pose(const double *x1, const double *x2, size_t n, const double *pp2,
                                             const orc_robust_opt *opt, double *pose7, double *focal, uint8_t *inliers,
                                             orc_stats *st);
void orc_ransac_pnpf(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7, double *focal,
                     uint8_t *inliers, orc_stats *st);
void orc_estimate_absolute_pose(const double *p2d, const double *p3d, size_t n, const orc_robust_opt *opt,
                                orc_camera *cam, double *pose7, uint8_t *inliers, orc_stats *st);
void orc_estimate_relative_pose(const double *x1, const double *x2, size_t n, const orc_camera *cam1,
                                const orc_camera *cam2, const orc_robust_opt *opt, double *pose7, uint8_t *inliers,
                                orc_stats *st);
void orc_estimate_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                              uint8_t *inliers, orc_stats *st);
void orc_estimate_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                             uint8_t *inliers, orc_stats *st);

#ifdef __cplusplus
}
#endif
#endif
