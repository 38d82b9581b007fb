/* poselib_amd — C-ABI of the MI355X-native LO-RANSAC hot path.
 *
 * Drop-in boundary for the robust-estimation path of PoseLib 3.0.0 (citations relative to the
 * reference tree, PoseLib/...).  Every entry point is `extern "C"`, takes plain pointers and sizes
 * and is what a binding of the reference's API for this path would call:
 *
 *   pl_estimate_absolute_pose   <- robust.h:45-46     estimate_absolute_pose(points2D, points3D, opt, Image*, inliers*)
 *   pl_estimate_relative_pose   <- robust.h:68-70     estimate_relative_pose(x1, x2, camera1, camera2, opt, pose*, inliers*)
 *   pl_estimate_fundamental     <- robust.h:112-113   estimate_fundamental(x1, x2, opt, F*, inliers*)
 *   pl_estimate_homography      <- robust.h:133-134   estimate_homography(x1, x2, opt, H*, inliers*)
 *   pl_ransac_pnp / _relpose / _fundamental / _homography
 *                               <- robust/ransac.h:39-40, 60-61, 85-87, 99-101
 *   pl_p3p / pl_relpose_5pt / pl_essential_matrix_5pt / pl_relpose_7pt / pl_homography_4pt
 *                               <- solvers/p3p.h:42, relpose_5pt.h:40-43, relpose_7pt.h:39-40, homography_4pt.h:38-39
 *   pl_problem_* / pl_ransac_run <- the same ransac_* loop on correspondences that are already resident in
 *                                  HBM (what bench.py times), and its batched / multi-stream form.
 *
 * Conventions (identical to the reference's containers):
 *   - 2-D / 3-D points: contiguous AoS doubles, N x 2 / N x 3  (== std::vector<Eigen::Vector2d/3d>::data()).
 *   - pose: q[4] (w, x, y, z) then t[3];  3x3 matrices: COLUMN-major (== Eigen::Matrix3d::data()).
 *   - `inliers`: caller-allocated N bytes, 0/1 (the reference resizes a std::vector<char>).
 *   - in/out models are the initial model when ransac.score_initial_model != 0, otherwise they are
 *     overwritten (ransac.cc:47-50, 144-147, 252-254, 304-306).
 *   - Algorithmic "failure" is reported through the stats exactly like the reference (e.g. too few points
 *     => iterations == 0).  The int return value is 0 on success and a negative PL_ERR_* code for
 *     runtime failures only.  There is NO CPU fallback: without a usable HIP device every compute entry
 *     point returns PL_ERR_NO_DEVICE.
 *   - Thread-safety: re-entrant; each host thread gets its own HIP stream and scratch arena.
 */
#ifndef POSELIB_AMD_H_
#define POSELIB_AMD_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PL_OK 0
#define PL_ERR_NO_DEVICE (-1)
#define PL_ERR_HIP (-2)
#define PL_ERR_INVALID (-3)
#define PL_ERR_UNSUPPORTED (-4)
#define PL_ERR_COMM (-5) /* the caller's all-gather callback failed (pl_ransac_run_sharded) */

/* types.h:39-50 */
typedef struct {
    uint64_t max_iterations;   /* 100000 */
    uint64_t min_iterations;   /* 1000 */
    double dyn_num_trials_mult; /* 3.0 */
    double success_prob;        /* 0.9999 */
    uint64_t seed;              /* 0 */
    int32_t progressive_sampling; /* PROSAC (sampling.cc:85-136): samples drawn on the host, models on the device */
    int32_t score_initial_model;
    uint64_t max_prosac_iterations;
} pl_ransac_options;

/* types.h:60-95 */
typedef struct {
    uint64_t max_iterations; /* 100 */
    int32_t loss_type;       /* 0 TRIVIAL, 1 TRUNCATED, 2 HUBER, 3 CAUCHY (default), 4 TRUNCATED_CAUCHY, 5 TRUNCATED_LE_ZACH */
    int32_t lambda_update;   /* 0 NIELSEN, 1 FIXED_FACTOR */
    int32_t damping;         /* 0 LEVENBERG, 1 MARQUARDT */
    int32_t verbose;         /* ignored */
    double loss_scale, gradient_tol, step_tol, relative_cost_tol, initial_lambda, min_lambda, max_lambda, lambda_factor;
    /* types.h:92-94.  Used by pl_estimate_absolute_pose's final bundle (robust.cc:103-123), its batched form and
     * pl_bundle_adjust_camera: the selected intrinsics of the camera move with the pose and are returned.  The two-view
     * refiners have no camera to move: ignored there, as in the reference. */
    int32_t refine_focal_length, refine_extra_params, refine_principal_point;
    int32_t reserved;
} pl_bundle_options;

/* types.h:108-126, 128-145, 170-175 folded into one record; fields that do not apply are ignored */
typedef struct {
    pl_ransac_options ransac;
    pl_bundle_options bundle;
    double max_error;         /* 12.0 absolute pose, 1.0 otherwise */
    int32_t real_focal_check; /* fundamental only */
    int32_t tangent_sampson;  /* relative pose: must be 0 (out of scope) */
    int32_t estimate_focal_length; /* pl_estimate_absolute_pose: robust.cc:47-54 - RANSAC over pose AND focal length (ransac_pnpf,
                                      P3.5Pf), the camera's focal length is replaced and refined in the final bundle; every
                                      other entry point: must be 0 */
    int32_t estimate_extra_params; /* must be 0 (radial distortion estimation: out of scope) */
    double min_fov;           /* types.h:126 AbsolutePoseOptions::min_fov, degrees (5.0): with estimate_focal_length / pl_ransac_pnpf the
                                 largest focal length a hypothesis may have is max|x| / tan(min_fov / 2) (absolute_pose.cc:159-177);
                                 <= 0 disables the bound; NaN is rejected (PL_ERR_INVALID).  Ignored by every other entry
                                 point, like in the reference.
                                 ABI note: this field was appended in PL_ABI_VERSION 4.  A caller must fill the record with
                                 pl_default_robust_options() and then change what it needs - a zero-initialised record
                                 means min_fov = 0 (NO bound) where the reference defaults to 5 degrees - and must be built
                                 against the header of the library it loads: check pl_abi_version() == PL_ABI_VERSION once. */
} pl_robust_options;

/* types.h:52-58 (+ the metric numerator and timing, which the reference does not report) */
typedef struct {
    uint64_t refinements, iterations, num_inliers;
    double inlier_ratio, model_score;
    uint64_t hypotheses;          /* minimal models scored against all N points in the main loop */
    uint64_t iterations_evaluated; /* iterations the device evaluated (>= iterations: speculative batches) */
    double seconds;                /* host wall time of the ransac_* part */
    double score_kernel_ms;        /* HIP-event time of the scoring kernel launches */
    uint32_t score_kernel_launches;
    uint32_t nan_hypotheses;       /* of the hypotheses the device evaluated: models with a NaN entry (the reference's
                                      P3P emits them for inconsistent samples; they have no inliers, utils.cc:36-65) */
} pl_ransac_stats;

/* misc/camera_models.h:56-60; supported ids: -1 NULL, 0 SIMPLE_PINHOLE, 1 PINHOLE, 4 OPENCV */
typedef struct {
    int32_t model_id, width, height, num_params;
    double params[12];
} pl_camera;

typedef struct {
    double q[4];
    double t[3];
} pl_camera_pose;

void pl_default_ransac_options(pl_ransac_options *o);
void pl_default_bundle_options(pl_bundle_options *o);
/* kind: 0 absolute pose (max_error 12), 1 relative pose, 2 fundamental, 3 homography (max_error 1) */
void pl_default_robust_options(pl_robust_options *o, int kind);

/* ---- device management ---- */
int pl_device_count(void);
int pl_set_device(int device);      /* device used by the calling thread from now on */
const char *pl_last_error(void);    /* thread-local description of the last failure */
const char *pl_version(void);
/* Layout version of the records in this header (bumped whenever a struct gains, loses or moves a field): 4 = round 4's
 * pl_robust_options.min_fov, 5 = round 5 (pl_set_lm_mode takes a mode 0 / 1 / 2, records unchanged).  A binding - C++, ctypes, cgo -
 * compares pl_abi_version() with the PL_ABI_VERSION it was written against before the first call that passes a record. */
#define PL_ABI_VERSION 5
int pl_abi_version(void);
/* Summation order of the non-linear refinements (optim/jacobian_accumulator.h:82-97 adds correspondence after correspondence).
 * 0 (default): poses and homographies - problems up to 256 correspondences are summed in the reference's order (refined models
 *    bit-identical), larger ones in tree order (refined models 1e-13 off; the decisions were identical in every test and soak);
 *    FUNDAMENTAL matrices - every sum in the reference's order at every size: FactorizedFundamentalMatrix (optim_utils.h:57-72)
 *    enters each refinement through an SVD and negates U / V by their determinants, so the SIGN of the F that is returned is the
 *    sign of a rounding-level singular value of the refinement's input - only bit-identical intermediate models give the
 *    reference's sign (1.8 x the refinement time at 10^4 correspondences with the truncated loss, none with Cauchy's).
 * 1: EVERY sum of every refinement in the reference's order at every size (k_lm_ordered) - refined models bit-identical at every
 *    n at 1.3 - 2 x the refinement time (6 x for homographies beyond 5000 correspondences).
 * 2: tree sums beyond 256 correspondences for every estimator, fundamental matrices included (their sign is then unpinned).
 * Process-wide; also POSELIB_AMD_LM_ORDERED = 0 / 1 / 2.  Returns the previous mode. */
int pl_set_lm_mode(int mode);

/* ---- robust front-ends (robust.h) ---- */
int pl_estimate_absolute_pose(const double *points2D, const double *points3D, size_t n, const pl_robust_options *opt,
                              pl_camera *camera, pl_camera_pose *pose, uint8_t *inliers, pl_ransac_stats *stats);
int pl_estimate_relative_pose(const double *points2D_1, const double *points2D_2, size_t n, const pl_camera *camera1,
                              const pl_camera *camera2, const pl_robust_options *opt, pl_camera_pose *pose,
                              uint8_t *inliers, pl_ransac_stats *stats);
/* robust.h:84-90 estimate_shared_focal_relative_pose (robust.cc:366-424): pixel coordinates, principal point pp[2]; the image pair
 * of the reference is (pose, SIMPLE_PINHOLE {focal, pp[0], pp[1]} for both cameras).  focal is read only with
 * ransac.score_initial_model (image_pair->camera1.focal()). */
int pl_estimate_shared_focal_relative_pose(const double *points2D_1, const double *points2D_2, size_t n, const double *pp,
                                           const pl_robust_options *opt, pl_camera_pose *pose, double *focal, uint8_t *inliers,
                                           pl_ransac_stats *stats);
int pl_estimate_fundamental(const double *points2D_1, const double *points2D_2, size_t n, const pl_robust_options *opt,
                            double *F /* 9, column-major */, uint8_t *inliers, pl_ransac_stats *stats);
int pl_estimate_homography(const double *points2D_1, const double *points2D_2, size_t n, const pl_robust_options *opt,
                           double *H /* 9, column-major */, uint8_t *inliers, pl_ransac_stats *stats);

/* ---- un-distortion as a stage of its own (BASELINE config 3: pixels of an OPENCV camera in front of the homography /
 * 7-point estimators, which take no camera).  Every point goes through Camera::unproject (misc/camera_models.h:98-102;
 * OPENCV: the iterative inverse of misc/camera_models.cc:972-990) and comes back as the pixel of the distortion-free
 * camera with the same focal lengths and principal point: out = (fx u + cx, fy v + cy).  points2D, out: N x 2. ---- */
int pl_undistort_points(const pl_camera *camera, const double *points2D, size_t n, double *out);

/* ---- batched front-end: an array of independent problems (BASELINE config 4: many image pairs) ----
 * Every item is one call of the matching pl_estimate_* above; `max_in_flight` problems (<= 0: 8) are worked on
 * concurrently by an internal pool of host threads, each with its own HIP stream and scratch arena, on the device the
 * calling thread selected with pl_set_device().  The reference has no such call: a PoseLib user loops over
 * estimate_*() (robust.h) from a thread pool, the Python wrappers release the GIL for that purpose. */
typedef struct {
    int32_t kind;            /* 0 absolute pose, 1 relative pose, 2 fundamental, 3 homography, 4 relative pose with a shared
                              * unknown focal length (pl_estimate_shared_focal_relative_pose).  Kind-4 items and kind-0 items with
                              * estimate_focal_length advance in lock-step groups of their own since round 5 (one launch sequence per
                              * group, every item bit-identical to its single call).  Kinds 0 - 3 with PROSAC, a warm start
                              * (ransac.score_initial_model) or an OPENCV camera are group members like any other since round 6;
                              * what still runs one at a time: min_iterations > 4096, fewer correspondences than sample size + 4,
                              * more than 16384, warm starts of the focal-length kinds (pl_last_batch_report counts them) */
    int32_t status;          /* out: PL_OK or the error of this item */
    const double *a;         /* points2D (kind 0) / points2D_1: N x 2 */
    const double *b;         /* points3D: N x 3 (kind 0) / points2D_2: N x 2 */
    size_t n;
    const pl_robust_options *opt;
    pl_camera *camera1;      /* kind 0: in/out camera; kind 1: first camera; kind 4: SIMPLE_PINHOLE {focal, cx, cy} in/out; otherwise NULL */
    const pl_camera *camera2; /* kind 1: second camera; otherwise NULL */
    void *model;             /* in/out: pl_camera_pose (kinds 0, 1, 4) or double[9] column-major (kinds 2, 3) */
    uint8_t *inliers;        /* n bytes */
    pl_ransac_stats *stats;
} pl_batch_item;
/* returns PL_OK if every item succeeded, otherwise the status of the first failing item */
int pl_estimate_batch(pl_batch_item *items, size_t count, int max_in_flight);
/* The same call over SEVERAL devices of the node from one process (north_star: independent problems round-robined across the
 * GPUs; SURVEY 5 "one process drives <= 8 GPUs"): item i runs on devices[i mod num_devices], every entry of the list is served by
 * a host thread and a worker pool of its own (`max_in_flight` workers each), the results land in the caller's arrays - no
 * collective is needed inside one process.  devices == NULL or num_devices <= 0: every visible device.  An entry may repeat a
 * device (the list {0, 0} runs two half-batches side by side on device 0); every item is bit-identical to its single call
 * whatever the list.  The process-per-GPU form (one rank per device, RCCL gather of the result records) is the Python
 * harness's: poselib_amd/sharding.py, bench.py --gpus N; INTEGRATION.md says which to use when. */
int pl_estimate_batch_devices(pl_batch_item *items, size_t count, const int *devices, int num_devices, int max_in_flight);
/* What the calling thread's last pl_estimate_batch / pl_estimate_batch_devices / pl_ransac_batch did with its items.  Items in
 * lock-step groups share one launch sequence (the advertised rate); `solo` items were outside the group path from the start
 * (options or sizes a group does not take - see pl_batch_item / pl_ransac_batch) and `fallback` items were handed back by their
 * group: both kinds ran through the single-problem entry points, correct but at that path's rate - a batch that is mostly solo
 * items runs ~20 x slower than one that is grouped, and this report is where a caller sees it. */
typedef struct {
    uint64_t items;         /* items of the call */
    uint64_t grouped;       /* ... in the lock-step groups of kinds 0 - 3 */
    uint64_t focal_grouped; /* ... in the groups of the two focal-length estimators */
    uint64_t solo;          /* ... run one at a time from the start */
    uint64_t fallback;      /* of the grouped ones: handed back by their group and re-run one at a time */
} pl_batch_report;
void pl_last_batch_report(pl_batch_report *out);

/* ---- RANSAC entry points on normalised points (robust/ransac.h) ---- */
int pl_ransac_pnp(const double *x, const double *X, size_t n, const pl_robust_options *opt, pl_camera_pose *pose,
                  uint8_t *inliers, pl_ransac_stats *stats);
/* robust/ransac.h:52-54 ransac_pnpf (FocalAbsolutePoseEstimator, estimators/absolute_pose.h:69-113): pose and focal length of a
 * SIMPLE_PINHOLE camera whose principal point is the origin of the image points x.  The model is reset before the loop as in
 * the reference (ransac.cc:61-66); opt->min_fov bounds the focal length (absolute_pose.h:78).  PROSAC sampling: samples drawn on the host. */
int pl_ransac_pnpf(const double *x, const double *X, size_t n, const pl_robust_options *opt, pl_camera_pose *pose, double *focal,
                   uint8_t *inliers, pl_ransac_stats *stats);
int pl_ransac_relpose(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, pl_camera_pose *pose,
                      uint8_t *inliers, pl_ransac_stats *stats);
/* robust/ransac.h:71-73 ransac_shared_focal_relpose (SharedFocalRelativePoseEstimator, estimators/relative_pose.h:148-175; solver:
 * the 6-point shared-focal problem of solvers/relpose_6pt_focal.h): relative pose of two views of ONE camera with unknown focal
 * length; x1, x2 relative to the principal point.  pose / focal: the initial model when ransac.score_initial_model is set (otherwise
 * reset as in ransac.cc:185-190), the result on return.  PROSAC sampling (sampling.cc:85-136): samples drawn on the host, as for pl_ransac_pnpf. */
int pl_ransac_shared_focal_relpose(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, pl_camera_pose *pose,
                                   double *focal, uint8_t *inliers, pl_ransac_stats *stats);
/* robust/bundle.h:108-111 refine_shared_focal_relpose (SharedFocalRelativePoseRefiner, optim/relative.h:488-592): pose and the
 * shared focal length refined on all n correspondences (Sampson error of F = K^-1 E K^-1); both in / out. */
int pl_refine_shared_focal_relpose(const double *x1, const double *x2, size_t n, const pl_bundle_options *opt, pl_camera_pose *pose,
                                   double *focal, uint32_t *lm_iterations);

int pl_ransac_fundamental(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, double *F,
                          uint8_t *inliers, pl_ransac_stats *stats);
int pl_ransac_homography(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, double *H,
                         uint8_t *inliers, pl_ransac_stats *stats);

/* ---- device-resident problems (inputs stay in HBM across calls; what bench.py times) ---- */
typedef struct pl_problem pl_problem;
/* kind as in pl_default_robust_options; a = first point set (N x 2), b = second (N x 3 for kind 0, else N x 2) */
int pl_problem_create(int kind, const double *a, const double *b, size_t n, pl_problem **out);
void pl_problem_destroy(pl_problem *p);
/* model: pl_camera_pose for kinds 0/1, double[9] column-major for kinds 2/3 */
int pl_ransac_run(pl_problem *p, const pl_robust_options *opt, void *model, uint8_t *inliers, pl_ransac_stats *stats);

/* ---- many device-resident problems at once: every item is one pl_ransac_run (a loop over ransac_pnp / ransac_relpose /
 * ransac_fundamental / ransac_homography of robust/ransac.h:45-66 on the reference side), same results bit for bit.
 * Items of the same kind advance in lock-step groups of `group_size` problems (<= 0: 16, at most 64): one launch sequence
 * per group and batch of iterations instead of one per problem, which is what keeps the device busy when thousands of
 * problems are queued (bench.py's throughput mode).  `max_in_flight` host threads (<= 0: 4), each with its own stream,
 * work on different groups.  Items a group cannot take (fewer correspondences than sample size + 4, more than 16384) run
 * through pl_ransac_run; PROSAC and warm starts are group members since round 6.  The problems may live on DIFFERENT devices (pl_set_device before
 * pl_problem_create): every item runs on the device that holds its problem, one host thread and one worker pool per device
 * side by side (round 6). ---- */
typedef struct {
    pl_problem *problem;
    const pl_robust_options *opt;
    void *model;            /* out: pl_camera_pose (kinds 0, 1) or double[9] column-major (kinds 2, 3); in as well when
                               opt->ransac.score_initial_model is set */
    uint8_t *inliers;       /* optional, problem size bytes */
    pl_ransac_stats *stats; /* optional */
    int32_t status;         /* out: PL_OK or the error of this item */
    int32_t reserved;
} pl_ransac_item;
int pl_ransac_batch(pl_ransac_item *items, size_t count, int max_in_flight, int group_size);

/* ---- ONE problem across several GPUs: every rank holds the same correspondences (its own pl_problem on
 * its own device) and calls pl_ransac_run_sharded with the same options.  Each batch of iterations is cut into `world`
 * contiguous ranges; a rank draws the whole batch's sample positions (integer work, replicated) but generates and
 * scores only its range.  Two exchange steps per batch: the ranks all-gather their improving hypotheses (<= 7 KB per
 * rank, another message only when a rank has more than 32 of them), deal the triggered local optimisations out
 * among themselves (job j on rank j mod world) and all-gather the refined models (0.2 KB per job); every rank then
 * replays the sequential loop of ransac_impl.h:157-201 on the merged data, so all ranks return the same model, mask
 * and stats, identical to the single-device run.  `allgather` is the caller's
 * collective (RCCL / gloo through torch.distributed, MPI, or shared memory between threads): it must copy `bytes` bytes
 * from `send` of rank r to `recv + r * bytes` on every rank, and return 0. */
typedef int (*pl_allgather_fn)(void *user, const void *send, void *recv, size_t bytes);
typedef struct pl_shard {
    int32_t rank, world;
    pl_allgather_fn allgather;
    void *user;
} pl_shard;
int pl_ransac_run_sharded(pl_problem *p, const pl_robust_options *opt, const pl_shard *shard, void *model,
                          uint8_t *inliers, pl_ransac_stats *stats);

/* Score one model against the resident correspondences with the estimator's MSAC score
 * (robust/utils.cc:36-65, 158-239, 300-329 through estimators' score_model()).  model as in pl_ransac_run. */
int pl_score_model(pl_problem *p, const void *model, double max_error, uint64_t *inlier_count, double *score);
/* Diagnostic entry (no counterpart in the reference): `n` models - pl_camera_pose[n] for kinds 0/1, double[n][9]
 * column-major for kinds 2/3 - through the STREAMING scorer of the batched main loop, i.e. through the conservative
 * pre-filters (fp16/MFMA or fp32) in front of the exact fp64 evaluation, instead of the sequential scorer behind
 * pl_score_model.  counts / scores: n entries each (scores summed in tree order: equal to pl_score_model's to rounding,
 * counts exactly).  path_used: 2 = matrix-core filter (k_score_mfma: absolute pose; k_score_mfma2: Sampson scores on
 * coordinates bounded by 8), 1 = fp32 filter (k_score_queue), 0 = no filter
 * (threshold / coordinates outside the filters' range, or POSELIB_AMD_NO_PREFILTER).  tests/ plants adversarial
 * models and correspondences here to check that a filter never drops an inlier. */
int pl_debug_score_stream(pl_problem *p, const void *models, size_t n, double max_error, uint32_t *counts,
                          double *scores, int32_t *path_used);
/* Diagnostic entry: the scalar math of the device kernels, element-wise on `n` doubles - fn 0: the cube of the LM's
 * Nielsen update (optim/lm_impl.h:124 std::pow(., 3)), 1: sqrt, 2: reciprocal, 3 .. 6: cbrt / cos / sin / acos of
 * pl_libm.h.  tests/ compares them with the host's libm bit for bit. */
int pl_debug_device_math(int fn, const double *x, size_t n, double *out);
/* Non-linear refinement of one model on the resident correspondences (robust/bundle.h:41-170:
 * bundle_adjust / refine_relpose / refine_fundamental / refine_homography).  camera: absolute pose only
 * (NULL pointer = identity camera, i.e. normalised image points).  mask: optional N bytes, refine on the
 * flagged correspondences only. */
int pl_refine_model(pl_problem *p, const pl_bundle_options *opt, const pl_camera *camera, const uint8_t *mask,
                    void *model, uint32_t *lm_iterations);

/* bundle_adjust(x, X, Image *image, BundleOptions) - the Image overload of robust/bundle.h:47-49 / bundle.cc:94-113 - on
 * an absolute-pose problem whose 2-D points are PIXELS: pose and, per opt->refine_focal_length / refine_principal_point /
 * refine_extra_params, the camera's intrinsics are refined together (robust/optim/absolute.h:49-171); both in / out. */
int pl_bundle_adjust_camera(pl_problem *p, const pl_bundle_options *opt, pl_camera *camera, const uint8_t *mask,
                            pl_camera_pose *pose, uint32_t *lm_iterations);

/* ---- minimal solvers (solvers/ headers); unit bearing vectors in, solutions out; return = #solutions or <0 ---- */
int pl_p3p(const double *x /* 3x3 */, const double *X /* 3x3 */, pl_camera_pose *out /* 4 */);
int pl_relpose_5pt(const double *x1 /* 5x3 */, const double *x2 /* 5x3 */, pl_camera_pose *out /* 40 */);
int pl_essential_matrix_5pt(const double *x1, const double *x2, double *E /* 10 x 9 column-major */);
int pl_relpose_7pt(const double *x1 /* 7x3 */, const double *x2 /* 7x3 */, double *F /* 3 x 9 column-major */);
int pl_homography_4pt(const double *x1 /* 4x3 */, const double *x2 /* 4x3 */, double *H /* 9 column-major */);
/* the solvers of the two focal-length estimators (interfaces of solvers/p35pf.h:39-54 and solvers/relpose_6pt_focal.h:12-13; the
 * algorithms restate the reference's action-matrix templates since round 6 and return its roots in its order).  pl_p35pf: x = four image points relative to the principal point
 * (4 x 2; of the fourth only x is used), X = 4 x 3; out / focals: room for 10.  pl_relpose_6pt_shared_focal: six pairs of unit
 * bearings (6 x 3 each); out / focals: room for 60, in the reference's order.  Return = #solutions or < 0. */
int pl_p35pf(const double *x, const double *X, pl_camera_pose *out, double *focals);
int pl_relpose_6pt_shared_focal(const double *x1, const double *x2, pl_camera_pose *out, double *focals);
/* batched: kind 0 = P3.5Pf (in: count x [x 4 x 2 | X 4 x 3]; 10 slots), kind 1 = 6-point shared focal (in: count x [x1 6 x 3 |
 * x2 6 x 3]; 60 slots); out_models: count x slots x 8 doubles (q[4] t[3] focal), out_counts: count. */
int pl_solve_focal_batch(int kind, const double *in, size_t count, double *out_models, uint32_t *out_counts);
/* batched form: `count` independent minimal problems, one GPU lane each.
 * in: count x (2*K*3) doubles ([first set K x 3][second set K x 3]); out_models: count x max_models x 24 doubles
 * (model records of 24 doubles: q[4] t[3] M[9 row-major] + 8 doubles of internal fp32 shadow, see
 * poselib_amd/csrc/pl_math.h); out_counts: count. */
int pl_solve_batch(int kind, const double *in, size_t count, double *out_models, uint32_t *out_counts);

#ifdef __cplusplus
}
#endif
#endif /* POSELIB_AMD_H_ */
