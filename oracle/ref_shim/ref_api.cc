// ORACLE — TEST INFRASTRUCTURE ONLY.
// C interface over the REFERENCE'S OWN sources, compiled in place from /root/reference against the Eigen-API shim in
// oracle/eigen_shim (recipe: oracle/Makefile.ref).  Used by tests/test_oracle_vs_reference.py to pin the oracle's
// restatement.  Everything behind these wrappers is PoseLib's code: the front-ends (robust.cc), the RANSAC entry
// points (robust/ransac.cc), the estimator classes (robust/estimators/*.cc), the loop template
// (robust/ransac_impl.h), the sampler, the minimal solvers, scoring and masks (robust/utils.cc), the refiners
// and LM loop (robust/bundle.cc, robust/optim/*.h, robust_loss.*), the camera models (misc/camera_models.cc).
// Only Eigen is not the real one.  Reference translation units the shim cannot carry (solver families that are
// not on this path and need Eigen::EigenSolver etc.) are left out; the symbols they would define become traps
// (oracle/ref_shim/make_traps.sh), which abort if anything ever calls them.
#include <PoseLib/solvers/p35pf.h>
#include <PoseLib/solvers/relpose_6pt_focal.h>
#include <PoseLib/camera_pose.h>
#include <PoseLib/misc/essential.h>
#include <PoseLib/robust/ransac_impl.h>
#include <PoseLib/robust/sampling.h>
#include <PoseLib/robust/utils.h>
#include <PoseLib/solvers/homography_4pt.h>
#include <PoseLib/solvers/p3p.h>
#include <PoseLib/solvers/relpose_5pt.h>
#include <PoseLib/solvers/relpose_7pt.h>
#include <PoseLib/misc/univariate.h>
#include <PoseLib/types.h>

#include "../oracle.h"

#include <PoseLib/misc/camera_models.h>
#include <PoseLib/robust.h>
#include <PoseLib/robust/bundle.h>
#include <PoseLib/robust/ransac.h>

#include <chrono>
#include <cstdlib>
#include <cstring>

using namespace poselib;


namespace {

std::vector<Point2D> pts2(const double *p, size_t n) {
    std::vector<Point2D> v(n);
    for (size_t i = 0; i < n; ++i)
        v[i] = Point2D(p[2 * i], p[2 * i + 1]);
    return v;
}
std::vector<Point3D> pts3(const double *p, size_t n) {
    std::vector<Point3D> v(n);
    for (size_t i = 0; i < n; ++i)
        v[i] = Point3D(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    return v;
}
CameraPose pose_in(const double *p) {
    CameraPose r;
    r.q << p[0], p[1], p[2], p[3];
    r.t << p[4], p[5], p[6];
    return r;
}
void pose_out(const CameraPose &r, double *p) {
    for (int i = 0; i < 4; ++i)
        p[i] = r.q(i);
    for (int i = 0; i < 3; ++i)
        p[4 + i] = r.t(i);
}
Eigen::Matrix3d mat_in(const double *m) {
    Eigen::Matrix3d A;
    std::memcpy(A.data(), m, sizeof(double) * 9);
    return A;
}
void mat_out(const Eigen::Matrix3d &A, double *m) { std::memcpy(m, A.data(), sizeof(double) * 9); }
RansacOptions ropt(const orc_ransac_opt &o) {
    RansacOptions r;
    r.max_iterations = o.max_iterations;
    r.min_iterations = o.min_iterations;
    r.dyn_num_trials_mult = o.dyn_num_trials_mult;
    r.success_prob = o.success_prob;
    r.seed = o.seed;
    r.progressive_sampling = o.progressive_sampling != 0;
    r.max_prosac_iterations = o.max_prosac_iterations;
    r.score_initial_model = o.score_initial_model != 0;
    return r;
}
BundleOptions bopt(const orc_bundle_opt &o) {
    BundleOptions b;
    b.max_iterations = o.max_iterations;
    b.loss_type = static_cast<BundleOptions::LossType>(o.loss_type);
    b.loss_scale = o.loss_scale;
    b.gradient_tol = o.gradient_tol;
    b.step_tol = o.step_tol;
    b.relative_cost_tol = o.relative_cost_tol;
    b.initial_lambda = o.initial_lambda;
    b.min_lambda = o.min_lambda;
    b.max_lambda = o.max_lambda;
    b.lambda_update = static_cast<BundleOptions::LambdaUpdateType>(o.lambda_update);
    b.lambda_factor = o.lambda_factor;
    b.damping = static_cast<BundleOptions::DampingType>(o.damping);
    b.refine_focal_length = (o.refine_flags & 1) != 0;
    b.refine_principal_point = (o.refine_flags & 2) != 0;
    b.refine_extra_params = (o.refine_flags & 4) != 0;
    return b;
}
void bstats_out(const BundleStats &s, orc_bundle_stats *o) {
    if (!o)
        return;
    o->iterations = s.iterations;
    o->initial_cost = s.initial_cost;
    o->cost = s.cost;
    o->lambda = s.lambda;
    o->nu = s.nu;
    o->invalid_steps = s.invalid_steps;
    o->step_norm = s.step_norm;
    o->grad_norm = s.grad_norm;
}
Camera cam_in(const orc_camera *c) {
    Camera cam;
    cam.model_id = c ? c->model_id : -1;
    if (c) {
        cam.width = c->width;
        cam.height = c->height;
        cam.params.assign(c->params, c->params + c->num_params);
    }
    return cam;
}
template <typename Opt> Opt robust_in(const orc_robust_opt *o) {
    Opt r;
    r.ransac = ropt(o->ransac);
    r.bundle = bopt(o->bundle);
    r.max_error = o->max_error;
    return r;
}
AbsolutePoseOptions abs_in(const orc_robust_opt *o) {
    AbsolutePoseOptions r = robust_in<AbsolutePoseOptions>(o);
    r.estimate_focal_length = o->estimate_focal_length != 0;
    r.min_fov = o->min_fov;
    return r;
}
RelativePoseOptions rel_in(const orc_robust_opt *o) {
    RelativePoseOptions r = robust_in<RelativePoseOptions>(o);
    r.real_focal_check = o->real_focal_check != 0;
    return r;
}
void stats_out(const RansacStats &s, size_t hyp, orc_stats *o) {
    o->refinements = s.refinements;
    o->iterations = s.iterations;
    o->num_inliers = s.num_inliers;
    o->inlier_ratio = s.inlier_ratio;
    o->model_score = s.model_score;
    o->hypotheses = hyp;
    o->seconds = 0;
}
void mask_out(const std::vector<char> &m, uint8_t *out) {
    for (size_t i = 0; i < m.size(); ++i)
        out[i] = m[i] ? 1 : 0;
}
void bearings(const double *p, int n, std::vector<Eigen::Vector3d> &out) {
    out.resize(n);
    for (int i = 0; i < n; ++i)
        out[i] = Eigen::Vector3d(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
}

struct MockEstimator { // tests/ransac_test.cc:12-28
    MockEstimator(size_t n, size_t k, size_t c) : sample_sz(k), num_data(n), inlier_count(c) {}
    void generate_models(std::vector<int> *models) const { models->push_back(0); }
    double score_model(const int &, size_t *c) const {
        *c = inlier_count;
        return 0.0;
    }
    void refine_model(int *) const {}
    size_t sample_sz, num_data, inlier_count;
};

} // namespace

extern "C" {

void ref_sampler_draw(uint64_t seed, uint64_t N, uint64_t K, uint64_t n_samples, int32_t prosac,
                      uint64_t max_prosac_iterations, uint64_t *out_idx, uint64_t *state_after) {
    RandomSampler s(N, K, seed, prosac != 0, (int)max_prosac_iterations);
    std::vector<size_t> sample(K);
    for (uint64_t i = 0; i < n_samples; ++i) {
        s.generate_sample(&sample);
        for (uint64_t k = 0; k < K; ++k)
            out_idx[i * K + k] = sample[k];
    }
    if (state_after)
        *state_after = s.state;
}
double ref_all_inlier_probability(uint64_t inl, uint64_t N, uint64_t K) {
    return detail::all_inlier_sample_probability(inl, N, K);
}
uint64_t ref_dynamic_max_iter(uint64_t inl, uint64_t N, uint64_t K, double log_fail, double mult, uint64_t mn, uint64_t mx) {
    return detail::compute_dynamic_max_iter(inl, N, K, log_fail, mult, mn, mx);
}
void ref_mock_ransac(uint64_t num_data, uint64_t sample_sz, uint64_t inlier_count, const orc_ransac_opt *opt, orc_stats *out) {
    MockEstimator est(num_data, sample_sz, inlier_count);
    int best = -1;
    const RansacStats s = ransac<MockEstimator, int>(est, ropt(*opt), &best);
    stats_out(s, s.iterations, out);
}
int ref_solve_cubic_single_real(double c2, double c1, double c0, double *root) {
    return univariate::solve_cubic_single_real(c2, c1, c0, *root) ? 1 : 0;
}
int ref_solve_cubic_real(double c2, double c1, double c0, double *roots) {
    return univariate::solve_cubic_real(c2, c1, c0, roots);
}
int ref_p3p(const double *x, const double *X, double *poses) {
    std::vector<Eigen::Vector3d> xb, Xp;
    bearings(x, 3, xb);
    bearings(X, 3, Xp);
    std::vector<CameraPose> out;
    const int n = p3p(xb, Xp, &out);
    for (int i = 0; i < n; ++i)
        pose_out(out[i], poses + 7 * i);
    return n;
}
int ref_essential_5pt(const double *x1, const double *x2, double *E) {
    std::vector<Eigen::Vector3d> a, b;
    bearings(x1, 5, a);
    bearings(x2, 5, b);
    std::vector<Eigen::Matrix3d> out;
    const int n = relpose_5pt(a, b, &out);
    for (int i = 0; i < n; ++i)
        mat_out(out[i], E + 9 * i);
    return n;
}
int ref_relpose_5pt(const double *x1, const double *x2, double *poses) {
    std::vector<Eigen::Vector3d> a, b;
    bearings(x1, 5, a);
    bearings(x2, 5, b);
    std::vector<CameraPose> out;
    const int n = relpose_5pt(a, b, &out);
    for (int i = 0; i < n; ++i)
        pose_out(out[i], poses + 7 * i);
    return n;
}
int ref_relpose_7pt(const double *x1, const double *x2, double *F) {
    std::vector<Eigen::Vector3d> a, b;
    bearings(x1, 7, a);
    bearings(x2, 7, b);
    std::vector<Eigen::Matrix3d> out;
    const int n = relpose_7pt(a, b, &out);
    for (int i = 0; i < n; ++i)
        mat_out(out[i], F + 9 * i);
    return n;
}
int ref_homography_4pt(const double *x1, const double *x2, double *H, int check) {
    std::vector<Eigen::Vector3d> a, b;
    bearings(x1, 4, a);
    bearings(x2, 4, b);
    Eigen::Matrix3d out;
    out.setIdentity();
    const int n = homography_4pt(a, b, &out, check != 0);
    mat_out(out, H);
    return n;
}
double ref_score_reproj(const double *pose7, const double *x, const double *X, size_t n, double sq_thr, uint64_t *cnt) {
    size_t c = 0;
    const double s = compute_msac_score(pose_in(pose7), pts2(x, n), pts3(X, n), sq_thr, &c);
    *cnt = c;
    return s;
}
double ref_score_sampson_pose(const double *pose7, const double *x1, const double *x2, size_t n, double sq_thr, uint64_t *cnt) {
    size_t c = 0;
    const double s = compute_sampson_msac_score(pose_in(pose7), pts2(x1, n), pts2(x2, n), sq_thr, &c);
    *cnt = c;
    return s;
}
double ref_score_sampson_F(const double *F9, const double *x1, const double *x2, size_t n, double sq_thr, uint64_t *cnt) {
    size_t c = 0;
    const double s = compute_sampson_msac_score(mat_in(F9), pts2(x1, n), pts2(x2, n), sq_thr, &c);
    *cnt = c;
    return s;
}
double ref_score_homography(const double *H9, const double *x1, const double *x2, size_t n, double sq_thr, uint64_t *cnt) {
    size_t c = 0;
    const double s = compute_homography_msac_score(mat_in(H9), pts2(x1, n), pts2(x2, n), sq_thr, &c);
    *cnt = c;
    return s;
}
void ref_inliers_reproj(const double *pose7, const double *x, const double *X, size_t n, double sq_thr, uint8_t *mask) {
    std::vector<char> m;
    get_inliers(pose_in(pose7), pts2(x, n), pts3(X, n), sq_thr, &m);
    mask_out(m, mask);
}
void ref_inliers_sampson_pose(const double *pose7, const double *x1, const double *x2, size_t n, double sq_thr, uint8_t *mask) {
    std::vector<char> m;
    get_inliers(pose_in(pose7), pts2(x1, n), pts2(x2, n), sq_thr, &m);
    mask_out(m, mask);
}
void ref_inliers_sampson_F(const double *F9, const double *x1, const double *x2, size_t n, double sq_thr, uint8_t *mask) {
    std::vector<char> m;
    get_inliers(mat_in(F9), pts2(x1, n), pts2(x2, n), sq_thr, &m);
    mask_out(m, mask);
}
void ref_inliers_homography(const double *H9, const double *x1, const double *x2, size_t n, double sq_thr, uint8_t *mask) {
    std::vector<char> m;
    get_homography_inliers(mat_in(H9), pts2(x1, n), pts2(x2, n), sq_thr, &m);
    mask_out(m, mask);
}
double ref_normalize_points(double *x1, double *x2, size_t n, double *T1, double *T2, int normalize_scale,
                            int normalize_centroid, int shared_scale) {
    std::vector<Point2D> a = pts2(x1, n), b = pts2(x2, n);
    Eigen::Matrix3d A, B;
    const double s = normalize_points(a, b, A, B, normalize_scale != 0, normalize_centroid != 0, shared_scale != 0);
    for (size_t i = 0; i < n; ++i) {
        x1[2 * i] = a[i](0), x1[2 * i + 1] = a[i](1);
        x2[2 * i] = b[i](0), x2[2 * i + 1] = b[i](1);
    }
    mat_out(A, T1);
    mat_out(B, T2);
    return s;
}

void ref_unproject(const orc_camera *cam, const double *xp, size_t n, double *out) {
    const Camera c = cam_in(cam);
    for (size_t i = 0; i < n; ++i) {
        Eigen::Vector3d d;
        c.unproject(Eigen::Vector2d(xp[2 * i], xp[2 * i + 1]), &d);
        out[2 * i] = d(0) / d(2);
        out[2 * i + 1] = d(1) / d(2);
    }
}

// robust/bundle.h entry points
void ref_bundle_adjust(const double *x, const double *X, size_t n, const orc_camera *cam, double *pose7,
                       const orc_bundle_opt *opt, orc_bundle_stats *st) {
    Image image;
    image.pose = pose_in(pose7);
    image.camera = cam_in(cam);
    bstats_out(bundle_adjust(pts2(x, n), pts3(X, n), &image, bopt(*opt)), st);
    pose_out(image.pose, pose7);
}
void ref_bundle_adjust_camera(const double *x, const double *X, size_t n, orc_camera *cam, double *pose7,
                              const orc_bundle_opt *opt, orc_bundle_stats *st) {
    Image image;
    image.pose = pose_in(pose7);
    image.camera = cam_in(cam);
    bstats_out(bundle_adjust(pts2(x, n), pts3(X, n), &image, bopt(*opt)), st);
    pose_out(image.pose, pose7);
    for (size_t i = 0; i < image.camera.params.size(); ++i)
        cam->params[i] = image.camera.params[i];
}
void ref_refine_relpose(const double *x1, const double *x2, size_t n, double *pose7, const orc_bundle_opt *opt,
                        orc_bundle_stats *st) {
    CameraPose pose = pose_in(pose7);
    bstats_out(refine_relpose(pts2(x1, n), pts2(x2, n), &pose, bopt(*opt)), st);
    pose_out(pose, pose7);
}
void ref_refine_fundamental(const double *x1, const double *x2, size_t n, double *F9, const orc_bundle_opt *opt,
                            orc_bundle_stats *st) {
    Eigen::Matrix3d F = mat_in(F9);
    bstats_out(refine_fundamental(pts2(x1, n), pts2(x2, n), &F, bopt(*opt)), st);
    mat_out(F, F9);
}
void ref_refine_homography(const double *x1, const double *x2, size_t n, double *H9, const orc_bundle_opt *opt,
                           orc_bundle_stats *st) {
    Eigen::Matrix3d H = mat_in(H9);
    bstats_out(refine_homography(pts2(x1, n), pts2(x2, n), &H, bopt(*opt)), st);
    mat_out(H, H9);
}

// robust/ransac.h entry points (hypotheses are not observable from outside: reported as 0; `seconds` is the wall time of
// the reference's ransac_* call itself, measured here the way the oracle measures its own loop)
struct CallTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
void ref_ransac_pnp(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7,
                    uint8_t *inliers, orc_stats *st) {
    CameraPose best = pose_in(pose7);
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = ransac_pnp(pts2(x, n), pts3(X, n), robust_in<AbsolutePoseOptions>(opt), &best, &m);
    const double call_seconds = timer.seconds();
    pose_out(best, pose7);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}
// ---- the focal-length estimator: the reference's own FocalAbsolutePoseEstimator with its default solver
// P3.5Pf (robust/ransac.cc:58-75, estimators/absolute_pose.cc:73-160, solvers/p35pf.cc; Eigen::EigenSolver from the shim) ----
int ref_p35pf(const double *x /* 4 x 2 */, const double *X /* 4 x 3 */, double *poses7 /* 10 x 7 */, double *focals /* 10 */) {
    std::vector<CameraPose> poses;
    std::vector<double> f;
    const int n = p35pf(pts2(x, 4), pts3(X, 4), &poses, &f, true);
    for (int i = 0; i < n && i < 10; ++i) {
        pose_out(poses[i], poses7 + 7 * i);
        focals[i] = f[i];
    }
    return n;
}
void ref_ransac_pnpf(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7, double *focal,
                     uint8_t *inliers, orc_stats *st) {
    Image best;
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = ransac_pnpf(pts2(x, n), pts3(X, n), abs_in(opt), &best, &m);
    const double call_seconds = timer.seconds();
    pose_out(best.pose, pose7);
    *focal = best.camera.focal();
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}
// shared unknown focal length, two views (robust.cc:366-430, robust/ransac.cc:183-197, estimators/relative_pose.cc
// SharedFocalRelativePoseEstimator, solvers/relpose_6pt_focal.cc): pixel coordinates, principal point pp
void ref_estimate_shared_focal_relative_pose(const double *x1, const double *x2, size_t n, const double *pp2,
                                             const orc_robust_opt *opt, double *pose7, double *focal, uint8_t *inliers,
                                             orc_stats *st) {
    ImagePair pair;
    pair.pose = pose_in(pose7);
    pair.camera1 = Camera(SimplePinholeCameraModel::model_id, std::vector<double>{*focal, 0.0, 0.0}, -1, -1);
    pair.camera2 = pair.camera1;
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = estimate_shared_focal_relative_pose(pts2(x1, n), pts2(x2, n), Point2D(pp2[0], pp2[1]), rel_in(opt), &pair, &m);
    const double call_seconds = timer.seconds();
    pose_out(pair.pose, pose7);
    *focal = pair.camera1.focal();
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}
// the minimal solver alone (solvers/relpose_6pt_focal.cc:1083-1144): six pairs of unit bearings (principal point at the origin,
// unit focal length), up to 15 * 4 (pose, focal) models in the solver's order
int ref_relpose_6pt_shared_focal(const double *x1 /* 6 x 3 */, const double *x2, double *poses7 /* 60 x 7 */, double *focals /* 60 */) {
    std::vector<Eigen::Vector3d> a(6), b(6);
    for (int i = 0; i < 6; ++i) {
        a[i] = Eigen::Vector3d(x1[3 * i], x1[3 * i + 1], x1[3 * i + 2]);
        b[i] = Eigen::Vector3d(x2[3 * i], x2[3 * i + 1], x2[3 * i + 2]);
    }
    ImagePairVector models;
    const int n = relpose_6pt_shared_focal(a, b, &models);
    for (int i = 0; i < n && i < 60; ++i) {
        pose_out(models[i].pose, poses7 + 7 * i);
        focals[i] = models[i].camera1.focal();
    }
    return n;
}
// ransac_shared_focal_relpose (robust/ransac.cc:182-203) on points relative to the principal point
void ref_ransac_shared_focal_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                                     double *focal, uint8_t *inliers, orc_stats *st) {
    ImagePair best;
    best.pose = pose_in(pose7);
    best.camera1 = Camera(SimplePinholeCameraModel::model_id, std::vector<double>{*focal, 0.0, 0.0}, -1, -1);
    best.camera2 = best.camera1;
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = ransac_shared_focal_relpose(pts2(x1, n), pts2(x2, n), rel_in(opt), &best, &m);
    const double call_seconds = timer.seconds();
    pose_out(best.pose, pose7);
    *focal = best.camera1.focal();
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}
// refine_shared_focal_relpose (robust/bundle.cc:277-297, SharedFocalRelativePoseRefiner of optim/relative.h)
void ref_refine_shared_focal_relpose(const double *x1, const double *x2, size_t n, double *pose7, double *focal,
                                     const orc_bundle_opt *opt, orc_bundle_stats *st) {
    ImagePair pair;
    pair.pose = pose_in(pose7);
    pair.camera1 = Camera(SimplePinholeCameraModel::model_id, std::vector<double>{*focal, 0.0, 0.0}, -1, -1);
    pair.camera2 = pair.camera1;
    bstats_out(refine_shared_focal_relpose(pts2(x1, n), pts2(x2, n), &pair, bopt(*opt)), st);
    pose_out(pair.pose, pose7);
    *focal = pair.camera1.focal();
}
void ref_ransac_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                        uint8_t *inliers, orc_stats *st) {
    CameraPose best = pose_in(pose7);
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = ransac_relpose(pts2(x1, n), pts2(x2, n), rel_in(opt), &best, &m);
    const double call_seconds = timer.seconds();
    pose_out(best, pose7);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}
void ref_ransac_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                            uint8_t *inliers, orc_stats *st) {
    Eigen::Matrix3d best = mat_in(F9);
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = ransac_fundamental(pts2(x1, n), pts2(x2, n), rel_in(opt), &best, &m);
    const double call_seconds = timer.seconds();
    mat_out(best, F9);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}
void ref_ransac_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                           uint8_t *inliers, orc_stats *st) {
    Eigen::Matrix3d best = mat_in(H9);
    std::vector<char> m;
    const CallTimer timer;
    const RansacStats s = ransac_homography(pts2(x1, n), pts2(x2, n), robust_in<HomographyOptions>(opt), &best, &m);
    const double call_seconds = timer.seconds();
    mat_out(best, H9);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
    st->seconds = call_seconds;
}

// robust.h front-ends
void ref_estimate_absolute_pose(const double *p2d, const double *p3d, size_t n, const orc_robust_opt *opt,
                                orc_camera *cam, double *pose7, uint8_t *inliers, orc_stats *st) {
    Image image;
    image.pose = pose_in(pose7);
    image.camera = cam_in(cam);
    std::vector<char> m;
    const RansacStats s = estimate_absolute_pose(pts2(p2d, n), pts3(p3d, n), abs_in(opt), &image, &m);
    pose_out(image.pose, pose7);
    cam->num_params = static_cast<int32_t>(image.camera.params.size());
    for (size_t i = 0; i < image.camera.params.size() && i < 12; ++i)
        cam->params[i] = image.camera.params[i];
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
}
void ref_estimate_relative_pose(const double *x1, const double *x2, size_t n, const orc_camera *cam1,
                                const orc_camera *cam2, const orc_robust_opt *opt, double *pose7, uint8_t *inliers,
                                orc_stats *st) {
    CameraPose pose = pose_in(pose7);
    std::vector<char> m;
    const RansacStats s =
        estimate_relative_pose(pts2(x1, n), pts2(x2, n), cam_in(cam1), cam_in(cam2), rel_in(opt), &pose, &m);
    pose_out(pose, pose7);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
}
void ref_estimate_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                              uint8_t *inliers, orc_stats *st) {
    Eigen::Matrix3d F = mat_in(F9);
    std::vector<char> m;
    const RansacStats s = estimate_fundamental(pts2(x1, n), pts2(x2, n), rel_in(opt), &F, &m);
    mat_out(F, F9);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
}
void ref_estimate_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                             uint8_t *inliers, orc_stats *st) {
    Eigen::Matrix3d H = mat_in(H9);
    std::vector<char> m;
    const RansacStats s = estimate_homography(pts2(x1, n), pts2(x2, n), robust_in<HomographyOptions>(opt), &H, &m);
    mat_out(H, H9);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, 0, st);
}

} // extern "C"
