// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).  extern "C" surface of the oracle.
#include "../oracle.h"

#include "estimators.h"
#include "scoring.h"
#include "solvers.h"

#include <chrono>
#include <cstring>

using namespace orc;

namespace {

std::vector<V2> pts2(const double *p, size_t n) {
    std::vector<V2> v(n);
    for (size_t i = 0; i < n; ++i)
        v[i] = V2{p[2 * i], p[2 * i + 1]};
    return v;
}
std::vector<V3> pts3(const double *p, size_t n) {
    std::vector<V3> v(n);
    for (size_t i = 0; i < n; ++i)
        v[i] = V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    return v;
}
Pose pose_in(const double *p) {
    Pose r;
    for (int i = 0; i < 4; ++i)
        r.q[i] = p[i];
    r.t = V3{p[4], p[5], p[6]};
    return r;
}
void pose_out(const Pose &r, double *p) {
    for (int i = 0; i < 4; ++i)
        p[i] = r.q[i];
    p[4] = r.t.x, p[5] = r.t.y, p[6] = r.t.z;
}
M3 mat_in(const double *m) { // column-major
    M3 A;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            A.m[i][j] = m[3 * j + i];
    return A;
}
void mat_out(const M3 &A, double *m) {
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            m[3 * j + i] = A.m[i][j];
}
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
    b.loss_type = o.loss_type;
    b.loss_scale = o.loss_scale;
    b.gradient_tol = o.gradient_tol;
    b.step_tol = o.step_tol;
    b.relative_cost_tol = o.relative_cost_tol;
    b.initial_lambda = o.initial_lambda;
    b.min_lambda = o.min_lambda;
    b.max_lambda = o.max_lambda;
    b.lambda_update = o.lambda_update;
    b.lambda_factor = o.lambda_factor;
    b.damping = o.damping;
    b.refine_focal_length = (o.refine_flags & 1) != 0;
    b.refine_principal_point = (o.refine_flags & 2) != 0;
    b.refine_extra_params = (o.refine_flags & 4) != 0;
    return b;
}
Camera cam_in(const orc_camera *c) {
    Camera cam;
    cam.model_id = c->model_id;
    cam.width = c->width;
    cam.height = c->height;
    cam.params.assign(c->params, c->params + c->num_params);
    return cam;
}
void cam_out(const Camera &cam, orc_camera *c) {
    for (size_t i = 0; i < cam.params.size(); ++i)
        c->params[i] = cam.params[i];
}
void stats_out(const RansacStats &s, const LoopTrace &tr, double secs, orc_stats *o) {
    o->refinements = s.refinements;
    o->iterations = s.iterations;
    o->num_inliers = s.num_inliers;
    o->inlier_ratio = s.inlier_ratio;
    o->model_score = s.model_score;
    o->hypotheses = tr.hypotheses;
    o->seconds = secs;
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
void mask_out(const std::vector<char> &m, uint8_t *out) {
    for (size_t i = 0; i < m.size(); ++i)
        out[i] = m[i] ? 1 : 0;
}
double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
void bearings(const double *p, int n, V3 *out) {
    for (int i = 0; i < n; ++i)
        out[i] = V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
}

struct MockEstimator { // tests/ransac_test.cc:12-28
    size_t sample_sz, num_data;
    uint64_t inlier_count;
    void generate(std::vector<int> *m) const { m->push_back(0); }
    double score(const int &, uint64_t *c) const {
        *c = inlier_count;
        return 0.0;
    }
    void refine(int *) const {}
};

} // namespace

extern "C" {

void orc_sampler_draw(uint64_t seed, uint64_t N, uint64_t K, uint64_t n_samples, int32_t prosac,
                      uint64_t max_prosac_iterations, uint64_t *out_idx, uint64_t *state_after) {
    RansacOptions o;
    o.seed = seed;
    o.progressive_sampling = prosac != 0;
    o.max_prosac_iterations = max_prosac_iterations;
    Sampler s(N, K, o);
    for (uint64_t i = 0; i < n_samples; ++i)
        s.next(out_idx + i * K);
    if (state_after)
        *state_after = s.state;
}
int32_t orc_random_int(uint64_t *state) { return splitmix_next_int(*state); }
double orc_all_inlier_probability(uint64_t inliers, uint64_t N, uint64_t K) {
    return prob_all_inlier_sample(inliers, N, K);
}
uint64_t orc_dynamic_max_iter(uint64_t inliers, uint64_t N, uint64_t K, double log_fail, double mult, uint64_t min_it,
                              uint64_t max_it) {
    return dynamic_iteration_bound(inliers, N, K, log_fail, mult, min_it, max_it);
}
void orc_mock_ransac(uint64_t num_data, uint64_t sample_sz, uint64_t inlier_count, const orc_ransac_opt *opt,
                     orc_stats *out) {
    MockEstimator est{sample_sz, num_data, inlier_count};
    int best = -1;
    LoopTrace tr;
    // the mock appends (never clears) like the reference's; lo_ransac clears between iterations
    const RansacStats s = lo_ransac(est, ropt(*opt), &best, &tr);
    stats_out(s, tr, 0.0, out);
}

int orc_solve_cubic_single_real(double c2, double c1, double c0, double *root) {
    return cubic_one_real_root(c2, c1, c0, *root) ? 1 : 0;
}
int orc_solve_cubic_real(double c2, double c1, double c0, double *roots) { return cubic_real_roots(c2, c1, c0, roots); }
int orc_sturm_roots(const double *coeffs, int degree, double *roots) { return sturm_real_roots(coeffs, degree, roots); }
int orc_sturm_roots_tol(const double *coeffs, int degree, double tol, double *roots) { return sturm_real_roots(coeffs, degree, roots, tol); }

int orc_p3p(const double *x, const double *X, double *poses) {
    V3 xb[3], Xp[3];
    bearings(x, 3, xb);
    bearings(X, 3, Xp);
    Pose out[4];
    const int n = p3p(xb, Xp, out);
    for (int i = 0; i < n; ++i)
        pose_out(out[i], poses + 7 * i);
    return n;
}
int orc_p35pf(const double *x, const double *X, double *poses7, double *focals) {
    V2 xi[4];
    V3 Xp[4];
    for (int i = 0; i < 4; ++i) {
        xi[i] = V2{x[2 * i], x[2 * i + 1]};
        Xp[i] = V3{X[3 * i], X[3 * i + 1], X[3 * i + 2]};
    }
    Pose out[10];
    const int n = p35pf(xi, Xp, out, focals);
    for (int i = 0; i < n; ++i)
        pose_out(out[i], poses7 + 7 * i);
    return n;
}
int orc_relpose_6pt_shared_focal(const double *x1, const double *x2, double *poses7, double *focals) {
    V3 a[6], b[6];
    for (int i = 0; i < 6; ++i) { // unit bearings as they are (the estimator normalises, relative_pose.cc:157-160)
        a[i] = V3{x1[3 * i], x1[3 * i + 1], x1[3 * i + 2]};
        b[i] = V3{x2[3 * i], x2[3 * i + 1], x2[3 * i + 2]};
    }
    Pose out[60];
    const int n = relpose_6pt_shared_focal(a, b, out, focals);
    for (int i = 0; i < n; ++i)
        pose_out(out[i], poses7 + 7 * i);
    return n;
}
void orc_set_exact_cubes(int on) { set_exact_cubes(on != 0); }
int orc_essential_5pt(const double *x1, const double *x2, double *E) {
    V3 a[5], b[5];
    bearings(x1, 5, a);
    bearings(x2, 5, b);
    M3 out[10];
    const int n = essential_5pt(a, b, out);
    for (int i = 0; i < n; ++i)
        mat_out(out[i], E + 9 * i);
    return n;
}
int orc_relpose_5pt(const double *x1, const double *x2, double *poses) {
    V3 a[5], b[5];
    bearings(x1, 5, a);
    bearings(x2, 5, b);
    Pose out[40];
    const int n = relpose_5pt(a, b, out);
    for (int i = 0; i < n; ++i)
        pose_out(out[i], poses + 7 * i);
    return n;
}
int orc_relpose_7pt(const double *x1, const double *x2, double *F) {
    V3 a[7], b[7];
    bearings(x1, 7, a);
    bearings(x2, 7, b);
    M3 out[3];
    const int n = relpose_7pt(a, b, out);
    for (int i = 0; i < n; ++i)
        mat_out(out[i], F + 9 * i);
    return n;
}
int orc_homography_4pt(const double *x1, const double *x2, double *H, int check) {
    V3 a[4], b[4];
    bearings(x1, 4, a);
    bearings(x2, 4, b);
    M3 out;
    const int n = homography_4pt(a, b, &out, check != 0);
    mat_out(out, H);
    return n;
}
void orc_nullspace(const double *A, int rows, int cols, double *basis) { householder_complement(A, rows, cols, basis); }

double orc_score_reproj(const double *pose7, const double *x, const double *X, size_t n, double sq_thr, uint64_t *cnt) {
    return msac_reproj(pose_in(pose7), pts2(x, n), pts3(X, n), sq_thr, cnt);
}
double orc_score_sampson_pose(const double *pose7, const double *x1, const double *x2, size_t n, double sq_thr,
                              uint64_t *cnt) {
    return msac_sampson_pose(pose_in(pose7), pts2(x1, n), pts2(x2, n), sq_thr, cnt);
}
double orc_score_sampson_F(const double *F9, const double *x1, const double *x2, size_t n, double sq_thr,
                           uint64_t *cnt) {
    return msac_sampson_F(mat_in(F9), pts2(x1, n), pts2(x2, n), sq_thr, cnt);
}
double orc_score_homography(const double *H9, const double *x1, const double *x2, size_t n, double sq_thr,
                            uint64_t *cnt) {
    return msac_homography(mat_in(H9), pts2(x1, n), pts2(x2, n), sq_thr, cnt);
}
void orc_inliers_reproj(const double *pose7, const double *x, const double *X, size_t n, double sq_thr, uint8_t *mask) {
    std::vector<char> m;
    inliers_reproj(pose_in(pose7), pts2(x, n), pts3(X, n), sq_thr, &m);
    mask_out(m, mask);
}
void orc_inliers_sampson_pose(const double *pose7, const double *x1, const double *x2, size_t n, double sq_thr,
                              uint8_t *mask) {
    std::vector<char> m;
    inliers_sampson_pose(pose_in(pose7), pts2(x1, n), pts2(x2, n), sq_thr, &m);
    mask_out(m, mask);
}
void orc_inliers_sampson_F(const double *F9, const double *x1, const double *x2, size_t n, double sq_thr,
                           uint8_t *mask) {
    std::vector<char> m;
    inliers_sampson_F(mat_in(F9), pts2(x1, n), pts2(x2, n), sq_thr, &m);
    mask_out(m, mask);
}
void orc_inliers_homography(const double *H9, const double *x1, const double *x2, size_t n, double sq_thr,
                            uint8_t *mask) {
    std::vector<char> m;
    inliers_homography(mat_in(H9), pts2(x1, n), pts2(x2, n), sq_thr, &m);
    mask_out(m, mask);
}
double orc_normalize_points(double *x1, double *x2, size_t n, double *T1, double *T2, int normalize_scale,
                            int normalize_centroid, int shared_scale) {
    std::vector<V2> a = pts2(x1, n), b = pts2(x2, n);
    M3 A, B;
    const double s = normalize_points(a, b, A, B, normalize_scale != 0, normalize_centroid != 0, shared_scale != 0);
    for (size_t i = 0; i < n; ++i) {
        x1[2 * i] = a[i].x, x1[2 * i + 1] = a[i].y;
        x2[2 * i] = b[i].x, x2[2 * i + 1] = b[i].y;
    }
    mat_out(A, T1);
    mat_out(B, T2);
    return s;
}
void orc_unproject(const orc_camera *cam, const double *xp, size_t n, double *out) {
    const Camera c = cam_in(cam);
    for (size_t i = 0; i < n; ++i) {
        const V2 u = c.unproject(V2{xp[2 * i], xp[2 * i + 1]});
        out[2 * i] = u.x, out[2 * i + 1] = u.y;
    }
}

void orc_bundle_adjust(const double *x, const double *X, size_t n, const orc_camera *cam, double *pose7,
                       const orc_bundle_opt *opt, orc_bundle_stats *st) {
    Image im;
    im.pose = pose_in(pose7);
    im.camera = cam_in(cam);
    const BundleStats s = bundle_adjust(pts2(x, n), pts3(X, n), &im, bopt(*opt));
    pose_out(im.pose, pose7);
    bstats_out(s, st);
}
void orc_bundle_adjust_camera(const double *x, const double *X, size_t n, orc_camera *cam, double *pose7,
                              const orc_bundle_opt *opt, orc_bundle_stats *st) {
    Image im;
    im.pose = pose_in(pose7);
    im.camera = cam_in(cam);
    const BundleStats s = bundle_adjust(pts2(x, n), pts3(X, n), &im, bopt(*opt));
    pose_out(im.pose, pose7);
    cam_out(im.camera, cam);
    bstats_out(s, st);
}
void orc_refine_relpose(const double *x1, const double *x2, size_t n, double *pose7, const orc_bundle_opt *opt,
                        orc_bundle_stats *st) {
    Pose p = pose_in(pose7);
    const BundleStats s = refine_relpose(pts2(x1, n), pts2(x2, n), &p, bopt(*opt));
    pose_out(p, pose7);
    bstats_out(s, st);
}
void orc_refine_shared_focal_relpose(const double *x1, const double *x2, size_t n, double *pose7, double *focal,
                                     const orc_bundle_opt *opt, orc_bundle_stats *st) {
    ImagePair p;
    p.pose = pose_in(pose7);
    p.focal = *focal;
    const BundleStats s = refine_shared_focal_relpose(pts2(x1, n), pts2(x2, n), &p, bopt(*opt));
    pose_out(p.pose, pose7);
    *focal = p.focal;
    bstats_out(s, st);
}
void orc_refine_homography(const double *x1, const double *x2, size_t n, double *H9, const orc_bundle_opt *opt,
                           orc_bundle_stats *st) {
    M3 H = mat_in(H9);
    const BundleStats s = refine_homography(pts2(x1, n), pts2(x2, n), &H, bopt(*opt));
    mat_out(H, H9);
    bstats_out(s, st);
}
void orc_refine_fundamental(const double *x1, const double *x2, size_t n, double *F9, const orc_bundle_opt *opt,
                            orc_bundle_stats *st) {
    M3 F = mat_in(F9);
    const BundleStats s = refine_fundamental(pts2(x1, n), pts2(x2, n), &F, bopt(*opt));
    mat_out(F, F9);
    bstats_out(s, st);
}

// The SVD at the entry of the fundamental-matrix refinement (optim_utils.h:60), row-major 3x3 in and out.
void orc_svd3(const double *A9, double *U9, double *s3, double *V9) {
    M3 A, U, V;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A.m[i][j] = A9[3 * i + j];
    svd3(A, U, s3, V);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            U9[3 * i + j] = U.m[i][j], V9[3 * i + j] = V.m[i][j];
}

void orc_ransac_pnp(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7,
                    uint8_t *inliers, orc_stats *st) {
    AbsolutePoseOptions o;
    o.ransac = ropt(opt->ransac);
    o.bundle = bopt(opt->bundle);
    o.max_error = opt->max_error;
    const std::vector<V2> a = pts2(x, n);
    const std::vector<V3> b = pts3(X, n);
    Pose p = pose_in(pose7);
    std::vector<char> m;
    LoopTrace tr;
    const double t0 = now();
    const RansacStats s = ransac_pnp(a, b, o, &p, &m, &tr);
    const double t1 = now();
    pose_out(p, pose7);
    mask_out(m, inliers);
    stats_out(s, tr, t1 - t0, st);
}
static RelativePoseOptions relopt(const orc_robust_opt *opt) {
    RelativePoseOptions o;
    o.ransac = ropt(opt->ransac);
    o.bundle = bopt(opt->bundle);
    o.max_error = opt->max_error;
    o.real_focal_check = opt->real_focal_check != 0;
    return o;
}
static HomographyOptions homopt(const orc_robust_opt *opt) {
    HomographyOptions o;
    o.ransac = ropt(opt->ransac);
    o.bundle = bopt(opt->bundle);
    o.max_error = opt->max_error;
    return o;
}
void orc_ransac_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                        uint8_t *inliers, orc_stats *st) {
    const std::vector<V2> a = pts2(x1, n), b = pts2(x2, n);
    Pose p = pose_in(pose7);
    std::vector<char> m;
    LoopTrace tr;
    const double t0 = now();
    const RansacStats s = ransac_relpose(a, b, relopt(opt), &p, &m, &tr);
    const double t1 = now();
    pose_out(p, pose7);
    mask_out(m, inliers);
    stats_out(s, tr, t1 - t0, st);
}
void orc_ransac_shared_focal_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                                     double *focal, uint8_t *inliers, orc_stats *st) {
    const std::vector<V2> a = pts2(x1, n), b = pts2(x2, n);
    ImagePair p;
    p.pose = pose_in(pose7);
    p.focal = *focal;
    std::vector<char> m;
    LoopTrace tr;
    const double t0 = now();
    const RansacStats s = ransac_shared_focal_relpose(a, b, relopt(opt), &p, &m, &tr);
    const double t1 = now();
    pose_out(p.pose, pose7);
    *focal = p.focal;
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, tr, t1 - t0, st);
}
void orc_estimate_shared_focal_relative_pose(const double *x1, const double *x2, size_t n, const double *pp2,
                                             const orc_robust_opt *opt, double *pose7, double *focal, uint8_t *inliers,
                                             orc_stats *st) {
    const std::vector<V2> a = pts2(x1, n), b = pts2(x2, n);
    ImagePair p;
    p.pose = pose_in(pose7);
    p.focal = *focal;
    std::vector<char> m;
    const double t0 = now();
    const RansacStats s = estimate_shared_focal_relative_pose(a, b, V2{pp2[0], pp2[1]}, relopt(opt), &p, &m);
    const double t1 = now();
    pose_out(p.pose, pose7);
    *focal = p.focal;
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, LoopTrace(), t1 - t0, st);
}
void orc_ransac_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                            uint8_t *inliers, orc_stats *st) {
    const std::vector<V2> a = pts2(x1, n), b = pts2(x2, n);
    M3 F = mat_in(F9);
    std::vector<char> m;
    LoopTrace tr;
    const double t0 = now();
    const RansacStats s = ransac_fundamental(a, b, relopt(opt), &F, &m, &tr);
    const double t1 = now();
    mat_out(F, F9);
    mask_out(m, inliers);
    stats_out(s, tr, t1 - t0, st);
}
void orc_ransac_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                           uint8_t *inliers, orc_stats *st) {
    const std::vector<V2> a = pts2(x1, n), b = pts2(x2, n);
    M3 H = mat_in(H9);
    std::vector<char> m;
    LoopTrace tr;
    const double t0 = now();
    const RansacStats s = ransac_homography(a, b, homopt(opt), &H, &m, &tr);
    const double t1 = now();
    mat_out(H, H9);
    mask_out(m, inliers);
    stats_out(s, tr, t1 - t0, st);
}

void orc_ransac_pnpf(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7, double *focal,
                     uint8_t *inliers, orc_stats *st) {
    AbsolutePoseOptions o;
    o.ransac = ropt(opt->ransac);
    o.bundle = bopt(opt->bundle);
    o.max_error = opt->max_error;
    o.min_fov = opt->min_fov;
    Image best;
    std::vector<char> m;
    LoopTrace tr;
    const double t0 = now();
    const RansacStats s = ransac_pnpf(pts2(x, n), pts3(X, n), o, &best, &m, &tr);
    const double t1 = now();
    pose_out(best.pose, pose7);
    *focal = best.camera.focal();
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, tr, t1 - t0, st);
}
void orc_estimate_absolute_pose(const double *p2d, const double *p3d, size_t n, const orc_robust_opt *opt,
                                orc_camera *cam, double *pose7, uint8_t *inliers, orc_stats *st) {
    AbsolutePoseOptions o;
    o.ransac = ropt(opt->ransac);
    o.bundle = bopt(opt->bundle);
    o.max_error = opt->max_error;
    o.estimate_focal_length = opt->estimate_focal_length != 0;
    o.min_fov = opt->min_fov;
    Image im;
    im.pose = pose_in(pose7);
    im.camera = cam_in(cam);
    std::vector<char> m;
    const double t0 = now();
    const RansacStats s = estimate_absolute_pose(pts2(p2d, n), pts3(p3d, n), o, &im, &m);
    const double t1 = now();
    pose_out(im.pose, pose7);
    cam_out(im.camera, cam);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, LoopTrace(), t1 - t0, st);
}
void orc_estimate_relative_pose(const double *x1, const double *x2, size_t n, const orc_camera *cam1,
                                const orc_camera *cam2, const orc_robust_opt *opt, double *pose7, uint8_t *inliers,
                                orc_stats *st) {
    Pose p = pose_in(pose7);
    std::vector<char> m;
    const double t0 = now();
    const RansacStats s =
        estimate_relative_pose(pts2(x1, n), pts2(x2, n), cam_in(cam1), cam_in(cam2), relopt(opt), &p, &m);
    const double t1 = now();
    pose_out(p, pose7);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, LoopTrace(), t1 - t0, st);
}
void orc_estimate_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                              uint8_t *inliers, orc_stats *st) {
    M3 F = mat_in(F9);
    std::vector<char> m;
    const double t0 = now();
    const RansacStats s = estimate_fundamental(pts2(x1, n), pts2(x2, n), relopt(opt), &F, &m);
    const double t1 = now();
    mat_out(F, F9);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, LoopTrace(), t1 - t0, st);
}
void orc_estimate_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                             uint8_t *inliers, orc_stats *st) {
    M3 H = mat_in(H9);
    std::vector<char> m;
    const double t0 = now();
    const RansacStats s = estimate_homography(pts2(x1, n), pts2(x2, n), homopt(opt), &H, &m);
    const double t1 = now();
    mat_out(H, H9);
    m.resize(n, 0);
    mask_out(m, inliers);
    stats_out(s, LoopTrace(), t1 - t0, st);
}

} // extern "C"
