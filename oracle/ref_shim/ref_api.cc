// ORACLE — TEST INFRASTRUCTURE ONLY.
// C interface over the REFERENCE'S OWN hot-path sources (compiled in place from /root/reference against the
// Eigen-API shim in oracle/eigen_shim; see oracle/Makefile.ref).  Used by tests/test_oracle_vs_reference.py to
// pin the oracle's restatement.  What is the reference's here: the sampler (robust/sampling.cc), the RANSAC loop
// template (robust/ransac_impl.h), the minimal solvers (solvers/p3p.cc, relpose_5pt.cc, relpose_7pt.cc,
// homography_4pt.cc, misc/univariate.cc, misc/sturm.h, misc/essential.cc), scoring and inlier masks
// (robust/utils.cc), CameraPose / quaternion helpers, the LM refiners (robust/optim/*.h, robust_loss.h) and the
// camera models (misc/camera_models.cc).  What is NOT: Eigen (shim), the estimator classes'
// the estimator classes and ransac.cc / robust.cc / bundle.cc translation units themselves (they pull in every other
// solver and refiner family): their few lines for this path are mirrored by the adapters below, which call the
// reference's sampler, solvers, scoring, refiner classes and LM loop.  REF_USE_ORACLE_LM=1 swaps the
// adapters' refine_model() onto the oracle's LM (to separate solver from LM differences when a test fails).
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
#include <PoseLib/robust/optim/absolute.h>
#include <PoseLib/robust/optim/fundamental.h>
#include <PoseLib/robust/optim/homography.h>
#include <PoseLib/robust/optim/jacobian_accumulator.h>
#include <PoseLib/robust/optim/lm_impl.h>
#include <PoseLib/robust/optim/relative.h>
#include <PoseLib/robust/robust_loss.h>

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
orc_bundle_opt lo_opt(double max_error) { // estimators/absolute_pose.cc:61-64
    orc_bundle_opt b;
    std::memset(&b, 0, sizeof(b));
    b.max_iterations = 25;
    b.loss_type = 1; // TRUNCATED
    b.loss_scale = max_error;
    b.gradient_tol = 1e-12;
    b.step_tol = 1e-8;
    b.relative_cost_tol = 1e-10;
    b.initial_lambda = 1e-3;
    b.min_lambda = 1e-10;
    b.max_lambda = 1e10;
    b.lambda_factor = 10.0;
    return b;
}
std::vector<double> flat2(const std::vector<Point2D> &v) {
    std::vector<double> f(2 * v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        f[2 * i] = v[i](0);
        f[2 * i + 1] = v[i](1);
    }
    return f;
}
std::vector<double> flat3(const std::vector<Point3D> &v) {
    std::vector<double> f(3 * v.size());
    for (size_t i = 0; i < v.size(); ++i)
        for (int d = 0; d < 3; ++d)
            f[3 * i + d] = v[i](d);
    return f;
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
IterationCallback le_zach_callback(const BundleOptions &opt) { // bundle.cc:53-77, non-verbose branches
    if (opt.loss_type == BundleOptions::TRUNCATED_LE_ZACH)
        return [](const BundleStats &, RobustLoss *loss_fn) {
            static_cast<TruncatedLossLeZach *>(loss_fn)->mu *= TruncatedLossLeZach::alpha;
        };
    return nullptr;
}
// The four refinement entry points as bundle.cc instantiates them for unit weights (bundle.cc:94-103, 206-213,
// 313-323, 394-401): the REFERENCE's refiner classes, Jacobian accumulator, robust losses and LM loop
// (robust/optim/{absolute,relative,fundamental,homography,jacobian_accumulator,lm_impl}.h, robust_loss.h,
// misc/camera_models.cc).  bundle.cc itself is not compiled: it also instantiates every other refiner family.
BundleStats ref_lm_abs(const std::vector<Point2D> &x, const std::vector<Point3D> &X, Image *image, const BundleOptions &opt) {
    std::vector<size_t> camera_refine_idx = image->camera.get_param_refinement_idx(opt);
    UniformWeightVector weights;
    AbsolutePoseRefiner<UniformWeightVector> refiner(x, X, camera_refine_idx, weights);
    return lm_impl<decltype(refiner)>(refiner, image, opt, le_zach_callback(opt));
}
BundleStats ref_lm_rel(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2, CameraPose *pose, const BundleOptions &opt) {
    UniformWeightVector weights;
    PinholeRelativePoseRefiner<UniformWeightVector> refiner(x1, x2, weights);
    return lm_impl<decltype(refiner)>(refiner, pose, opt, le_zach_callback(opt));
}
BundleStats ref_lm_fund(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2, Eigen::Matrix3d *F, const BundleOptions &opt) {
    FactorizedFundamentalMatrix factorized(*F);
    UniformWeightVector weights;
    PinholeFundamentalRefiner<UniformWeightVector> refiner(x1, x2, weights);
    BundleStats stats = lm_impl<decltype(refiner)>(refiner, &factorized, opt, le_zach_callback(opt));
    *F = factorized.F();
    return stats;
}
BundleStats ref_lm_hom(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2, Eigen::Matrix3d *H, const BundleOptions &opt) {
    UniformWeightVector weights;
    PinholeHomographyRefiner<UniformWeightVector> refiner(x1, x2, weights);
    return lm_impl<decltype(refiner)>(refiner, H, opt, le_zach_callback(opt));
}
BundleOptions lo_bundle(double max_error) { // estimators/absolute_pose.cc:61-64 (same in the other three)
    BundleOptions b;
    b.loss_type = BundleOptions::LossType::TRUNCATED;
    b.loss_scale = max_error;
    b.max_iterations = 25;
    return b;
}
bool use_oracle_lm() { return std::getenv("REF_USE_ORACLE_LM") != nullptr; }

// Adapters with the reference's estimator concept (ransac_impl.h:77-97).  generate_models / score_model follow
// estimators/absolute_pose.cc:46-58, relative_pose.cc:48-60 and :384-403, homography.cc:36-52 line by line and
// call the REFERENCE's sampler, solvers, scoring and (refine_model) refiners + LM loop.
struct AbsEst {
    AbsEst(const RansacOptions &ro, double max_error, const std::vector<Point2D> &x_, const std::vector<Point3D> &X_)
        : sample_sz(3), num_data(x_.size()), thr(max_error), x(x_), X(X_), sampler(num_data, sample_sz, ro),
          fx(flat2(x_)), fX(flat3(X_)) {
        xs.resize(3);
        Xs.resize(3);
        sample.resize(3);
    }
    void generate_models(std::vector<CameraPose> *models) {
        models->clear();
        sampler.generate_sample(&sample);
        for (size_t k = 0; k < sample_sz; ++k) {
            xs[k] = x[sample[k]].homogeneous().normalized();
            Xs[k] = X[sample[k]];
        }
        p3p(xs, Xs, models);
        hyp += models->size();
    }
    double score_model(const CameraPose &pose, size_t *inlier_count) const {
        return compute_msac_score(pose, x, X, thr * thr, inlier_count);
    }
    void refine_model(CameraPose *pose) const {
        if (!use_oracle_lm()) {
            Image image;
            image.pose = *pose;
            image.camera.model_id = NullCameraModel::model_id; // bundle.cc:84-92
            ref_lm_abs(x, X, &image, lo_bundle(thr));
            *pose = image.pose;
            return;
        }
        double p[7];
        pose_out(*pose, p);
        orc_camera cam;
        std::memset(&cam, 0, sizeof(cam));
        cam.model_id = -1;
        orc_bundle_opt b = lo_opt(thr);
        orc_bundle_adjust(fx.data(), fX.data(), num_data, &cam, p, &b, nullptr);
        *pose = pose_in(p);
    }
    size_t sample_sz, num_data;
    double thr;
    const std::vector<Point2D> &x;
    const std::vector<Point3D> &X;
    RandomSampler sampler;
    std::vector<double> fx, fX;
    std::vector<Eigen::Vector3d> xs, Xs;
    std::vector<size_t> sample;
    size_t hyp = 0;
};

struct TwoViewBase {
    TwoViewBase(size_t K, const RansacOptions &ro, double max_error, const std::vector<Point2D> &a,
                const std::vector<Point2D> &b)
        : sample_sz(K), num_data(a.size()), thr(max_error), x1(a), x2(b), sampler(num_data, sample_sz, ro),
          f1(flat2(a)), f2(flat2(b)) {
        x1s.resize(K);
        x2s.resize(K);
        sample.resize(K);
    }
    void draw() {
        sampler.generate_sample(&sample);
        for (size_t k = 0; k < sample_sz; ++k) {
            x1s[k] = x1[sample[k]].homogeneous().normalized();
            x2s[k] = x2[sample[k]].homogeneous().normalized();
        }
    }
    size_t sample_sz, num_data;
    double thr;
    const std::vector<Point2D> &x1;
    const std::vector<Point2D> &x2;
    RandomSampler sampler;
    std::vector<double> f1, f2;
    std::vector<Eigen::Vector3d> x1s, x2s;
    std::vector<size_t> sample;
    size_t hyp = 0;
};
struct RelEst : TwoViewBase {
    using TwoViewBase::TwoViewBase;
    void generate_models(std::vector<CameraPose> *models) {
        models->clear();
        draw();
        relpose_5pt(x1s, x2s, models);
        hyp += models->size();
    }
    double score_model(const CameraPose &pose, size_t *cnt) const {
        return compute_sampson_msac_score(pose, x1, x2, thr * thr, cnt);
    }
    void refine_model(CameraPose *pose) const { // relative_pose.cc:62-86
        std::vector<char> inl;
        int num_inl = get_inliers(*pose, x1, x2, 5 * (thr * thr), &inl);
        if (num_inl <= 5)
            return;
        if (!use_oracle_lm()) {
            std::vector<Point2D> a, b;
            a.reserve(num_inl), b.reserve(num_inl);
            for (size_t k = 0; k < x1.size(); ++k)
                if (inl[k])
                    a.push_back(x1[k]), b.push_back(x2[k]);
            ref_lm_rel(a, b, pose, lo_bundle(thr));
            return;
        }
        std::vector<double> a, b;
        for (size_t k = 0; k < x1.size(); ++k)
            if (inl[k]) {
                a.push_back(x1[k](0)), a.push_back(x1[k](1));
                b.push_back(x2[k](0)), b.push_back(x2[k](1));
            }
        double p[7];
        pose_out(*pose, p);
        orc_bundle_opt bo = lo_opt(thr);
        orc_refine_relpose(a.data(), b.data(), a.size() / 2, p, &bo, nullptr);
        *pose = pose_in(p);
    }
};
struct FundEst : TwoViewBase {
    using TwoViewBase::TwoViewBase;
    bool rfc = false;
    void generate_models(std::vector<Eigen::Matrix3d> *models) {
        models->clear();
        draw();
        relpose_7pt(x1s, x2s, models);
        if (rfc)
            for (int i = models->size() - 1; i >= 0; i--)
                if (!calculate_RFC((*models)[i]))
                    models->erase(models->begin() + i);
        hyp += models->size();
    }
    double score_model(const Eigen::Matrix3d &F, size_t *cnt) const {
        return compute_sampson_msac_score(F, x1, x2, thr * thr, cnt);
    }
    void refine_model(Eigen::Matrix3d *F) const {
        if (!use_oracle_lm()) {
            ref_lm_fund(x1, x2, F, lo_bundle(thr));
            return;
        }
        double m[9];
        mat_out(*F, m);
        orc_bundle_opt bo = lo_opt(thr);
        orc_refine_fundamental(f1.data(), f2.data(), num_data, m, &bo, nullptr);
        *F = mat_in(m);
    }
};
struct HomEst : TwoViewBase {
    using TwoViewBase::TwoViewBase;
    void generate_models(std::vector<Eigen::Matrix3d> *models) {
        models->clear();
        draw();
        Eigen::Matrix3d H;
        int sols = homography_4pt(x1s, x2s, &H, true);
        if (sols > 0)
            models->push_back(H);
        hyp += models->size();
    }
    double score_model(const Eigen::Matrix3d &H, size_t *cnt) const {
        return compute_homography_msac_score(H, x1, x2, thr * thr, cnt);
    }
    void refine_model(Eigen::Matrix3d *H) const {
        if (!use_oracle_lm()) {
            ref_lm_hom(x1, x2, H, lo_bundle(thr));
            return;
        }
        double m[9];
        mat_out(*H, m);
        orc_bundle_opt bo = lo_opt(thr);
        orc_refine_homography(f1.data(), f2.data(), num_data, m, &bo, nullptr);
        *H = mat_in(m);
    }
};

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

void ref_bundle_adjust(const double *x, const double *X, size_t n, const orc_camera *cam, double *pose7,
                       const orc_bundle_opt *opt, orc_bundle_stats *st) {
    Image image;
    image.pose = pose_in(pose7);
    image.camera.model_id = cam ? cam->model_id : -1;
    if (cam) {
        image.camera.width = cam->width;
        image.camera.height = cam->height;
        image.camera.params.assign(cam->params, cam->params + cam->num_params);
    }
    bstats_out(ref_lm_abs(pts2(x, n), pts3(X, n), &image, bopt(*opt)), st);
    pose_out(image.pose, pose7);
}
void ref_refine_relpose(const double *x1, const double *x2, size_t n, double *pose7, const orc_bundle_opt *opt,
                        orc_bundle_stats *st) {
    CameraPose pose = pose_in(pose7);
    bstats_out(ref_lm_rel(pts2(x1, n), pts2(x2, n), &pose, bopt(*opt)), st);
    pose_out(pose, pose7);
}
void ref_refine_fundamental(const double *x1, const double *x2, size_t n, double *F9, const orc_bundle_opt *opt,
                            orc_bundle_stats *st) {
    Eigen::Matrix3d F = mat_in(F9);
    bstats_out(ref_lm_fund(pts2(x1, n), pts2(x2, n), &F, bopt(*opt)), st);
    mat_out(F, F9);
}
void ref_refine_homography(const double *x1, const double *x2, size_t n, double *H9, const orc_bundle_opt *opt,
                           orc_bundle_stats *st) {
    Eigen::Matrix3d H = mat_in(H9);
    bstats_out(ref_lm_hom(pts2(x1, n), pts2(x2, n), &H, bopt(*opt)), st);
    mat_out(H, H9);
}

// ransac.cc:44-57, 142-154, 248-262, 300-314 with the adapters above
void ref_ransac_pnp(const double *x, const double *X, size_t n, const orc_robust_opt *opt, double *pose7,
                    uint8_t *inliers, orc_stats *st) {
    const std::vector<Point2D> a = pts2(x, n);
    const std::vector<Point3D> b = pts3(X, n);
    const RansacOptions ro = ropt(opt->ransac);
    CameraPose best = pose_in(pose7);
    if (!ro.score_initial_model) {
        best.q << 1.0, 0.0, 0.0, 0.0;
        best.t.setZero();
    }
    AbsEst est(ro, opt->max_error, a, b);
    const RansacStats s = ransac<AbsEst>(est, ro, &best);
    std::vector<char> m;
    get_inliers(best, a, b, opt->max_error * opt->max_error, &m);
    pose_out(best, pose7);
    mask_out(m, inliers);
    stats_out(s, est.hyp, st);
}
void ref_ransac_relpose(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *pose7,
                        uint8_t *inliers, orc_stats *st) {
    const std::vector<Point2D> a = pts2(x1, n), b = pts2(x2, n);
    const RansacOptions ro = ropt(opt->ransac);
    CameraPose best = pose_in(pose7);
    if (!ro.score_initial_model) {
        best.q << 1.0, 0.0, 0.0, 0.0;
        best.t.setZero();
    }
    RelEst est(5, ro, opt->max_error, a, b);
    const RansacStats s = ransac<RelEst>(est, ro, &best);
    std::vector<char> m;
    get_inliers(best, a, b, opt->max_error * opt->max_error, &m);
    pose_out(best, pose7);
    mask_out(m, inliers);
    stats_out(s, est.hyp, st);
}
void ref_ransac_fundamental(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *F9,
                            uint8_t *inliers, orc_stats *st) {
    const std::vector<Point2D> a = pts2(x1, n), b = pts2(x2, n);
    const RansacOptions ro = ropt(opt->ransac);
    Eigen::Matrix3d best = mat_in(F9);
    if (!ro.score_initial_model)
        best.setIdentity();
    FundEst est(7, ro, opt->max_error, a, b);
    est.rfc = opt->real_focal_check != 0;
    const RansacStats s = ransac<FundEst, Eigen::Matrix3d>(est, ro, &best);
    std::vector<char> m;
    get_inliers(best, a, b, opt->max_error * opt->max_error, &m);
    mat_out(best, F9);
    mask_out(m, inliers);
    stats_out(s, est.hyp, st);
}
void ref_ransac_homography(const double *x1, const double *x2, size_t n, const orc_robust_opt *opt, double *H9,
                           uint8_t *inliers, orc_stats *st) {
    const std::vector<Point2D> a = pts2(x1, n), b = pts2(x2, n);
    const RansacOptions ro = ropt(opt->ransac);
    Eigen::Matrix3d best = mat_in(H9);
    if (!ro.score_initial_model)
        best.setIdentity();
    HomEst est(4, ro, opt->max_error, a, b);
    const RansacStats s = ransac<HomEst, Eigen::Matrix3d>(est, ro, &best);
    std::vector<char> m;
    get_homography_inliers(best, a, b, opt->max_error * opt->max_error, &m);
    mat_out(best, H9);
    mask_out(m, inliers);
    stats_out(s, est.hyp, st);
}

} // extern "C"
