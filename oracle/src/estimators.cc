// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Citations: estimators.h.
#include "estimators.h"

#include "scoring.h"
#include "solvers.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace orc {

namespace {

BundleOptions lo_options(double max_error) { // absolute_pose.cc:61-64 (same in all four estimators)
    BundleOptions b;
    b.loss_type = LOSS_TRUNCATED;
    b.loss_scale = max_error;
    b.max_iterations = 25;
    return b;
}

struct AbsEstimator {
    const AbsolutePoseOptions &opt;
    const std::vector<V2> &x;
    const std::vector<V3> &X;
    Sampler sampler;
    size_t sample_sz = 3, num_data;
    AbsEstimator(const AbsolutePoseOptions &o, const std::vector<V2> &x_, const std::vector<V3> &X_)
        : opt(o), x(x_), X(X_), sampler(x_.size(), 3, o.ransac), num_data(x_.size()) {}
    void generate(std::vector<Pose> *models) {
        uint64_t s[3];
        sampler.next(s);
        V3 xs[3], Xs[3];
        for (int k = 0; k < 3; ++k) {
            xs[k] = bearing(x[s[k]]);
            Xs[k] = X[s[k]];
        }
        Pose sol[4];
        const int n = p3p(xs, Xs, sol);
        models->assign(sol, sol + n);
    }
    double score(const Pose &p, uint64_t *cnt) const { return msac_reproj(p, x, X, opt.max_error * opt.max_error, cnt); }
    void refine(Pose *p) const { bundle_adjust(x, X, p, lo_options(opt.max_error)); }
};

// utils.cc:66-100: MSAC score through a camera (here always the SIMPLE_PINHOLE camera of the focal estimator: f x + cx)
double msac_reproj_image(const Image &im, const std::vector<V2> &x, const std::vector<V3> &X, double sq_thr, uint64_t *inliers) {
    *inliers = 0;
    double score = 0.0;
    const M3 R = im.pose.R();
    const V3 &t = im.pose.t;
    const double f = im.camera.params[0], cx = im.camera.params[1], cy = im.camera.params[2];
    for (size_t k = 0; k < x.size(); ++k) {
        const double X0 = X[k].x, X1 = X[k].y, X2 = X[k].z;
        const double z0 = R.m[0][0] * X0 + R.m[0][1] * X1 + R.m[0][2] * X2 + t.x;
        const double z1 = R.m[1][0] * X0 + R.m[1][1] * X1 + R.m[1][2] * X2 + t.y;
        const double z2 = R.m[2][0] * X0 + R.m[2][1] * X1 + R.m[2][2] * X2 + t.z;
        if (z2 <= 0.0)
            continue;
        const double inv_z2 = 1.0 / z2;
        const double r0 = (f * (z0 * inv_z2) + cx) - x[k].x;
        const double r1 = (f * (z1 * inv_z2) + cy) - x[k].y;
        const double r_sq = r0 * r0 + r1 * r1;
        if (r_sq < sq_thr) {
            ++*inliers;
            score += r_sq;
        }
    }
    score += static_cast<double>(x.size() - *inliers) * sq_thr;
    return score;
}
// utils.cc:385-399 (hnormalized: a division, not the reciprocal of the score above)
void inliers_reproj_image(const Image &im, const std::vector<V2> &x, const std::vector<V3> &X, double sq_thr, std::vector<char> *mask) {
    mask->resize(x.size());
    const M3 R = im.pose.R();
    const double f = im.camera.params[0], cx = im.camera.params[1], cy = im.camera.params[2];
    for (size_t k = 0; k < x.size(); ++k) {
        const V3 Z = R * X[k] + im.pose.t;
        const double r0 = (f * (Z.x / Z.z) + cx) - x[k].x, r1 = (f * (Z.y / Z.z) + cy) - x[k].y;
        const double r2 = r0 * r0 + r1 * r1;
        (*mask)[k] = (r2 < sq_thr && Z.z > 0.0);
    }
}

// estimators/absolute_pose.{h:69-113, cc:71-177} with the defaults of the class: solver P3.5Pf, refine_minimal_sample and
// filter_minimal_sample off, inlier_scoring on.  The solver is the oracle's own (solvers_focal.cc) - the same solution SET as the
// reference's generated template to ~1e-7, in ascending order of the eigenvalue instead of the order of Eigen's EigenSolver
struct FocalAbsEstimator {
    const AbsolutePoseOptions &opt;
    const std::vector<V2> &x;
    const std::vector<V3> &X;
    Sampler sampler;
    size_t sample_sz = 4, num_data;
    double max_focal_length = -1.0;
    FocalAbsEstimator(const AbsolutePoseOptions &o, const std::vector<V2> &x_, const std::vector<V3> &X_)
        : opt(o), x(x_), X(X_), sampler(x_.size(), 4, o.ransac), num_data(x_.size()) {
        if (o.min_fov > 0) { // absolute_pose.cc:159-177
            double max_coord = 0.0;
            for (const V2 &p : x) {
                max_coord = std::max(max_coord, std::abs(p.x));
                max_coord = std::max(max_coord, std::abs(p.y));
            }
            max_focal_length = max_coord / std::tan(o.min_fov * M_PI / 180.0 / 2.0);
        }
    }
    void generate(std::vector<Image> *models) {
        uint64_t s[4];
        sampler.next(s);
        V2 xs[4];
        V3 Xs[4];
        for (int k = 0; k < 4; ++k) {
            xs[k] = x[s[k]];
            Xs[k] = X[s[k]];
        }
        Pose sol[10];
        double focals[10];
        const int n = p35pf(xs, Xs, sol, focals);
        models->clear();
        for (int i = 0; i < n; ++i) {
            if (focals[i] < 0)
                continue;
            if (max_focal_length >= 0 && focals[i] > max_focal_length)
                continue;
            Image im;
            im.pose = sol[i];
            im.camera.model_id = CAM_SIMPLE_PINHOLE;
            im.camera.params = {focals[i], 0.0, 0.0};
            models->push_back(im);
        }
    }
    double score(const Image &im, uint64_t *cnt) const { // absolute_pose.cc:128-143
        if (im.camera.focal() < 0)
            return std::numeric_limits<double>::max();
        double sc = msac_reproj_image(im, x, X, opt.max_error * opt.max_error, cnt);
        // inlier_scoring (default on): the outliers are charged a second time, associated as ((n - c) e) e
        sc += static_cast<double>(x.size() - *cnt) * opt.max_error * opt.max_error;
        if (max_focal_length > 0 && im.camera.focal() > max_focal_length)
            sc = std::numeric_limits<double>::max();
        return sc;
    }
    void refine(Image *im) const { // absolute_pose.cc:145-157
        BundleOptions b = lo_options(opt.max_error);
        b.refine_focal_length = true;
        bundle_adjust(x, X, im, b);
    }
};

struct RelEstimator {
    const RelativePoseOptions &opt;
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    Sampler sampler;
    size_t sample_sz = 5, num_data;
    RelEstimator(const RelativePoseOptions &o, const std::vector<V2> &a, const std::vector<V2> &b)
        : opt(o), x1(a), x2(b), sampler(a.size(), 5, o.ransac), num_data(a.size()) {}
    void generate(std::vector<Pose> *models) {
        uint64_t s[5];
        sampler.next(s);
        V3 a[5], b[5];
        for (int k = 0; k < 5; ++k) {
            a[k] = bearing(x1[s[k]]);
            b[k] = bearing(x2[s[k]]);
        }
        Pose sol[40];
        const int n = relpose_5pt(a, b, sol);
        models->assign(sol, sol + n);
    }
    double score(const Pose &p, uint64_t *cnt) const {
        return msac_sampson_pose(p, x1, x2, opt.max_error * opt.max_error, cnt);
    }
    void refine(Pose *p) const { // relative_pose.cc:62-86
        std::vector<char> mask;
        const int n = inliers_sampson_pose(*p, x1, x2, 5 * (opt.max_error * opt.max_error), &mask);
        if (n <= 5)
            return;
        std::vector<V2> a, b;
        a.reserve(n);
        b.reserve(n);
        for (size_t k = 0; k < x1.size(); ++k)
            if (mask[k]) {
                a.push_back(x1[k]);
                b.push_back(x2[k]);
            }
        refine_relpose(a, b, p, lo_options(opt.max_error));
    }
};

// F = K_inv * (E * K_inv), K_inv = diag(1, 1, f), as the estimator and ransac_shared_focal_relpose associate it
// (relative_pose.cc:165-168, :179-182, ransac.cc:194-198)
M3 shared_focal_fundamental(const ImagePair &p) {
    const M3 E = essential_from_motion(p.pose);
    M3 F = E;
    for (int i = 0; i < 3; ++i)
        F.m[i][2] = F.m[i][2] * p.focal;
    for (int j = 0; j < 3; ++j)
        F.m[2][j] = p.focal * F.m[2][j];
    return F;
}

// estimators/relative_pose.{h:148-175, cc:154-203}.  The solver is the oracle's own (solvers_focal.cc, polynomial eigenvalue
// problem) - the reference's solutions where its template is accurate, in the reference's order
struct SharedFocalRelEstimator {
    const RelativePoseOptions &opt;
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    Sampler sampler;
    size_t sample_sz = 6, num_data;
    SharedFocalRelEstimator(const RelativePoseOptions &o, const std::vector<V2> &a, const std::vector<V2> &b)
        : opt(o), x1(a), x2(b), sampler(a.size(), 6, o.ransac), num_data(a.size()) {}
    void generate(std::vector<ImagePair> *models) {
        uint64_t s[6];
        sampler.next(s);
        V3 a[6], b[6];
        for (int k = 0; k < 6; ++k) {
            a[k] = bearing(x1[s[k]]);
            b[k] = bearing(x2[s[k]]);
        }
        Pose sol[60];
        double focals[60];
        const int n = relpose_6pt_shared_focal(a, b, sol, focals);
        models->clear();
        for (int i = 0; i < n; ++i) {
            ImagePair m;
            m.pose = sol[i];
            m.focal = focals[i];
            models->push_back(m);
        }
    }
    double score(const ImagePair &p, uint64_t *cnt) const { // relative_pose.cc:164-171
        return msac_sampson_F(shared_focal_fundamental(p), x1, x2, opt.max_error * opt.max_error, cnt);
    }
    void refine(ImagePair *p) const { // relative_pose.cc:173-203
        std::vector<char> mask;
        const int n = inliers_sampson_F(shared_focal_fundamental(*p), x1, x2, 5 * (opt.max_error * opt.max_error), &mask);
        if (n <= 6)
            return;
        std::vector<V2> a, b;
        a.reserve(n);
        b.reserve(n);
        for (size_t k = 0; k < x1.size(); ++k)
            if (mask[k]) {
                a.push_back(x1[k]);
                b.push_back(x2[k]);
            }
        refine_shared_focal_relpose(a, b, p, lo_options(opt.max_error));
    }
};

struct FundEstimator {
    const RelativePoseOptions &opt;
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    Sampler sampler;
    size_t sample_sz = 7, num_data;
    FundEstimator(const RelativePoseOptions &o, const std::vector<V2> &a, const std::vector<V2> &b)
        : opt(o), x1(a), x2(b), sampler(a.size(), 7, o.ransac), num_data(a.size()) {}
    void generate(std::vector<M3> *models) {
        uint64_t s[7];
        sampler.next(s);
        V3 a[7], b[7];
        for (int k = 0; k < 7; ++k) {
            a[k] = bearing(x1[s[k]]);
            b[k] = bearing(x2[s[k]]);
        }
        M3 sol[3];
        const int n = relpose_7pt(a, b, sol);
        models->assign(sol, sol + n);
        if (opt.real_focal_check) // relative_pose.cc:393-398
            for (int i = static_cast<int>(models->size()) - 1; i >= 0; --i)
                if (!real_focal_check((*models)[i]))
                    models->erase(models->begin() + i);
    }
    double score(const M3 &F, uint64_t *cnt) const {
        return msac_sampson_F(F, x1, x2, opt.max_error * opt.max_error, cnt);
    }
    void refine(M3 *F) const { refine_fundamental(x1, x2, F, lo_options(opt.max_error)); }
};

struct HomEstimator {
    const HomographyOptions &opt;
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    Sampler sampler;
    size_t sample_sz = 4, num_data;
    HomEstimator(const HomographyOptions &o, const std::vector<V2> &a, const std::vector<V2> &b)
        : opt(o), x1(a), x2(b), sampler(a.size(), 4, o.ransac), num_data(a.size()) {}
    void generate(std::vector<M3> *models) {
        uint64_t s[4];
        sampler.next(s);
        V3 a[4], b[4];
        for (int k = 0; k < 4; ++k) {
            a[k] = bearing(x1[s[k]]);
            b[k] = bearing(x2[s[k]]);
        }
        models->clear();
        M3 H;
        if (homography_4pt(a, b, &H, true) > 0)
            models->push_back(H);
    }
    double score(const M3 &H, uint64_t *cnt) const {
        return msac_homography(H, x1, x2, opt.max_error * opt.max_error, cnt);
    }
    void refine(M3 *H) const { refine_homography(x1, x2, H, lo_options(opt.max_error)); }
};

void reset_pose(Pose *p) {
    p->q = V4();
    p->q[0] = 1.0;
    p->t = V3{0, 0, 0};
}

} // namespace

RansacStats ransac_pnp(const std::vector<V2> &x, const std::vector<V3> &X, const AbsolutePoseOptions &opt, Pose *best,
                       std::vector<char> *inliers, LoopTrace *trace) {
    if (!opt.ransac.score_initial_model)
        reset_pose(best);
    AbsEstimator est(opt, x, X);
    const RansacStats st = lo_ransac(est, opt.ransac, best, trace);
    inliers_reproj(*best, x, X, opt.max_error * opt.max_error, inliers);
    return st;
}
RansacStats ransac_pnpf(const std::vector<V2> &x, const std::vector<V3> &X, const AbsolutePoseOptions &opt, Image *best,
                        std::vector<char> *inliers, LoopTrace *trace) { // ransac.cc:58-75
    reset_pose(&best->pose);
    best->camera.model_id = CAM_SIMPLE_PINHOLE;
    best->camera.width = best->camera.height = 0;
    best->camera.params = {1.0, 0.0, 0.0};
    FocalAbsEstimator est(opt, x, X);
    const RansacStats st = lo_ransac(est, opt.ransac, best, trace);
    inliers_reproj_image(*best, x, X, opt.max_error * opt.max_error, inliers);
    return st;
}
RansacStats ransac_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                           Pose *best, std::vector<char> *inliers, LoopTrace *trace) {
    if (!opt.ransac.score_initial_model)
        reset_pose(best);
    RelEstimator est(opt, x1, x2);
    const RansacStats st = lo_ransac(est, opt.ransac, best, trace);
    inliers_sampson_pose(*best, x1, x2, opt.max_error * opt.max_error, inliers);
    return st;
}
RansacStats ransac_shared_focal_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                                        ImagePair *best, std::vector<char> *inliers, LoopTrace *trace) { // ransac.cc:182-203
    if (!opt.ransac.score_initial_model) {
        reset_pose(&best->pose);
        best->focal = 1.0;
    }
    SharedFocalRelEstimator est(opt, x1, x2);
    const RansacStats st = lo_ransac(est, opt.ransac, best, trace);
    inliers_sampson_F(shared_focal_fundamental(*best), x1, x2, opt.max_error * opt.max_error, inliers);
    return st;
}
RansacStats ransac_fundamental(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                               M3 *best, std::vector<char> *inliers, LoopTrace *trace) {
    if (!opt.ransac.score_initial_model)
        *best = M3::identity();
    FundEstimator est(opt, x1, x2);
    const RansacStats st = lo_ransac(est, opt.ransac, best, trace);
    inliers_sampson_F(*best, x1, x2, opt.max_error * opt.max_error, inliers);
    return st;
}
RansacStats ransac_homography(const std::vector<V2> &x1, const std::vector<V2> &x2, const HomographyOptions &opt,
                              M3 *best, std::vector<char> *inliers, LoopTrace *trace) {
    if (!opt.ransac.score_initial_model)
        *best = M3::identity();
    HomEstimator est(opt, x1, x2);
    const RansacStats st = lo_ransac(est, opt.ransac, best, trace);
    inliers_homography(*best, x1, x2, opt.max_error * opt.max_error, inliers);
    return st;
}

// ------------------------------------------------------------------------------------ front-ends
RansacStats estimate_absolute_pose(const std::vector<V2> &p2d, const std::vector<V3> &p3d, AbsolutePoseOptions opt,
                                   Image *image, std::vector<char> *inliers) { // robust.cc:36-126
    AbsolutePoseOptions scaled = opt;
    std::vector<V2> norm_pts(p2d.size());
    for (size_t k = 0; k < p2d.size(); ++k)
        norm_pts[k] = image->camera.unproject(p2d[k]);
    double scale = 1.0 / image->camera.focal();
    scaled.max_error *= scale;

    RansacStats st;
    if (opt.estimate_focal_length) { // robust.cc:47-54
        Image img;
        st = ransac_pnpf(norm_pts, p3d, scaled, &img, inliers);
        image->pose = img.pose;
        image->camera.set_focal(img.camera.focal() / scale);
        scaled.bundle.refine_focal_length = true;
    } else {
        st = ransac_pnp(norm_pts, p3d, scaled, &image->pose, inliers);
    }

    if (st.num_inliers > 3) {
        std::vector<V2> xin;
        std::vector<V3> Xin;
        xin.reserve(p2d.size());
        Xin.reserve(p3d.size());
        scale = 1.0 / image->camera.focal();
        scaled.bundle.loss_scale = opt.bundle.loss_scale * scale;
        for (size_t k = 0; k < p2d.size(); ++k) {
            if (!(*inliers)[k])
                continue;
            xin.push_back(p2d[k] * scale);
            Xin.push_back(p3d[k]);
        }
        image->camera.rescale(scale);
        bundle_adjust(xin, Xin, image, scaled.bundle);
        image->camera.rescale(1.0 / scale);
    }
    return st;
}

RansacStats estimate_relative_pose(const std::vector<V2> &x1, const std::vector<V2> &x2, const Camera &cam1,
                                   const Camera &cam2, const RelativePoseOptions &opt, Pose *pose,
                                   std::vector<char> *inliers) { // robust.cc:242-314 (non-tangent branch)
    const size_t n = x1.size();
    const double scale = 0.5 * (1.0 / cam1.focal() + 1.0 / cam2.focal());
    RelativePoseOptions scaled = opt;
    scaled.max_error *= scale;
    scaled.bundle.loss_scale *= scale;

    std::vector<V2> a(n), b(n);
    for (size_t k = 0; k < n; ++k) {
        a[k] = cam1.unproject(x1[k]);
        b[k] = cam2.unproject(x2[k]);
    }
    const RansacStats st = ransac_relpose(a, b, scaled, pose, inliers);
    if (st.num_inliers > 5) {
        std::vector<V2> ai, bi;
        ai.reserve(st.num_inliers);
        bi.reserve(st.num_inliers);
        for (size_t k = 0; k < n; ++k)
            if ((*inliers)[k]) {
                ai.push_back(a[k]);
                bi.push_back(b[k]);
            }
        refine_relpose(ai, bi, pose, scaled.bundle);
    }
    return st;
}

RansacStats estimate_shared_focal_relative_pose(const std::vector<V2> &x1, const std::vector<V2> &x2, const V2 &pp,
                                                const RelativePoseOptions &opt, ImagePair *pair,
                                                std::vector<char> *inliers) { // robust.cc:366-424
    const size_t n = x1.size();
    M3 T1, T2;
    std::vector<V2> a = x1, b = x2;
    for (size_t k = 0; k < n; ++k) {
        a[k] = a[k] - pp;
        b[k] = b[k] - pp;
    }
    const double scale = normalize_points(a, b, T1, T2, true, false, true);
    RelativePoseOptions scaled = opt;
    scaled.max_error /= scale;
    scaled.bundle.loss_scale /= scale;
    if (opt.ransac.score_initial_model)
        pair->focal = pair->focal / scale;
    const RansacStats st = ransac_shared_focal_relpose(a, b, scaled, pair, inliers);
    if (st.num_inliers > 6) {
        std::vector<V2> ai, bi;
        ai.reserve(st.num_inliers);
        bi.reserve(st.num_inliers);
        for (size_t k = 0; k < n; ++k)
            if ((*inliers)[k]) {
                ai.push_back(a[k]);
                bi.push_back(b[k]);
            }
        refine_shared_focal_relpose(ai, bi, pair, scaled.bundle);
    }
    pair->focal *= scale;
    return st;
}

static void scale_to_unit_frobenius(M3 &A) {
    const double n = frob(A);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A.m[i][j] /= n;
}

RansacStats estimate_fundamental(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                                 M3 *F, std::vector<char> *inliers) { // robust.cc:544-594
    const size_t n = x1.size();
    if (n < 7)
        return RansacStats();
    M3 T1, T2;
    std::vector<V2> a = x1, b = x2;
    const double scale = normalize_points(a, b, T1, T2, true, !opt.real_focal_check, true);
    RelativePoseOptions scaled = opt;
    scaled.max_error /= scale;
    scaled.bundle.loss_scale /= scale;
    if (opt.ransac.score_initial_model) {
        *F = inverse(transpose(T2)) * (*F) * inverse(T1);
        scale_to_unit_frobenius(*F);
    }
    const RansacStats st = ransac_fundamental(a, b, scaled, F, inliers);
    if (st.num_inliers > 7) {
        std::vector<V2> ai, bi;
        ai.reserve(st.num_inliers);
        bi.reserve(st.num_inliers);
        for (size_t k = 0; k < n; ++k)
            if ((*inliers)[k]) {
                ai.push_back(a[k]);
                bi.push_back(b[k]);
            }
        refine_fundamental(ai, bi, F, scaled.bundle);
    }
    *F = transpose(T2) * (*F) * T1;
    scale_to_unit_frobenius(*F);
    return st;
}

RansacStats estimate_homography(const std::vector<V2> &x1, const std::vector<V2> &x2, const HomographyOptions &opt,
                                M3 *H, std::vector<char> *inliers) { // robust.cc:712-757
    const size_t n = x1.size();
    if (n < 4)
        return RansacStats();
    M3 T1, T2;
    std::vector<V2> a = x1, b = x2;
    const double scale = normalize_points(a, b, T1, T2, true, true, true);
    HomographyOptions scaled = opt;
    scaled.max_error /= scale;
    scaled.bundle.loss_scale /= scale;
    if (opt.ransac.score_initial_model) {
        *H = T2 * (*H) * inverse(T1);
        scale_to_unit_frobenius(*H);
    }
    const RansacStats st = ransac_homography(a, b, scaled, H, inliers);
    if (st.num_inliers > 4) {
        std::vector<V2> ai, bi;
        ai.reserve(st.num_inliers);
        bi.reserve(st.num_inliers);
        for (size_t k = 0; k < n; ++k)
            if ((*inliers)[k]) {
                ai.push_back(a[k]);
                bi.push_back(b[k]);
            }
        refine_homography(ai, bi, H, scaled.bundle);
    }
    *H = inverse(T2) * (*H) * T1;
    scale_to_unit_frobenius(*H);
    return st;
}

} // namespace orc
