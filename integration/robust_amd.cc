// Reference-side binding of poselib_amd: the file a PoseLib maintainer would add to the PoseLib tree (e.g. as
// PoseLib/robust_amd.cc, compiled INSTEAD of robust.cc / robust/ransac.cc for the four estimators below).  It
// defines the reference's own entry points (PoseLib/robust.h:45-46, 68-70, 84-90, 112-113, 133-134; robust/ransac.h:39-40,
// 52-54, 60-61, 71-73, 85-87, 99-101; solvers/p3p.h:42, relpose_5pt.h:40) by forwarding to the C-ABI of include/poselib_amd.h.
// std::vector<Eigen::Vector2d/3d> is contiguous AoS doubles - exactly the layout the C-ABI takes - and
// Eigen::Matrix3d::data() is column-major like the double[9] arguments, so only options and the camera are copied.
//
// tests/test_integration_shim.py compiles this file against the reference's headers (where /root/reference exists)
// and links it with libposelib_amd.so, so the binding shown in INTEGRATION.md is known to build.
#include <PoseLib/robust.h>
#include <PoseLib/robust/ransac.h>
#include <PoseLib/solvers/p35pf.h>
#include <PoseLib/solvers/p3p.h>
#include <PoseLib/solvers/relpose_5pt.h>

#include <poselib_amd.h>

#include <algorithm>
#include <stdexcept>

namespace poselib {
namespace {

pl_robust_options to_pl(int kind, const RansacOptions &r, const BundleOptions &b, double max_error) {
    pl_robust_options o;
    pl_default_robust_options(&o, kind);
    o.ransac.max_iterations = r.max_iterations;
    o.ransac.min_iterations = r.min_iterations;
    o.ransac.dyn_num_trials_mult = r.dyn_num_trials_mult;
    o.ransac.success_prob = r.success_prob;
    o.ransac.seed = r.seed;
    o.ransac.score_initial_model = r.score_initial_model;
    o.ransac.progressive_sampling = r.progressive_sampling;
    o.ransac.max_prosac_iterations = r.max_prosac_iterations;
    o.bundle.max_iterations = b.max_iterations;
    o.bundle.loss_type = static_cast<int32_t>(b.loss_type);
    o.bundle.loss_scale = b.loss_scale;
    o.bundle.gradient_tol = b.gradient_tol;
    o.bundle.step_tol = b.step_tol;
    o.bundle.relative_cost_tol = b.relative_cost_tol;
    o.bundle.initial_lambda = b.initial_lambda;
    o.bundle.min_lambda = b.min_lambda;
    o.bundle.max_lambda = b.max_lambda;
    o.bundle.lambda_factor = b.lambda_factor;
    o.bundle.lambda_update = static_cast<int32_t>(b.lambda_update);
    o.bundle.damping = static_cast<int32_t>(b.damping);
    o.bundle.refine_focal_length = b.refine_focal_length;       // (absolute pose: the final bundle moves the camera,
    o.bundle.refine_extra_params = b.refine_extra_params;       //  robust.cc:103-123; ignored by the two-view refiners)
    o.bundle.refine_principal_point = b.refine_principal_point;
    o.max_error = max_error;
    return o;
}

RansacStats from_pl(const pl_ransac_stats &s) {
    RansacStats r;
    r.refinements = s.refinements;
    r.iterations = s.iterations;
    r.num_inliers = s.num_inliers;
    r.inlier_ratio = s.inlier_ratio;
    r.model_score = s.model_score;
    return r;
}

pl_camera to_pl(const Camera &c) {
    // pl_camera carries at most 12 parameters (the supported models need <= 8); the reference has models with more
    // (RadTanThinPrismFisheye: 16) - reject them here instead of overrunning the array
    if (c.params.size() > sizeof(pl_camera::params) / sizeof(double))
        throw std::runtime_error("poselib_amd: camera model not supported (NULL, SIMPLE_PINHOLE, PINHOLE, OPENCV)");
    pl_camera cam{c.model_id, c.width, c.height, static_cast<int32_t>(c.params.size()), {}};
    std::copy(c.params.begin(), c.params.end(), cam.params);
    return cam;
}

// The SIMPLE_PINHOLE camera {f, cx, cy} the focal-length estimators return (fields set directly: nothing of camera_models.cc is
// needed to link this file)
void set_simple_pinhole(Camera *c, double f, double cx, double cy) {
    c->model_id = SimplePinholeCameraModel::model_id;
    c->width = c->height = -1;
    c->params = {f, cx, cy};
}
// Camera::focal() (misc/camera_models.cc:304-321: the mean of the parameters at the model's focal_idx) without linking the camera
// models: models with ONE focal length (focal_idx = {0}: SIMPLE_PINHOLE, SIMPLE_RADIAL, RADIAL, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE,
// SIMPLE_DIVISION) return params[0], models without one (1D_RADIAL, SPHERICAL) 1, all others fx and fy (focal_idx = {0, 1}).  ADVICE r3:
// the round-3 form averaged f and cx for the single-focal models other than SIMPLE_PINHOLE.
double focal_of(const Camera &c) {
    if (c.params.empty())
        return 1.0;
    const int id = c.model_id;
    if (id == SimplePinholeCameraModel::model_id || id == SimpleRadialCameraModel::model_id || id == RadialCameraModel::model_id ||
        id == SimpleRadialFisheyeCameraModel::model_id || id == RadialFisheyeCameraModel::model_id || id == SimpleDivisionCameraModel::model_id)
        return c.params[0];
    if (id == Radial1DCameraModel::model_id || id == SphericalCameraModel::model_id)
        return 1.0;
    double f = 0.0; // (the reference's association: 0 + p0 / 2 + p1 / 2)
    f += c.params.at(0) / 2;
    f += c.params.at(1) / 2;
    return f;
}

pl_camera_pose to_pl(const CameraPose &p) {
    pl_camera_pose q;
    for (int i = 0; i < 4; ++i)
        q.q[i] = p.q(i);
    for (int i = 0; i < 3; ++i)
        q.t[i] = p.t(i);
    return q;
}

void from_pl(const pl_camera_pose &q, CameraPose *p) {
    for (int i = 0; i < 4; ++i)
        p->q(i) = q.q[i];
    for (int i = 0; i < 3; ++i)
        p->t(i) = q.t[i];
}

void check(int rc) {
    if (rc != PL_OK)
        throw std::runtime_error(pl_last_error());
}

uint8_t *mask_of(std::vector<char> *inliers, size_t n) {
    inliers->resize(n); // the reference's get_inliers() resizes the caller's vector (utils.cc:376)
    return reinterpret_cast<uint8_t *>(inliers->data());
}

const double *raw(const std::vector<Point2D> &v) { return v.empty() ? nullptr : v[0].data(); }
const double *raw(const std::vector<Point3D> &v) { return v.empty() ? nullptr : v[0].data(); }

} // namespace

// ---- robust.h front-ends -------------------------------------------------------------------------------------
RansacStats estimate_absolute_pose(const std::vector<Point2D> &points2D, const std::vector<Point3D> &points3D,
                                   AbsolutePoseOptions opt, Image *image, std::vector<char> *inliers) {
    pl_robust_options o = to_pl(0, opt.ransac, opt.bundle, opt.max_error);
    o.estimate_focal_length = opt.estimate_focal_length; // robust.cc:47-54: ransac_pnpf on the device since round 3
    o.estimate_extra_params = opt.estimate_extra_params;
    o.min_fov = opt.min_fov; // types.h:126, absolute_pose.h:78
    pl_camera cam = to_pl(image->camera);
    pl_camera_pose pose = to_pl(image->pose);
    pl_ransac_stats st;
    check(pl_estimate_absolute_pose(raw(points2D), raw(points3D), points2D.size(), &o, &cam, &pose,
                                    mask_of(inliers, points2D.size()), &st));
    from_pl(pose, &image->pose);
    std::copy(cam.params, cam.params + cam.num_params, image->camera.params.begin());
    return from_pl(st);
}

// ---- not in the reference: many independent absolute-pose problems in ONE call, spread over the GPUs of the node --------------
// (a PoseLib user loops over estimate_absolute_pose() from a thread pool; with the device library the loop is the library's:
// pl_estimate_batch_devices round-robins the problems over `devices` - empty: every visible GPU - from this one process, one worker
// pool per device, and every problem's result equals its single call bit for bit.)  images: in (camera, and the pose when
// ransac.score_initial_model is set) and out, one per problem.
std::vector<RansacStats> estimate_absolute_pose_batch(const std::vector<std::vector<Point2D>> &points2D,
                                                      const std::vector<std::vector<Point3D>> &points3D,
                                                      const std::vector<AbsolutePoseOptions> &opts, std::vector<Image> *images,
                                                      std::vector<std::vector<char>> *inliers, const std::vector<int> &devices = {}) {
    const size_t count = points2D.size();
    if (points3D.size() != count || opts.size() != count || images->size() != count)
        throw std::runtime_error("estimate_absolute_pose_batch: one entry per problem in every argument");
    inliers->resize(count);
    std::vector<pl_robust_options> o(count);
    std::vector<pl_camera> cams(count);
    std::vector<pl_camera_pose> poses(count);
    std::vector<pl_ransac_stats> st(count);
    std::vector<pl_batch_item> items(count);
    for (size_t i = 0; i < count; ++i) {
        o[i] = to_pl(0, opts[i].ransac, opts[i].bundle, opts[i].max_error);
        o[i].estimate_focal_length = opts[i].estimate_focal_length;
        o[i].estimate_extra_params = opts[i].estimate_extra_params;
        o[i].min_fov = opts[i].min_fov;
        cams[i] = to_pl((*images)[i].camera);
        poses[i] = to_pl((*images)[i].pose);
        pl_batch_item &it = items[i];
        it.kind = 0;
        it.status = 0;
        it.a = raw(points2D[i]);
        it.b = raw(points3D[i]);
        it.n = points2D[i].size();
        it.opt = &o[i];
        it.camera1 = &cams[i];
        it.camera2 = nullptr;
        it.model = &poses[i];
        it.inliers = mask_of(&(*inliers)[i], points2D[i].size());
        it.stats = &st[i];
    }
    check(pl_estimate_batch_devices(items.data(), count, devices.empty() ? nullptr : devices.data(), (int)devices.size(), 0));
    std::vector<RansacStats> out(count);
    for (size_t i = 0; i < count; ++i) {
        from_pl(poses[i], &(*images)[i].pose);
        std::copy(cams[i].params, cams[i].params + cams[i].num_params, (*images)[i].camera.params.begin());
        out[i] = from_pl(st[i]);
    }
    return out;
}

RansacStats estimate_relative_pose(const std::vector<Point2D> &points2D_1, const std::vector<Point2D> &points2D_2,
                                   const Camera &camera1, const Camera &camera2, const RelativePoseOptions &opt,
                                   CameraPose *relative_pose, std::vector<char> *inliers) {
    pl_robust_options o = to_pl(1, opt.ransac, opt.bundle, opt.max_error);
    o.tangent_sampson = opt.tangent_sampson; // -> PL_ERR_UNSUPPORTED
    const pl_camera c1 = to_pl(camera1), c2 = to_pl(camera2);
    pl_camera_pose pose = to_pl(*relative_pose);
    pl_ransac_stats st;
    check(pl_estimate_relative_pose(raw(points2D_1), raw(points2D_2), points2D_1.size(), &c1, &c2, &o, &pose,
                                    mask_of(inliers, points2D_1.size()), &st));
    from_pl(pose, relative_pose);
    return from_pl(st);
}

// robust.h:84-90
RansacStats estimate_shared_focal_relative_pose(const std::vector<Point2D> &points2D_1, const std::vector<Point2D> &points2D_2,
                                                const Point2D &pp, const RelativePoseOptions &opt, ImagePair *image_pair,
                                                std::vector<char> *inliers) {
    pl_robust_options o = to_pl(1, opt.ransac, opt.bundle, opt.max_error);
    o.tangent_sampson = opt.tangent_sampson; // -> PL_ERR_UNSUPPORTED
    pl_camera_pose pose = to_pl(image_pair->pose);
    double focal = focal_of(image_pair->camera1); // read with ransac.score_initial_model only (robust.cc:392-397)
    const double pp2[2] = {pp(0), pp(1)};
    pl_ransac_stats st;
    check(pl_estimate_shared_focal_relative_pose(raw(points2D_1), raw(points2D_2), points2D_1.size(), pp2, &o, &pose, &focal,
                                                 mask_of(inliers, points2D_1.size()), &st));
    from_pl(pose, &image_pair->pose);
    set_simple_pinhole(&image_pair->camera1, focal, pp(0), pp(1)); // :417-421
    image_pair->camera2 = image_pair->camera1;
    return from_pl(st);
}

RansacStats estimate_fundamental(const std::vector<Point2D> &points2D_1, const std::vector<Point2D> &points2D_2,
                                 const RelativePoseOptions &opt, Eigen::Matrix3d *F, std::vector<char> *inliers) {
    pl_robust_options o = to_pl(2, opt.ransac, opt.bundle, opt.max_error);
    o.real_focal_check = opt.real_focal_check;
    pl_ransac_stats st;
    check(pl_estimate_fundamental(raw(points2D_1), raw(points2D_2), points2D_1.size(), &o, F->data(),
                                  mask_of(inliers, points2D_1.size()), &st));
    return from_pl(st);
}

RansacStats estimate_homography(const std::vector<Point2D> &points2D_1, const std::vector<Point2D> &points2D_2,
                                const HomographyOptions &opt, Eigen::Matrix3d *H, std::vector<char> *inliers) {
    pl_robust_options o = to_pl(3, opt.ransac, opt.bundle, opt.max_error);
    pl_ransac_stats st;
    check(pl_estimate_homography(raw(points2D_1), raw(points2D_2), points2D_1.size(), &o, H->data(),
                                 mask_of(inliers, points2D_1.size()), &st));
    return from_pl(st);
}

// ---- robust/ransac.h entry points (normalised image points) ------------------------------------------------
RansacStats ransac_pnp(const std::vector<Point2D> &x, const std::vector<Point3D> &X, const AbsolutePoseOptions &opt,
                       CameraPose *best_model, std::vector<char> *best_inliers) {
    const pl_robust_options o = to_pl(0, opt.ransac, opt.bundle, opt.max_error);
    pl_camera_pose pose = to_pl(*best_model);
    pl_ransac_stats st;
    check(pl_ransac_pnp(raw(x), raw(X), x.size(), &o, &pose, mask_of(best_inliers, x.size()), &st));
    from_pl(pose, best_model);
    return from_pl(st);
}

// ransac.h:52-54
RansacStats ransac_pnpf(const std::vector<Point2D> &x, const std::vector<Point3D> &X, const AbsolutePoseOptions &opt, Image *best_model,
                        std::vector<char> *best_inliers) {
    pl_robust_options o = to_pl(0, opt.ransac, opt.bundle, opt.max_error);
    o.min_fov = opt.min_fov; // absolute_pose.h:78: FocalAbsolutePoseEstimator bounds f by compute_max_focal_length(opt.min_fov)
    pl_camera_pose pose = to_pl(best_model->pose);
    double focal = 1.0;
    pl_ransac_stats st;
    check(pl_ransac_pnpf(raw(x), raw(X), x.size(), &o, &pose, &focal, mask_of(best_inliers, x.size()), &st));
    from_pl(pose, &best_model->pose);
    set_simple_pinhole(&best_model->camera, focal, 0.0, 0.0);
    return from_pl(st);
}

// ransac.h:71-73
RansacStats ransac_shared_focal_relpose(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2, const RelativePoseOptions &opt,
                                        ImagePair *best_model, std::vector<char> *best_inliers) {
    const pl_robust_options o = to_pl(1, opt.ransac, opt.bundle, opt.max_error);
    pl_camera_pose pose = to_pl(best_model->pose);
    double focal = focal_of(best_model->camera1);
    pl_ransac_stats st;
    check(pl_ransac_shared_focal_relpose(raw(x1), raw(x2), x1.size(), &o, &pose, &focal, mask_of(best_inliers, x1.size()), &st));
    from_pl(pose, &best_model->pose);
    set_simple_pinhole(&best_model->camera1, focal, 0.0, 0.0);
    best_model->camera2 = best_model->camera1;
    return from_pl(st);
}

RansacStats ransac_relpose(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2,
                           const RelativePoseOptions &opt, CameraPose *best_model, std::vector<char> *best_inliers) {
    const pl_robust_options o = to_pl(1, opt.ransac, opt.bundle, opt.max_error);
    pl_camera_pose pose = to_pl(*best_model);
    pl_ransac_stats st;
    check(pl_ransac_relpose(raw(x1), raw(x2), x1.size(), &o, &pose, mask_of(best_inliers, x1.size()), &st));
    from_pl(pose, best_model);
    return from_pl(st);
}

RansacStats ransac_fundamental(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2,
                               const RelativePoseOptions &opt, Eigen::Matrix3d *best_model,
                               std::vector<char> *best_inliers) {
    pl_robust_options o = to_pl(2, opt.ransac, opt.bundle, opt.max_error);
    o.real_focal_check = opt.real_focal_check;
    pl_ransac_stats st;
    check(pl_ransac_fundamental(raw(x1), raw(x2), x1.size(), &o, best_model->data(),
                                mask_of(best_inliers, x1.size()), &st));
    return from_pl(st);
}

RansacStats ransac_homography(const std::vector<Point2D> &x1, const std::vector<Point2D> &x2,
                              const HomographyOptions &opt, Eigen::Matrix3d *best_model,
                              std::vector<char> *best_inliers) {
    const pl_robust_options o = to_pl(3, opt.ransac, opt.bundle, opt.max_error);
    pl_ransac_stats st;
    check(pl_ransac_homography(raw(x1), raw(x2), x1.size(), &o, best_model->data(),
                               mask_of(best_inliers, x1.size()), &st));
    return from_pl(st);
}

// ---- bare minimal solvers (unit bearing vectors in) ---------------------------------------------------------
int p3p(const std::vector<Eigen::Vector3d> &x, const std::vector<Eigen::Vector3d> &X, std::vector<CameraPose> *output) {
    pl_camera_pose sols[4];
    const int n = pl_p3p(x[0].data(), X[0].data(), sols);
    if (n < 0)
        throw std::runtime_error(pl_last_error());
    output->assign(n, CameraPose());
    for (int i = 0; i < n; ++i)
        from_pl(sols[i], &(*output)[i]);
    return n;
}

// solvers/p35pf.h:39-54 (normalize_input only changes the scaling inside the reference's solver; this one always normalises)
int p35pf(const std::vector<Eigen::Vector2d> &points2d, const std::vector<Eigen::Vector3d> &points3d,
          std::vector<CameraPose> *output_poses, std::vector<double> *output_focals, bool /*normalize_input*/) {
    pl_camera_pose sols[10];
    double focals[10];
    const int n = pl_p35pf(points2d[0].data(), points3d[0].data(), sols, focals);
    if (n < 0)
        throw std::runtime_error(pl_last_error());
    output_poses->assign(n, CameraPose());
    output_focals->assign(focals, focals + n);
    for (int i = 0; i < n; ++i)
        from_pl(sols[i], &(*output_poses)[i]);
    return n;
}

int relpose_5pt(const std::vector<Eigen::Vector3d> &x1, const std::vector<Eigen::Vector3d> &x2,
                std::vector<CameraPose> *output) {
    pl_camera_pose sols[40];
    const int n = pl_relpose_5pt(x1[0].data(), x2[0].data(), sols);
    if (n < 0)
        throw std::runtime_error(pl_last_error());
    output->assign(n, CameraPose());
    for (int i = 0; i < n; ++i)
        from_pl(sols[i], &(*output)[i]);
    return n;
}

} // namespace poselib
