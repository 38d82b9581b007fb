// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Levenberg-Marquardt refinement used for local optimisation (LO) and the final polish.
//   options / stats      : PoseLib/types.h:60-106
//   robust losses        : PoseLib/robust/robust_loss.h:59-157, robust_loss.cc:32-55
//   normal equations     : PoseLib/robust/optim/jacobian_accumulator.h:46-166
//   LM driver            : PoseLib/robust/optim/lm_impl.h:56-140
//   camera (subset)      : PoseLib/misc/camera_models.cc:304-323 (focal), :432-455 (rescale),
//                          :668-705 (PINHOLE), :716-755 (SIMPLE_PINHOLE), :919-1032 (OPENCV),
//                          :2704-2726 (NULL), camera_models.h:98-102 (2-D unproject wrapper)
//   refiners             : optim/absolute.h:40-171, optim/relative.h:39-166, optim/homography.h:46-178,
//                          optim/fundamental.h:41-121, optim/optim_utils.h:57-82
//   entry points         : PoseLib/robust/bundle.cc:84-112, :207-222, :314-335, :395-411
#pragma once
#include "vecmath.h"

#include <cstdint>
#include <vector>

namespace orc {

enum LossType { LOSS_TRIVIAL = 0, LOSS_TRUNCATED, LOSS_HUBER, LOSS_CAUCHY, LOSS_TRUNCATED_CAUCHY, LOSS_TRUNCATED_LE_ZACH };

struct BundleOptions { // types.h:60-95
    uint64_t max_iterations = 100;
    int loss_type = LOSS_CAUCHY;
    double loss_scale = 1.0;
    double gradient_tol = 1e-12;
    double step_tol = 1e-8;
    double relative_cost_tol = 1e-10;
    double initial_lambda = 1e-3;
    double min_lambda = 1e-10;
    double max_lambda = 1e10;
    bool verbose = false;
    int lambda_update = 0; // 0 NIELSEN, 1 FIXED_FACTOR
    double lambda_factor = 10.0;
    int damping = 0; // 0 LEVENBERG, 1 MARQUARDT
    bool refine_focal_length = false;
    bool refine_extra_params = false;
    bool refine_principal_point = false;
};

struct BundleStats { // types.h:97-106
    uint64_t iterations = 0;
    double initial_cost = 0, cost = 0, lambda = 0, nu = 2.0;
    uint64_t invalid_steps = 0;
    double step_norm = 0, grad_norm = 0;
};

enum CameraModelId { CAM_NULL = -1, CAM_SIMPLE_PINHOLE = 0, CAM_PINHOLE = 1, CAM_OPENCV = 4 };

struct Camera {
    int model_id = CAM_NULL;
    int width = 0, height = 0;
    std::vector<double> params;
    double focal() const;
    void set_focal(double f); // camera_models.cc:96-107
    void rescale(double s);
    // unit bearing from pixel
    V3 unproject3(const V2 &xp) const;
    // camera_models.h:98-102 : pixel -> normalised image plane point
    V2 unproject(const V2 &xp) const;
    // projection with d(xp)/d(Z) (2x3, row-major) and, when Jp is given, d(xp)/d(params) (2 x num_params)
    V2 project_with_jac(const V3 &Z, double J[2][3], double (*Jp)[12] = nullptr) const;
    // camera_models.cc:545-560: the parameters a bundle adjustment refines (focal, principal point, extra - in this order)
    std::vector<size_t> refinement_idx(bool focal, bool principal_point, bool extra) const;
    V2 project(const V3 &Z) const;
};

struct Image {
    Pose pose;
    Camera camera;
};

BundleStats bundle_adjust(const std::vector<V2> &x, const std::vector<V3> &X, Image *image, const BundleOptions &opt);
BundleStats bundle_adjust(const std::vector<V2> &x, const std::vector<V3> &X, Pose *pose, const BundleOptions &opt);
BundleStats refine_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, Pose *pose, const BundleOptions &opt);
// two views of one SIMPLE_PINHOLE camera with the principal point at the origin (types.h ImagePair with camera1 == camera2)
struct ImagePair {
    Pose pose;
    double focal = 1.0;
};
BundleStats refine_shared_focal_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, ImagePair *pair,
                                        const BundleOptions &opt);
BundleStats refine_homography(const std::vector<V2> &x1, const std::vector<V2> &x2, M3 *H, const BundleOptions &opt);
BundleStats refine_fundamental(const std::vector<V2> &x1, const std::vector<V2> &x2, M3 *F, const BundleOptions &opt);

// 3x3 SVD: Eigen's two-sided Jacobi in Eigen's operation order (eigen_shim/Eigen/src/JacobiSVD3x3.h: the sign of a refined F depends on it); singular values descending, A = U diag(s) V^T.
void svd3(const M3 &A, M3 &U, double s[3], M3 &V);

} // namespace orc
