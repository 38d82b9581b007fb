// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// The four hot estimators, their ransac_* entry points and the estimate_* front-ends.
//   estimators : PoseLib/robust/estimators/absolute_pose.{h:39-66,cc:46-69}, relative_pose.{h:40-67,cc:48-86},
//                relative_pose.{h:309-336,cc:384-412} (fundamental), homography.{h:38-65,cc:36-61}
//   entry      : PoseLib/robust/ransac.cc:44-57, :142-154, :248-262, :300-314
//   front-ends : PoseLib/robust.cc:36-126, :242-314, :544-594, :712-757
//   options    : PoseLib/types.h:108-175
#pragma once
#include "ransac_core.h"
#include "refine.h"
#include "vecmath.h"

#include <vector>

namespace orc {

struct AbsolutePoseOptions { // types.h:108-126 (estimate_extra_params - the radial-distortion branch - is out of scope)
    RansacOptions ransac;
    BundleOptions bundle;
    double max_error = 12.0;
    bool estimate_focal_length = false;
    double min_fov = 5.0; // degrees; bounds the focal length the focal estimator accepts (types.h:126)
};
struct RelativePoseOptions { // types.h:128-145
    RansacOptions ransac;
    BundleOptions bundle;
    double max_error = 1.0;
    bool tangent_sampson = false; // not supported by the oracle (out of scope)
    bool real_focal_check = false;
};
struct HomographyOptions { // types.h:170-175
    RansacOptions ransac;
    BundleOptions bundle;
    double max_error = 1.0;
};

RansacStats ransac_pnp(const std::vector<V2> &x, const std::vector<V3> &X, const AbsolutePoseOptions &opt, Pose *best,
                       std::vector<char> *inliers, LoopTrace *trace = nullptr);
// robust/ransac.cc:58-75 with FocalAbsolutePoseEstimator (estimators/absolute_pose.{h:69-113,cc:71-177}, solver P3.5Pf): pose and
// focal length of a SIMPLE_PINHOLE camera whose principal point is the origin
RansacStats ransac_pnpf(const std::vector<V2> &x, const std::vector<V3> &X, const AbsolutePoseOptions &opt, Image *best,
                        std::vector<char> *inliers, LoopTrace *trace = nullptr);
RansacStats ransac_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                           Pose *best, std::vector<char> *inliers, LoopTrace *trace = nullptr);
// robust/ransac.cc:182-203 with SharedFocalRelativePoseEstimator (estimators/relative_pose.{h:148-175,cc:154-203}): relative pose and
// the focal length shared by both views; points relative to the principal point
RansacStats ransac_shared_focal_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                                        ImagePair *best, std::vector<char> *inliers, LoopTrace *trace = nullptr);
RansacStats ransac_fundamental(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                               M3 *best, std::vector<char> *inliers, LoopTrace *trace = nullptr);
RansacStats ransac_homography(const std::vector<V2> &x1, const std::vector<V2> &x2, const HomographyOptions &opt,
                              M3 *best, std::vector<char> *inliers, LoopTrace *trace = nullptr);

RansacStats estimate_absolute_pose(const std::vector<V2> &p2d, const std::vector<V3> &p3d, AbsolutePoseOptions opt,
                                   Image *image, std::vector<char> *inliers);
RansacStats estimate_relative_pose(const std::vector<V2> &x1, const std::vector<V2> &x2, const Camera &cam1,
                                   const Camera &cam2, const RelativePoseOptions &opt, Pose *pose,
                                   std::vector<char> *inliers);
RansacStats estimate_shared_focal_relative_pose(const std::vector<V2> &x1, const std::vector<V2> &x2, const V2 &pp,
                                                const RelativePoseOptions &opt, ImagePair *pair, std::vector<char> *inliers);
RansacStats estimate_fundamental(const std::vector<V2> &x1, const std::vector<V2> &x2, const RelativePoseOptions &opt,
                                 M3 *F, std::vector<char> *inliers);
RansacStats estimate_homography(const std::vector<V2> &x1, const std::vector<V2> &x2, const HomographyOptions &opt,
                                M3 *H, std::vector<char> *inliers);

} // namespace orc
