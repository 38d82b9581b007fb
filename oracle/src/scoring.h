// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// MSAC scoring, inlier masks and point normalisation restated from PoseLib/robust/utils.cc.
// The reference spells these out as scalar expressions; the association order below is the
// reference's, so given bit-identical models the per-point residuals are bit-identical.
#pragma once
#include "vecmath.h"

#include <cstdint>
#include <vector>

namespace orc {

// utils.cc:36-65
double msac_reproj(const Pose &pose, const std::vector<V2> &x, const std::vector<V3> &X, double sq_thr,
                   uint64_t *inliers);
// utils.cc:158-201  (Sampson + cheirality at min depth 0.01)
double msac_sampson_pose(const Pose &pose, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                         uint64_t *inliers);
// utils.cc:204-239
double msac_sampson_F(const M3 &F, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                      uint64_t *inliers);
// utils.cc:300-329
double msac_homography(const M3 &H, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                       uint64_t *inliers);

// utils.cc:374-384
void inliers_reproj(const Pose &pose, const std::vector<V2> &x, const std::vector<V3> &X, double sq_thr,
                    std::vector<char> *mask);
// utils.cc:434-476
int inliers_sampson_pose(const Pose &pose, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                         std::vector<char> *mask);
// utils.cc:479-513
int inliers_sampson_F(const M3 &F, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                      std::vector<char> *mask);
// utils.cc:331-351
void inliers_homography(const M3 &H, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                        std::vector<char> *mask);

// utils.cc:584-644
double normalize_points(std::vector<V2> &x1, std::vector<V2> &x2, M3 &T1, M3 &T2, bool normalize_scale,
                        bool normalize_centroid, bool shared_scale);
// utils.cc:646-671 (evaluated in float, as the reference does)
bool real_focal_check(const M3 &F);

} // namespace orc
