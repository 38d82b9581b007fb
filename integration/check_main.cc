// Executes the reference-side binding (integration/robust_amd.cc): a PoseLib user's program - the reference's own
// types (std::vector<Eigen::Vector2d>, Camera, Image, CameraPose, AbsolutePoseOptions, ...) and the reference's own
// entry-point signatures (PoseLib/robust.h:45-46, 68-70, 112-113, 133-134) - linked against robust_amd.o INSTEAD of
// the reference's robust.cc, so every poselib::estimate_* call below lands in libposelib_amd.so.
//
//   robust_amd_check <in.bin> <out.bin>
//     (kind 4: estimate_shared_focal_relative_pose, kind 5: estimate_absolute_pose with estimate_focal_length,
//      kind 6: ransac_pnpf on points relative to the principal point, AbsolutePoseOptions::min_fov = params[11],
//      kind 7: estimate_absolute_pose_batch - the binding's multi-device call, not in the reference: six problems (the scene
//              with seeds seed .. seed + 5) over the device list {0, 0}; every one must equal its single call, the first is
//              written out)
//     in : doubles [kind, n, seed, max_error, model_id, num_params, params[12], A (n x 2), B (n x 3 | n x 2)]
//     out: doubles [iterations, refinements, num_inliers, model_score, model (7: q t | 9: column-major 3x3),
//                   camera params[12] (kind 0), inliers (n)]
// tests/test_integration_shim.py (-m gpu) feeds it the scenes of the parity tests and compares the output with the
// ctypes path bit for bit.  Built by integration/Makefile against the reference's headers (oracle/eigen_shim stands in
// for Eigen, which this image lacks) where /root/reference exists; the binary travels to the GPU box.
#include <PoseLib/robust.h>
#include <PoseLib/robust/ransac.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace poselib;

namespace poselib { // (integration/robust_amd.cc; the reference's headers have no batched entry point)
std::vector<RansacStats> estimate_absolute_pose_batch(const std::vector<std::vector<Point2D>> &points2D,
                                                      const std::vector<std::vector<Point3D>> &points3D,
                                                      const std::vector<AbsolutePoseOptions> &opts, std::vector<Image> *images,
                                                      std::vector<std::vector<char>> *inliers, const std::vector<int> &devices);
}

int main(int argc, char **argv) {
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 2;
    }
    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f)
        return 2;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<double> in(bytes / sizeof(double));
    if (std::fread(in.data(), sizeof(double), in.size(), f) != in.size())
        return 2;
    std::fclose(f);

    const bool estimate_focal = (int)in[0] == 5;
    const bool pnpf = (int)in[0] == 6;
    const bool multi = (int)in[0] == 7;
    const int kind = (estimate_focal || pnpf || multi) ? 0 : (int)in[0];
    const size_t n = (size_t)in[1];
    RansacOptions ransac;
    ransac.seed = (size_t)in[2];
    const double max_error = in[3];
    const double min_fov = in[17]; // (kind 6 only)
    std::vector<double> cam_params(in.begin() + 6, in.begin() + 6 + (size_t)in[5]);
    const Camera camera((int)in[4], cam_params);
    const double *a = in.data() + 18, *b = a + 2 * n;
    std::vector<Point2D> x1(n), x2d(n);
    std::vector<Point3D> X(n);
    for (size_t i = 0; i < n; ++i) {
        x1[i] = Point2D(a[2 * i], a[2 * i + 1]);
        if (kind == 0)
            X[i] = Point3D(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
        else
            x2d[i] = Point2D(b[2 * i], b[2 * i + 1]);
    }

    std::vector<char> inliers;
    RansacStats st;
    std::vector<double> model;
    std::vector<double> cam_out(12, 0.0);
    if (multi) {
        const size_t K = 6;
        std::vector<std::vector<Point2D>> xs(K, x1);
        std::vector<std::vector<Point3D>> Xs(K, X);
        std::vector<AbsolutePoseOptions> opts(K);
        std::vector<Image> images(K);
        std::vector<std::vector<char>> masks;
        for (size_t j = 0; j < K; ++j) {
            opts[j].ransac = ransac;
            opts[j].ransac.seed = ransac.seed + j;
            opts[j].max_error = max_error;
            images[j].camera = camera;
        }
        const std::vector<RansacStats> stats = estimate_absolute_pose_batch(xs, Xs, opts, &images, &masks, {0, 0});
        for (size_t j = 0; j < K; ++j) { // every problem against its single call: bit for bit
            Image single;
            single.camera = camera;
            std::vector<char> m1;
            const RansacStats s1 = estimate_absolute_pose(x1, X, opts[j], &single, &m1);
            bool same = s1.iterations == stats[j].iterations && s1.refinements == stats[j].refinements && s1.num_inliers == stats[j].num_inliers &&
                        s1.model_score == stats[j].model_score && m1 == masks[j] && single.camera.params == images[j].camera.params;
            for (int i = 0; i < 4; ++i)
                same = same && single.pose.q(i) == images[j].pose.q(i);
            for (int i = 0; i < 3; ++i)
                same = same && single.pose.t(i) == images[j].pose.t(i);
            if (!same) {
                std::fprintf(stderr, "estimate_absolute_pose_batch: problem %zu differs from its single call\n", j);
                return 3;
            }
        }
        st = stats[0];
        inliers = masks[0];
        for (int i = 0; i < 4; ++i)
            model.push_back(images[0].pose.q(i));
        for (int i = 0; i < 3; ++i)
            model.push_back(images[0].pose.t(i));
        for (size_t i = 0; i < images[0].camera.params.size() && i < 12; ++i)
            cam_out[i] = images[0].camera.params[i];
    } else if (pnpf) { // robust/ransac.h:52-54 with a non-default field-of-view bound (absolute_pose.h:78)
        AbsolutePoseOptions opt;
        opt.ransac = ransac;
        opt.max_error = max_error;
        opt.min_fov = min_fov;
        Image image;
        st = ransac_pnpf(x1, X, opt, &image, &inliers);
        for (int i = 0; i < 4; ++i)
            model.push_back(image.pose.q(i));
        for (int i = 0; i < 3; ++i)
            model.push_back(image.pose.t(i));
        for (size_t i = 0; i < image.camera.params.size() && i < 12; ++i)
            cam_out[i] = image.camera.params[i];
    } else if (kind == 0) {
        AbsolutePoseOptions opt;
        opt.ransac = ransac;
        opt.max_error = max_error;
        opt.estimate_focal_length = estimate_focal; // (kind 5: robust.cc:47-54)
        Image image;
        image.camera = camera;
        st = estimate_absolute_pose(x1, X, opt, &image, &inliers);
        for (int i = 0; i < 4; ++i)
            model.push_back(image.pose.q(i));
        for (int i = 0; i < 3; ++i)
            model.push_back(image.pose.t(i));
        for (size_t i = 0; i < image.camera.params.size() && i < 12; ++i)
            cam_out[i] = image.camera.params[i];
    } else if (kind == 1) {
        RelativePoseOptions opt;
        opt.ransac = ransac;
        opt.max_error = max_error;
        CameraPose pose;
        st = estimate_relative_pose(x1, x2d, camera, camera, opt, &pose, &inliers);
        for (int i = 0; i < 4; ++i)
            model.push_back(pose.q(i));
        for (int i = 0; i < 3; ++i)
            model.push_back(pose.t(i));
    } else if (kind == 4) { // two views sharing one unknown focal length (robust.h:84-90); principal point = camera params 1, 2
        RelativePoseOptions opt;
        opt.ransac = ransac;
        opt.max_error = max_error;
        ImagePair pair;
        st = estimate_shared_focal_relative_pose(x1, x2d, Point2D(cam_params[1], cam_params[2]), opt, &pair, &inliers);
        for (int i = 0; i < 4; ++i)
            model.push_back(pair.pose.q(i));
        for (int i = 0; i < 3; ++i)
            model.push_back(pair.pose.t(i));
        for (size_t i = 0; i < pair.camera1.params.size() && i < 12; ++i)
            cam_out[i] = pair.camera1.params[i];
    } else if (kind == 2) {
        RelativePoseOptions opt;
        opt.ransac = ransac;
        opt.max_error = max_error;
        Eigen::Matrix3d F;
        st = estimate_fundamental(x1, x2d, opt, &F, &inliers);
        model.assign(F.data(), F.data() + 9);
    } else {
        HomographyOptions opt;
        opt.ransac = ransac;
        opt.max_error = max_error;
        Eigen::Matrix3d H;
        st = estimate_homography(x1, x2d, opt, &H, &inliers);
        model.assign(H.data(), H.data() + 9);
    }

    std::vector<double> out = {(double)st.iterations, (double)st.refinements, (double)st.num_inliers, st.model_score};
    out.insert(out.end(), model.begin(), model.end());
    out.insert(out.end(), cam_out.begin(), cam_out.end());
    for (size_t i = 0; i < n; ++i)
        out.push_back(i < inliers.size() && inliers[i] ? 1.0 : 0.0);
    f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(out.data(), sizeof(double), out.size(), f) != out.size())
        return 2;
    std::fclose(f);
    return 0;
}
