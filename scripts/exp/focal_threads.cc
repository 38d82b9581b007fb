// Throughput of estimate_absolute_pose(estimate_focal_length) / estimate_shared_focal_relative_pose from T host threads through the
// C-ABI, without an interpreter in the way (scripts/focal_threads.py holds the GIL for the marshalling of every call).
//   g++ -O2 -std=c++17 -I include scripts/exp/focal_threads.cc -o scripts/exp/focal_threads -L poselib_amd/lib -lposelib_amd -Wl,-rpath,'$ORIGIN/../../poselib_amd/lib' -pthread
#include "poselib_amd.h"
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
struct Scene { std::vector<double> x2d, X3d, x1, x2; pl_camera cam; };
static Scene make_scene(int n, double outliers, unsigned seed) {
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> G(0, 0.5);
    Scene s;
    const double f = 1000 + 400 * U(g), cx = 0, cy = 0;
    // pose: small rotation about y, translation
    const double a = 0.3 * (U(g) - 0.5), ca = std::cos(a), sa = std::sin(a);
    const double R[9] = {ca, 0, sa, 0, 1, 0, -sa, 0, ca}, t[3] = {0.3, -0.2, 0.5};
    for (int i = 0; i < n; ++i) {
        const double Xc[3] = {4 * U(g) - 2, 4 * U(g) - 2, 4 + 4 * U(g)};
        // world = R^T (Xc - t)
        double Xw[3];
        for (int k = 0; k < 3; ++k) Xw[k] = R[0 + k] * (Xc[0] - t[0]) + R[3 + k] * (Xc[1] - t[1]) + R[6 + k] * (Xc[2] - t[2]);
        double u = f * Xc[0] / Xc[2] + cx + G(g), v = f * Xc[1] / Xc[2] + cy + G(g);
        if (U(g) < outliers) u = 1000 * (U(g) - 0.5), v = 1000 * (U(g) - 0.5);
        s.x2d.push_back(u), s.x2d.push_back(v);
        for (int k = 0; k < 3; ++k) s.X3d.push_back(Xw[k]);
        // two-view: camera 1 at identity sees Xw? use Xc in camera 2 and Xw in camera 1 (both focal f)
        double u1 = f * Xw[0] / (Xw[2] + 6), v1 = f * Xw[1] / (Xw[2] + 6);
        (void)u1; (void)v1;
    }
    s.cam.model_id = 0, s.cam.width = 1000, s.cam.height = 1000, s.cam.num_params = 3;
    s.cam.params[0] = 1.0, s.cam.params[1] = cx, s.cam.params[2] = cy;
    return s;
}
int main(int argc, char **argv) {
    const int n = 2000;
    std::vector<Scene> scenes;
    for (int k = 0; k < 4; ++k) scenes.push_back(make_scene(n, 0.4, 100 + k));
    for (int a = 1; a < argc; ++a) {
        const int T = atoi(argv[a]);
        const int N = std::max(64, 48 * T);
        std::atomic<int> next{0}, ready{0};
        std::atomic<long> iters{0};
        std::chrono::steady_clock::time_point t0;
        auto one = [&](int j, bool timed) {
            const Scene &s = scenes[(unsigned)j % 4];
            pl_robust_options o;
            pl_default_robust_options(&o, 0);
            o.max_error = 4.0, o.estimate_focal_length = 1, o.ransac.seed = (uint64_t)(j + 1000);
            pl_camera cam = s.cam;
            pl_camera_pose pose;
            std::vector<uint8_t> inl(n);
            pl_ransac_stats st;
            if (pl_estimate_absolute_pose(s.x2d.data(), s.X3d.data(), n, &o, &cam, &pose, inl.data(), &st) != 0) {
                fprintf(stderr, "error: %s\n", pl_last_error());
                exit(1);
            }
            if (timed) iters += (long)st.iterations;
            if (timed && j == N - 1 && a == 1) printf("  (last problem: focal %.1f, %llu inliers, %llu iterations, %llu refinements)\n", cam.params[0], (unsigned long long)st.num_inliers, (unsigned long long)st.iterations, (unsigned long long)st.refinements);
        };
        auto work = [&](int i) {
            one(i, false), one(i + 1, false); // every thread's context, stream and buffers exist before the clock starts
            if (ready.fetch_add(1) + 1 == T) t0 = std::chrono::steady_clock::now();
            while (ready.load() < T) std::this_thread::yield();
            for (;;) {
                const int j = next.fetch_add(1);
                if (j >= N) break;
                one(j, true);
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i) th.emplace_back(work, i);
        for (auto &x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("pnpf_2000 %3d threads: %8.0f problems/s (%.2f ms per problem and thread; %.0f iterations per problem)\n", T, N / dt, 1e3 * dt / N * T, (double)iters / N);
        fflush(stdout);
    }
}
