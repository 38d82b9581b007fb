// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Citations: scoring.h.
#include "scoring.h"

#include "solvers.h"

#include <cmath>

namespace orc {

double msac_reproj(const Pose &pose, const std::vector<V2> &x, const std::vector<V3> &X, double sq_thr,
                   uint64_t *inliers) {
    const M3 R = pose.R();
    const double r00 = R.m[0][0], r01 = R.m[0][1], r02 = R.m[0][2], t0 = pose.t.x;
    const double r10 = R.m[1][0], r11 = R.m[1][1], r12 = R.m[1][2], t1 = pose.t.y;
    const double r20 = R.m[2][0], r21 = R.m[2][1], r22 = R.m[2][2], t2 = pose.t.z;
    uint64_t cnt = 0;
    double score = 0.0;
    for (size_t k = 0; k < x.size(); ++k) {
        const double X0 = X[k].x, X1 = X[k].y, X2 = X[k].z;
        const double z0 = r00 * X0 + r01 * X1 + r02 * X2 + t0;
        const double z1 = r10 * X0 + r11 * X1 + r12 * X2 + t1;
        const double z2 = r20 * X0 + r21 * X1 + r22 * X2 + t2;
        if (z2 <= 0.0)
            continue; // behind the camera: pays the threshold through the (N - cnt) term below
        const double inv = 1.0 / z2;
        const double e0 = z0 * inv - x[k].x;
        const double e1 = z1 * inv - x[k].y;
        const double r2 = e0 * e0 + e1 * e1;
        if (r2 < sq_thr) {
            ++cnt;
            score += r2;
        }
    }
    score += (x.size() - cnt) * sq_thr;
    *inliers = cnt;
    return score;
}

namespace {
// Sampson r^2 for one correspondence under a 3x3 E/F given as nine scalars (utils.cc:174-185).
struct Epi {
    double e00, e01, e02, e10, e11, e12, e20, e21, e22;
    explicit Epi(const M3 &E)
        : e00(E.m[0][0]), e01(E.m[0][1]), e02(E.m[0][2]), e10(E.m[1][0]), e11(E.m[1][1]), e12(E.m[1][2]),
          e20(E.m[2][0]), e21(E.m[2][1]), e22(E.m[2][2]) {}
    double sampson(const V2 &a, const V2 &b) const {
        const double Ea0 = e00 * a.x + e01 * a.y + e02;
        const double Ea1 = e10 * a.x + e11 * a.y + e12;
        const double Ea2 = e20 * a.x + e21 * a.y + e22;
        const double Eb0 = e00 * b.x + e10 * b.y + e20;
        const double Eb1 = e01 * b.x + e11 * b.y + e21;
        const double C = b.x * Ea0 + b.y * Ea1 + Ea2;
        const double Cx = Ea0 * Ea0 + Ea1 * Ea1;
        const double Cy = Eb0 * Eb0 + Eb1 * Eb1;
        return C * C / (Cx + Cy);
    }
};
} // namespace

double msac_sampson_pose(const Pose &pose, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                         uint64_t *inliers) {
    const Epi E(essential_from_motion(pose));
    uint64_t cnt = 0;
    double score = 0.0;
    for (size_t k = 0; k < x1.size(); ++k) {
        const double r2 = E.sampson(x1[k], x2[k]);
        if (r2 < sq_thr && check_cheirality(pose, bearing(x1[k]), bearing(x2[k]), 0.01)) {
            ++cnt;
            score += r2;
        } else {
            score += sq_thr;
        }
    }
    *inliers = cnt;
    return score;
}

double msac_sampson_F(const M3 &F, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                      uint64_t *inliers) {
    const Epi E(F);
    uint64_t cnt = 0;
    double score = 0.0;
    for (size_t k = 0; k < x1.size(); ++k) {
        const double r2 = E.sampson(x1[k], x2[k]);
        if (r2 < sq_thr) {
            ++cnt;
            score += r2;
        } else {
            score += sq_thr;
        }
    }
    *inliers = cnt;
    return score;
}

static inline double transfer_sq_err(const M3 &H, const V2 &a, const V2 &b) {
    const double h0 = H.m[0][0] * a.x + H.m[0][1] * a.y + H.m[0][2];
    const double h1 = H.m[1][0] * a.x + H.m[1][1] * a.y + H.m[1][2];
    const double inv = 1.0 / (H.m[2][0] * a.x + H.m[2][1] * a.y + H.m[2][2]);
    const double e0 = h0 * inv - b.x;
    const double e1 = h1 * inv - b.y;
    return e0 * e0 + e1 * e1;
}

double msac_homography(const M3 &H, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                       uint64_t *inliers) {
    uint64_t cnt = 0;
    double score = 0;
    for (size_t k = 0; k < x1.size(); ++k) {
        const double r2 = transfer_sq_err(H, x1[k], x2[k]);
        if (r2 < sq_thr) {
            ++cnt;
            score += r2;
        } else {
            score += sq_thr;
        }
    }
    *inliers = cnt;
    return score;
}

void inliers_reproj(const Pose &pose, const std::vector<V2> &x, const std::vector<V3> &X, double sq_thr,
                    std::vector<char> *mask) {
    mask->resize(x.size());
    const M3 R = pose.R();
    for (size_t k = 0; k < x.size(); ++k) {
        const V3 Z = R * X[k] + pose.t;
        const double e0 = Z.x / Z.z - x[k].x;
        const double e1 = Z.y / Z.z - x[k].y;
        const double r2 = e0 * e0 + e1 * e1;
        (*mask)[k] = (r2 < sq_thr && Z.z > 0.0);
    }
}

int inliers_sampson_pose(const Pose &pose, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                         std::vector<char> *mask) {
    mask->resize(x1.size());
    const Epi E(essential_from_motion(pose));
    int cnt = 0;
    for (size_t k = 0; k < x1.size(); ++k) {
        const double r2 = E.sampson(x1[k], x2[k]);
        bool in = r2 < sq_thr;
        if (in) {
            if (check_cheirality(pose, bearing(x1[k]), bearing(x2[k]), 0.01))
                ++cnt;
            else
                in = false;
        }
        (*mask)[k] = in;
    }
    return cnt;
}

int inliers_sampson_F(const M3 &F, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                      std::vector<char> *mask) {
    mask->resize(x1.size());
    const Epi E(F);
    int cnt = 0;
    for (size_t k = 0; k < x1.size(); ++k) {
        const bool in = E.sampson(x1[k], x2[k]) < sq_thr;
        cnt += in;
        (*mask)[k] = in;
    }
    return cnt;
}

void inliers_homography(const M3 &H, const std::vector<V2> &x1, const std::vector<V2> &x2, double sq_thr,
                        std::vector<char> *mask) {
    mask->resize(x1.size());
    for (size_t k = 0; k < x1.size(); ++k)
        (*mask)[k] = transfer_sq_err(H, x1[k], x2[k]) < sq_thr;
}

double normalize_points(std::vector<V2> &x1, std::vector<V2> &x2, M3 &T1, M3 &T2, bool normalize_scale,
                        bool normalize_centroid, bool shared_scale) {
    T1 = M3::identity();
    T2 = M3::identity();
    const size_t n = x1.size();
    if (normalize_centroid) {
        V2 c1, c2;
        for (size_t k = 0; k < n; ++k) {
            c1 = c1 + x1[k];
            c2 = c2 + x2[k];
        }
        c1 = c1 / static_cast<double>(n);
        c2 = c2 / static_cast<double>(x2.size());
        T1.m[0][2] = -c1.x;
        T1.m[1][2] = -c1.y;
        T2.m[0][2] = -c2.x;
        T2.m[1][2] = -c2.y;
        for (size_t k = 0; k < n; ++k) {
            x1[k] = x1[k] - c1;
            x2[k] = x2[k] - c2;
        }
    }
    auto scale_rows = [](M3 &T, double f) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                T.m[i][j] *= f;
    };
    if (normalize_scale && shared_scale) {
        double scale = 0.0;
        for (size_t k = 0; k < n; ++k) {
            scale += norm(x1[k]);
            scale += norm(x2[k]);
        }
        scale /= std::sqrt(2) * n;
        for (size_t k = 0; k < n; ++k) {
            x1[k] = x1[k] / scale;
            x2[k] = x2[k] / scale;
        }
        scale_rows(T1, 1.0 / scale);
        scale_rows(T2, 1.0 / scale);
        return scale;
    } else if (normalize_scale && !shared_scale) {
        double s1 = 0.0, s2 = 0.0;
        for (size_t k = 0; k < n; ++k) {
            s1 += norm(x1[k]);
            s2 += norm(x2[k]);
        }
        s1 /= n / std::sqrt(2);
        s2 /= x2.size() / std::sqrt(2);
        for (size_t k = 0; k < n; ++k) {
            x1[k] = x1[k] / s1;
            x2[k] = x2[k] / s2;
        }
        scale_rows(T1, 1.0 / s1);
        scale_rows(T2, 1.0 / s2);
        return std::sqrt(s1 * s2);
    }
    return 1.0;
}

bool real_focal_check(const M3 &Fm) {
    auto F = [&](int i, int j) { return Fm.m[i][j]; };
    float den, num;
    den = F(0, 0) * F(0, 1) * F(2, 0) * F(2, 2) - F(0, 0) * F(0, 2) * F(2, 0) * F(2, 1) +
          F(0, 1) * F(0, 1) * F(2, 1) * F(2, 2) - F(0, 1) * F(0, 2) * F(2, 1) * F(2, 1) +
          F(1, 0) * F(1, 1) * F(2, 0) * F(2, 2) - F(1, 0) * F(1, 2) * F(2, 0) * F(2, 1) +
          F(1, 1) * F(1, 1) * F(2, 1) * F(2, 2) - F(1, 1) * F(1, 2) * F(2, 1) * F(2, 1);
    num = -F(2, 2) * (F(0, 1) * F(0, 2) * F(2, 2) - F(0, 2) * F(0, 2) * F(2, 1) + F(1, 1) * F(1, 2) * F(2, 2) -
                      F(1, 2) * F(1, 2) * F(2, 1));
    if (num * den < 0)
        return false;
    den = F(0, 0) * F(1, 0) * F(0, 2) * F(2, 2) - F(0, 0) * F(2, 0) * F(0, 2) * F(1, 2) +
          F(1, 0) * F(1, 0) * F(1, 2) * F(2, 2) - F(1, 0) * F(2, 0) * F(1, 2) * F(1, 2) +
          F(0, 1) * F(1, 1) * F(0, 2) * F(2, 2) - F(0, 1) * F(2, 1) * F(0, 2) * F(1, 2) +
          F(1, 1) * F(1, 1) * F(1, 2) * F(2, 2) - F(1, 1) * F(2, 1) * F(1, 2) * F(1, 2);
    num = -F(2, 2) * (F(1, 0) * F(2, 0) * F(2, 2) - F(2, 0) * F(2, 0) * F(1, 2) + F(1, 1) * F(2, 1) * F(2, 2) -
                      F(2, 1) * F(2, 1) * F(1, 2));
    if (num * den < 0)
        return false;
    return true;
}

} // namespace orc
