// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Citations: refine.h.
#include "refine.h"

#include "solvers.h"

#include "../eigen_shim/Eigen/src/JacobiSVD3x3.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

namespace orc {

// ------------------------------------------------------------------------------------ camera
void Camera::set_focal(double f) { // camera_models.cc:96-107 over focal_idx of the model
    switch (model_id) {
    case CAM_SIMPLE_PINHOLE:
        params.at(0) = f;
        break;
    case CAM_PINHOLE:
    case CAM_OPENCV:
        params.at(0) = f;
        params.at(1) = f;
        break;
    default:
        break;
    }
}
double Camera::focal() const { // camera_models.cc:304-323
    if (params.empty())
        return 1.0;
    switch (model_id) {
    case CAM_SIMPLE_PINHOLE:
        return 0.0 + params.at(0) / 1;
    case CAM_PINHOLE:
    case CAM_OPENCV:
        return 0.0 + params.at(0) / 2 + params.at(1) / 2;
    default:
        return 1.0;
    }
}
void Camera::rescale(double s) { // camera_models.cc:432-455
    if (params.empty())
        return;
    switch (model_id) {
    case CAM_SIMPLE_PINHOLE:
        params.at(0) *= s;
        params.at(1) *= s;
        params.at(2) *= s;
        break;
    case CAM_PINHOLE:
    case CAM_OPENCV:
        for (int i = 0; i < 4; ++i)
            params.at(i) *= s;
        break;
    default:
        break;
    }
}

namespace {
// camera_models.cc:932-963 (value + 2x2 Jacobian of the OpenCV distortion)
void opencv_distort(double k1, double k2, double p1, double p2, double u, double v, double &du, double &dv,
                    double J[2][2]) {
    const double u2 = u * u, uv = u * v, v2 = v * v;
    const double r2 = u * u + v * v;
    J[0][0] = k2 * r2 * r2 + 6 * p2 * u + 2 * p1 * v + u * (2 * k1 * u + 4 * k2 * u * r2) + k1 * r2 + 1.0;
    J[0][1] = 2 * p1 * u + 2 * p2 * v + v * (2 * k1 * u + 4 * k2 * u * r2);
    J[1][0] = 2 * p1 * u + 2 * p2 * v + u * (2 * k1 * v + 4 * k2 * v * r2);
    J[1][1] = k2 * r2 * r2 + 2 * p2 * u + 6 * p1 * v + v * (2 * k1 * v + 4 * k2 * v * r2) + k1 * r2 + 1.0;
    const double alpha = 1.0 + k1 * r2 + k2 * r2 * r2;
    du = alpha * u + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
    dv = alpha * v + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
}
// camera_models.cc:972-990 (damped Newton, <= 100 iterations, tolerance 1e-10)
V2 opencv_undistort(double k1, double k2, double p1, double p2, V2 xp) {
    V2 x = xp;
    for (int it = 0; it < 100; ++it) {
        double du, dv, J[2][2];
        opencv_distort(k1, k2, p1, p2, x.x, x.y, du, dv, J);
        J[0][0] += 1e-8;
        J[1][1] += 1e-8;
        const double r0 = du - xp.x, r1 = dv - xp.y;
        if (std::sqrt(r0 * r0 + r1 * r1) < 1e-10)
            break;
        const double dt = J[0][0] * J[1][1] - J[1][0] * J[0][1];
        const double id = 1.0 / dt;
        const double s0 = (J[1][1] * id) * r0 + (-J[0][1] * id) * r1;
        const double s1 = (-J[1][0] * id) * r0 + (J[0][0] * id) * r1;
        x.x = x.x - s0;
        x.y = x.y - s1;
    }
    return x;
}
} // namespace

V3 Camera::unproject3(const V2 &xp) const {
    switch (model_id) {
    case CAM_NULL:
        return V3{xp.x, xp.y, 1.0};
    case CAM_SIMPLE_PINHOLE:
        return normalized(V3{(xp.x - params[1]) / params[0], (xp.y - params[2]) / params[0], 1.0});
    case CAM_PINHOLE:
        return normalized(V3{(xp.x - params[2]) / params[0], (xp.y - params[3]) / params[1], 1.0});
    case CAM_OPENCV: {
        const V2 d{(xp.x - params[2]) / params[0], (xp.y - params[3]) / params[1]};
        const V2 u = opencv_undistort(params[4], params[5], params[6], params[7], d);
        return normalized(V3{u.x, u.y, 1.0});
    }
    default:
        throw std::runtime_error("NYI"); // camera_models.cc:184-186
    }
}
V2 Camera::unproject(const V2 &xp) const {
    const V3 b = unproject3(xp);
    return V2{b.x / b.z, b.y / b.z};
}
V2 Camera::project(const V3 &Z) const {
    switch (model_id) {
    case CAM_NULL:
        return V2{Z.x / Z.z, Z.y / Z.z};
    case CAM_SIMPLE_PINHOLE:
        return V2{params[0] * Z.x / Z.z + params[1], params[0] * Z.y / Z.z + params[2]};
    case CAM_PINHOLE:
        return V2{params[0] * Z.x / Z.z + params[2], params[1] * Z.y / Z.z + params[3]};
    case CAM_OPENCV: {
        double du, dv, J[2][2];
        opencv_distort(params[4], params[5], params[6], params[7], Z.x / Z.z, Z.y / Z.z, du, dv, J);
        return V2{params[0] * du + params[2], params[1] * dv + params[3]};
    }
    default:
        throw std::runtime_error("NYI");
    }
}
std::vector<size_t> Camera::refinement_idx(bool focal, bool principal_point, bool extra) const {
    // focal_idx / principal_point_idx / extra_idx of the models (camera_models.cc:708-710, 759-761, 1036-1038)
    std::vector<size_t> idx;
    auto add = [&](std::initializer_list<size_t> v) { idx.insert(idx.end(), v.begin(), v.end()); };
    switch (model_id) {
    case CAM_SIMPLE_PINHOLE:
        if (focal)
            add({0});
        if (principal_point)
            add({1, 2});
        break;
    case CAM_PINHOLE:
        if (focal)
            add({0, 1});
        if (principal_point)
            add({2, 3});
        break;
    case CAM_OPENCV:
        if (focal)
            add({0, 1});
        if (principal_point)
            add({2, 3});
        if (extra)
            add({4, 5, 6, 7});
        break;
    default: // NULL camera: no parameters
        break;
    }
    return idx;
}
V2 Camera::project_with_jac(const V3 &Z, double J[2][3], double (*Jp)[12]) const {
    switch (model_id) {
    case CAM_NULL: {
        const V2 xp{Z.x / Z.z, Z.y / Z.z};
        const double zi = 1.0 / Z.z;
        J[0][0] = zi, J[0][1] = 0.0, J[0][2] = -xp.x * zi;
        J[1][0] = 0.0, J[1][1] = zi, J[1][2] = -xp.y * zi;
        return xp;
    }
    case CAM_SIMPLE_PINHOLE:
    case CAM_PINHOLE: {
        const bool simple = model_id == CAM_SIMPLE_PINHOLE;
        const double fx = params[0], fy = simple ? params[0] : params[1];
        const double cx = simple ? params[1] : params[2], cy = simple ? params[2] : params[3];
        const double zi = 1.0 / Z.z;
        const double px = fx * Z.x * zi, py = fy * Z.y * zi;
        J[0][0] = fx * zi, J[0][1] = 0.0, J[0][2] = -px * zi;
        J[1][0] = 0.0, J[1][1] = fy * zi, J[1][2] = -py * zi;
        if (Jp) { // camera_models.cc:688-698, 739-747
            if (simple) {
                Jp[0][0] = Z.x * zi, Jp[1][0] = Z.y * zi;
                Jp[0][1] = 1.0, Jp[1][1] = 0.0;
                Jp[0][2] = 0.0, Jp[1][2] = 1.0;
            } else {
                Jp[0][0] = Z.x * zi, Jp[1][0] = 0.0;
                Jp[0][1] = 0.0, Jp[1][1] = Z.y * zi;
                Jp[0][2] = 1.0, Jp[1][2] = 0.0;
                Jp[0][3] = 0.0, Jp[1][3] = 1.0;
            }
        }
        return V2{px + cx, py + cy};
    }
    case CAM_OPENCV: {
        const double u = Z.x / Z.z, v = Z.y / Z.z;
        double du, dv, Jd[2][2];
        opencv_distort(params[4], params[5], params[6], params[7], u, v, du, dv, Jd);
        const double P[2][3] = {{1.0 / Z.z, 0.0, -u / Z.z}, {0.0, 1.0 / Z.z, -v / Z.z}};
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 3; ++b)
                J[a][b] = Jd[a][0] * P[0][b] + Jd[a][1] * P[1][b];
        for (int b = 0; b < 3; ++b) {
            J[0][b] *= params[0];
            J[1][b] *= params[1];
        }
        if (Jp) { // camera_models.cc:953-965 (d distortion / d (k1, k2, p1, p2)), :1005-1021
            const double u2 = u * u, uv = u * v, v2 = v * v, r2 = u * u + v * v;
            const double j1[2][4] = {{r2 * u, r2 * r2 * u, 2.0 * uv, (r2 + 2.0 * u2)}, {r2 * v, r2 * r2 * v, (r2 + 2.0 * v2), 2.0 * uv}};
            Jp[0][0] = du, Jp[1][0] = 0.0;
            Jp[0][1] = 0.0, Jp[1][1] = dv;
            Jp[0][2] = 1.0, Jp[1][2] = 0.0;
            Jp[0][3] = 0.0, Jp[1][3] = 1.0;
            for (int k = 0; k < 4; ++k) {
                Jp[0][4 + k] = params[0] * j1[0][k];
                Jp[1][4 + k] = params[1] * j1[1][k];
            }
        }
        return V2{params[0] * du + params[2], params[1] * dv + params[3]};
    }
    default:
        throw std::runtime_error("NYI");
    }
}

// ------------------------------------------------------------------------------------ losses
namespace {

struct Loss { // robust_loss.h:59-157
    int type;
    double thr, sq, inv_sq, max_loss, mu = 0.5;
    Loss(int t, double scale) : type(t), thr(scale), sq(scale * scale), inv_sq(1.0 / (scale * scale)) {
        max_loss = sq * std::log1p(1.0);
    }
    double loss(double r2) const {
        switch (type) {
        case LOSS_TRUNCATED:
        case LOSS_TRUNCATED_LE_ZACH:
            return std::min(r2, sq);
        case LOSS_HUBER: {
            const double r = std::sqrt(r2);
            return (r <= thr) ? r2 : thr * (2.0 * r - thr);
        }
        case LOSS_CAUCHY:
            return sq * std::log1p(r2 * inv_sq);
        case LOSS_TRUNCATED_CAUCHY:
            return (r2 > sq) ? max_loss : sq * std::log1p(r2 * inv_sq);
        default:
            return r2;
        }
    }
    double weight(double r2) const {
        switch (type) {
        case LOSS_TRUNCATED:
            return (r2 < sq) ? 1.0 : 0.0;
        case LOSS_TRUNCATED_LE_ZACH: {
            const double rh = r2 / sq;
            const double zstar = std::min(rh, 1.0);
            if (rh < 1.0)
                return 0.5;
            const double m1 = rh - 1.0;
            const double rho = (2.0 * m1 + std::sqrt(4.0 * m1 * m1 * mu * mu + 2 * mu * m1)) / mu;
            const double a = (rh + mu * rho * zstar - 0.5 * rho) / (1 + mu * rho);
            const double zbar = std::max(0.0, std::min(a, 1.0));
            return (zstar - zbar) / rho;
        }
        case LOSS_HUBER: {
            const double r = std::sqrt(r2);
            return (r <= thr) ? 1.0 : thr / r;
        }
        case LOSS_CAUCHY:
            return std::max(std::numeric_limits<double>::min(), 1.0 / (1.0 + r2 * inv_sq));
        case LOSS_TRUNCATED_CAUCHY:
            return (r2 > sq) ? 0.0 : std::max(std::numeric_limits<double>::min(), 1.0 / (1.0 + r2 * inv_sq));
        default:
            return 1.0;
        }
    }
};

// ------------------------------------------------------------------------------------ normal equations
constexpr int KMAX = 14; // 6 pose parameters + up to 8 camera parameters (OPENCV)
struct Normal { // jacobian_accumulator.h:46-166.  NB: ONE counter shared by both passes.
    int k;
    const Loss *loss;
    double JtJ[KMAX][KMAX]; // lower triangle used
    double Jtr[KMAX];
    double racc = 0;
    uint64_t count = 0;
    Normal(int k_, const Loss *l) : k(k_), loss(l) { reset_jacobian(); }
    double scale() const { return 1.0 / std::max(1.0, static_cast<double>(count)); }
    void reset_residual() {
        racc = 0;
        count = 0;
    }
    void add_residual(double r) {
        racc += 1.0 * loss->loss(r * r);
        count++;
    }
    void add_residual(double r0, double r1) {
        racc += 1.0 * loss->loss(r0 * r0 + r1 * r1);
        count++;
    }
    double residual() const { return racc * scale(); }
    void reset_jacobian() {
        count = 0;
        for (int i = 0; i < KMAX; ++i) {
            Jtr[i] = 0;
            for (int j = 0; j < KMAX; ++j)
                JtJ[i][j] = 0;
        }
    }
    void add_jacobian(double r0, double r1, const double (*J)[KMAX]) { // J[2][k]
        const double w = 1.0 * loss->weight(r0 * r0 + r1 * r1);
        if (w == 0)
            return;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j <= i; ++j)
                JtJ[i][j] += w * (J[0][i] * J[0][j] + J[1][i] * J[1][j]);
        const double wr0 = w * r0, wr1 = w * r1;
        for (int i = 0; i < k; ++i)
            Jtr[i] += J[0][i] * wr0 + J[1][i] * wr1;
        count++;
    }
    void add_jacobian(double r, const double *J) { // J[k]
        const double w = 1.0 * loss->weight(r * r);
        if (w == 0)
            return;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j <= i; ++j)
                JtJ[i][j] += w * (J[i] * J[j]);
        const double wr = w * r;
        for (int i = 0; i < k; ++i)
            Jtr[i] += wr * J[i];
        count++;
    }
    double grad_norm() const {
        double s = 0;
        for (int i = 0; i < k; ++i)
            s += Jtr[i] * Jtr[i];
        return scale() * std::sqrt(s);
    }
    void solve(double lambda, int damping, double *sol) const {
        const double sc = scale();
        double A[KMAX][KMAX], b[KMAX];
        for (int i = 0; i < k; ++i) {
            for (int j = 0; j <= i; ++j)
                A[i][j] = sc * JtJ[i][j];
            b[i] = -(sc * Jtr[i]);
        }
        for (int i = 0; i < k; ++i)
            A[i][i] += (damping == 1) ? std::max(A[i][i] * lambda, 1e-8) : lambda;
        // Cholesky A = L L^T with the operation order of Eigen's unblocked LLT (Eigen/src/Cholesky/LLT.h,
        // llt_inplace<Lower>::unblocked): per column c, x = A_cc - |A10|^2 (squared norm summed first),
        // A21 = (A21 - A20 * A10^T) / x (dot product summed first); then L y = b by column-oriented updates and
        // L^T x = y by "row dot product, then subtract" (Eigen's triangular_solve_vector, col-/row-major cases).
        for (int c = 0; c < k; ++c) {
            double d = A[c][c];
            if (c > 0) {
                double sq = A[c][0] * A[c][0];
                for (int m = 1; m < c; ++m)
                    sq += A[c][m] * A[c][m];
                d -= sq;
            }
            if (d <= 0)
                break; // not positive definite: Eigen stops factorising and solves with what it has
            d = std::sqrt(d);
            A[c][c] = d;
            for (int r = c + 1; r < k; ++r) {
                double s = A[r][c];
                if (c > 0) {
                    double dot = A[r][0] * A[c][0];
                    for (int m = 1; m < c; ++m)
                        dot += A[r][m] * A[c][m];
                    s -= dot;
                }
                A[r][c] = s / d;
            }
        }
        for (int i = 0; i < k; ++i)
            sol[i] = b[i];
        for (int i = 0; i < k; ++i) {
            sol[i] /= A[i][i];
            for (int r = i + 1; r < k; ++r)
                sol[r] -= sol[i] * A[r][i];
        }
        for (int i = k - 1; i >= 0; --i) {
            if (i + 1 < k) {
                double dot = A[i + 1][i] * sol[i + 1];
                for (int j = i + 2; j < k; ++j)
                    dot += A[j][i] * sol[j];
                sol[i] -= dot;
            }
            sol[i] /= A[i][i];
        }
    }
    double predicted_decrease(const double *step, double lambda) const {
        const double sc = scale();
        double s = 0;
        for (int i = 0; i < k; ++i)
            s += step[i] * (lambda * step[i] + sc * Jtr[i]);
        return -s;
    }
};

// ------------------------------------------------------------------------------------ LM driver
template <typename Problem, typename Model>
BundleStats levenberg_marquardt(Problem &prob, Model *params, const BundleOptions &opt) { // lm_impl.h:56-140
    Loss loss(opt.loss_type, opt.loss_scale);
    Normal acc(prob.k(), &loss);
    BundleStats st;
    acc.reset_residual();
    st.cost = prob.residual(acc, *params);
    st.initial_cost = st.cost;
    st.grad_norm = -1;
    st.step_norm = -1;
    st.invalid_steps = 0;
    st.lambda = opt.initial_lambda;
    st.nu = 2.0;
    bool rejac = true;
    double sol[KMAX];
    for (st.iterations = 0; st.iterations < opt.max_iterations; ++st.iterations) {
        if (rejac) {
            acc.reset_jacobian();
            prob.jacobian(acc, *params);
            st.grad_norm = acc.grad_norm();
            if (st.grad_norm < opt.gradient_tol)
                break;
        }
        acc.solve(st.lambda, opt.damping, sol);
        double sn = 0;
        for (int i = 0; i < prob.k(); ++i)
            sn += sol[i] * sol[i];
        st.step_norm = std::sqrt(sn);
        if (st.step_norm < opt.step_tol)
            break;
        const Model trial = prob.step(sol, *params);
        acc.reset_residual();
        const double cost_new = prob.residual(acc, trial);
        if (cost_new < st.cost) {
            const double decrease = st.cost - cost_new;
            *params = trial;
            st.cost = cost_new;
            rejac = true;
            if (opt.lambda_update == 0) {
                const double pred = acc.predicted_decrease(sol, st.lambda);
                if (pred > 0) {
                    const double rho = decrease / pred;
                    const double factor = 1.0 - std::pow(2.0 * rho - 1.0, 3);
                    st.lambda *= std::max(1.0 / 3.0, factor);
                } else {
                    st.lambda *= 1.0 / 3.0;
                }
                st.nu = 2.0;
            } else {
                st.lambda /= opt.lambda_factor;
            }
            st.lambda = std::max(opt.min_lambda, st.lambda);
            if (st.cost > 0 && decrease / st.cost < opt.relative_cost_tol)
                break;
        } else {
            st.invalid_steps++;
            rejac = false;
            if (opt.lambda_update == 0) {
                st.lambda *= st.nu;
                st.nu *= 2.0;
            } else {
                st.lambda *= opt.lambda_factor;
            }
            st.lambda = std::min(opt.max_lambda, st.lambda);
        }
        if (opt.loss_type == LOSS_TRUNCATED_LE_ZACH)
            loss.mu *= 1.5; // bundle.cc:52-75 iteration callback
    }
    return st;
}

// ------------------------------------------------------------------------------------ refiners
struct AbsProblem { // optim/absolute.h:40-171; cam_idx: the camera parameters refined along with the pose (:130-133, :163-165)
    const std::vector<V2> &x;
    const std::vector<V3> &X;
    std::vector<size_t> cam_idx;
    int k() const { return 6 + (int)cam_idx.size(); }
    double residual(Normal &acc, const Image &im) const {
        const M3 R = im.pose.R();
        for (size_t i = 0; i < x.size(); ++i) {
            const V3 Z = R * X[i] + im.pose.t;
            if (Z.z < 0)
                continue;
            const V2 p = im.camera.project(Z);
            acc.add_residual(p.x - x[i].x, p.y - x[i].y);
        }
        return acc.residual();
    }
    void jacobian(Normal &acc, const Image &im) const {
        const M3 R = im.pose.R();
        for (size_t i = 0; i < x.size(); ++i) {
            const V3 Xi = X[i];
            const V3 Z = R * Xi + im.pose.t;
            if (Z.z < 0)
                continue;
            double Jp[2][3], Jc[2][12];
            const V2 p = im.camera.project_with_jac(Z, Jp, cam_idx.empty() ? nullptr : Jc);
            const double r0 = p.x - x[i].x, r1 = p.y - x[i].y;
            double dZ[2][3];
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 3; ++b)
                    dZ[a][b] = Jp[a][0] * R.m[0][b] + Jp[a][1] * R.m[1][b] + Jp[a][2] * R.m[2][b];
            double J[2][KMAX];
            for (int a = 0; a < 2; ++a) {
                J[a][0] = -Xi.z * dZ[a][1] + Xi.y * dZ[a][2];
                J[a][1] = Xi.z * dZ[a][0] - Xi.x * dZ[a][2];
                J[a][2] = -Xi.y * dZ[a][0] + Xi.x * dZ[a][1];
                J[a][3] = dZ[a][0];
                J[a][4] = dZ[a][1];
                J[a][5] = dZ[a][2];
                for (size_t c = 0; c < cam_idx.size(); ++c)
                    J[a][6 + c] = Jc[a][cam_idx[c]];
            }
            acc.add_jacobian(r0, r1, J);
        }
    }
    Image step(const double *dp, const Image &im) const {
        Image out;
        out.camera = im.camera;
        out.pose.q = quat_step_post(im.pose.q, V3{dp[0], dp[1], dp[2]});
        out.pose.t = im.pose.t + im.pose.rotate(V3{dp[3], dp[4], dp[5]});
        for (size_t c = 0; c < cam_idx.size(); ++c)
            out.camera.params[cam_idx[c]] += dp[6 + c];
        return out;
    }
};

// d r / d vec(E)  (column-major vec) for the Sampson residual; shared by E and F refiners.
// optim/relative.h:121-140 == optim/fundamental.h:84-103
inline double sampson_residual_and_grad(const M3 &E, const V2 &a, const V2 &b, double dF[9]) {
    const double Ea0 = E.m[0][0] * a.x + E.m[0][1] * a.y + E.m[0][2];
    const double Ea1 = E.m[1][0] * a.x + E.m[1][1] * a.y + E.m[1][2];
    const double Ea2 = E.m[2][0] * a.x + E.m[2][1] * a.y + E.m[2][2];
    const double C = b.x * Ea0 + b.y * Ea1 + Ea2;
    double JC[4];
    JC[0] = E.m[0][0] * b.x + E.m[1][0] * b.y + E.m[2][0];
    JC[1] = E.m[0][1] * b.x + E.m[1][1] * b.y + E.m[2][1];
    JC[2] = Ea0;
    JC[3] = Ea1;
    const double nJ = std::sqrt(JC[0] * JC[0] + JC[1] * JC[1] + JC[2] * JC[2] + JC[3] * JC[3]);
    const double inv = 1.0 / nJ;
    const double r = C * inv;
    dF[0] = a.x * b.x, dF[1] = a.x * b.y, dF[2] = a.x;
    dF[3] = a.y * b.x, dF[4] = a.y * b.y, dF[5] = a.y;
    dF[6] = b.x, dF[7] = b.y, dF[8] = 1.0;
    const double s = C * inv * inv;
    dF[0] -= s * (JC[2] * a.x + JC[0] * b.x);
    dF[1] -= s * (JC[3] * a.x + JC[0] * b.y);
    dF[2] -= s * (JC[0]);
    dF[3] -= s * (JC[2] * a.y + JC[1] * b.x);
    dF[4] -= s * (JC[3] * a.y + JC[1] * b.y);
    dF[5] -= s * (JC[1]);
    dF[6] -= s * (JC[2]);
    dF[7] -= s * (JC[3]);
    for (int i = 0; i < 9; ++i)
        dF[i] *= inv;
    return r;
}
inline double sampson_residual(const M3 &E, const V2 &a, const V2 &b) { // relative.h:98-108
    const double Ea0 = E.m[0][0] * a.x + E.m[0][1] * a.y + E.m[0][2];
    const double Ea1 = E.m[1][0] * a.x + E.m[1][1] * a.y + E.m[1][2];
    const double Ea2 = E.m[2][0] * a.x + E.m[2][1] * a.y + E.m[2][2];
    const double C = b.x * Ea0 + b.y * Ea1 + Ea2;
    const double Eb0 = E.m[0][0] * b.x + E.m[1][0] * b.y + E.m[2][0];
    const double Eb1 = E.m[0][1] * b.x + E.m[1][1] * b.y + E.m[2][1];
    const double n2 = (Ea0 * Ea0 + Ea1 * Ea1) + (Eb0 * Eb0 + Eb1 * Eb1);
    return C / std::sqrt(n2);
}

struct RelProblem { // optim/relative.h:86-166
    static constexpr int K = 5;
    int k() const { return K; }
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    V3 tb0, tb1; // tangent basis of the translation, refreshed by jacobian()
    double residual(Normal &acc, const Pose &p) const {
        const M3 E = essential_from_motion(p);
        for (size_t k = 0; k < x1.size(); ++k)
            acc.add_residual(sampson_residual(E, x1[k], x2[k]));
        return acc.residual();
    }
    void jacobian(Normal &acc, const Pose &p) {
        const M3 R = p.R();
        const M3 E = essential_from_motion(p);
        // relative.h:63-83
        const V3 t = p.t;
        const V3 ex{1, 0, 0}, ey{0, 1, 0}, ez{0, 0, 1};
        if (std::abs(t.x) < std::abs(t.y))
            tb0 = normalized(cross(t, (std::abs(t.x) < std::abs(t.z)) ? ex : ez));
        else
            tb0 = normalized(cross(t, (std::abs(t.y) < std::abs(t.z)) ? ey : ez));
        tb1 = normalized(cross(tb0, t));
        // relative.h:39-61 : d vec(E) / d (rotation, translation-tangent)
        double dR[9][3], dt[9][2];
        const V3 e0 = E.col(0), e1 = E.col(1), e2 = E.col(2);
        auto put = [](double (*M)[3], int r0, int c, const V3 &v) {
            M[r0][c] = v.x;
            M[r0 + 1][c] = v.y;
            M[r0 + 2][c] = v.z;
        };
        const V3 zero{0, 0, 0};
        put(dR, 0, 0, zero), put(dR, 0, 1, -e2), put(dR, 0, 2, e1);
        put(dR, 3, 0, e2), put(dR, 3, 1, zero), put(dR, 3, 2, -e0);
        put(dR, 6, 0, -e1), put(dR, 6, 1, e0), put(dR, 6, 2, zero);
        for (int c = 0; c < 3; ++c) {
            const V3 a = cross(tb0, R.col(c)), b = cross(tb1, R.col(c));
            dt[3 * c][0] = a.x, dt[3 * c + 1][0] = a.y, dt[3 * c + 2][0] = a.z;
            dt[3 * c][1] = b.x, dt[3 * c + 1][1] = b.y, dt[3 * c + 2][1] = b.z;
        }
        for (size_t k = 0; k < x1.size(); ++k) {
            double dF[9];
            const double r = sampson_residual_and_grad(E, x1[k], x2[k], dF);
            double J[KMAX];
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int m = 0; m < 9; ++m)
                    s += dF[m] * dR[m][c];
                J[c] = s;
            }
            for (int c = 0; c < 2; ++c) {
                double s = 0;
                for (int m = 0; m < 9; ++m)
                    s += dF[m] * dt[m][c];
                J[3 + c] = s;
            }
            acc.add_jacobian(r, J);
        }
    }
    Pose step(const double *dp, const Pose &p) const {
        Pose out;
        out.q = quat_step_post(p.q, V3{dp[0], dp[1], dp[2]});
        out.t = V3{p.t.x + (tb0.x * dp[3] + tb1.x * dp[4]), p.t.y + (tb0.y * dp[3] + tb1.y * dp[4]),
                   p.t.z + (tb0.z * dp[3] + tb1.z * dp[4])};
        return out;
    }
};

// F = K_inv * E * K_inv with K_inv = diag(1, 1, f), evaluated left to right as the refiner does (relative.h:499-500, :519-520)
inline M3 focal_fundamental_lr(const M3 &E, double f) {
    M3 F = E;
    for (int j = 0; j < 3; ++j)
        F.m[2][j] = f * F.m[2][j];
    for (int i = 0; i < 3; ++i)
        F.m[i][2] = F.m[i][2] * f;
    return F;
}

struct SharedFocalProblem { // optim/relative.h:488-592 : rotation (3), translation tangent (2), shared focal length (1)
    static constexpr int K = 6;
    int k() const { return K; }
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    V3 tb0, tb1;
    double residual(Normal &acc, const ImagePair &p) const { // :496-510
        const M3 F = focal_fundamental_lr(essential_from_motion(p.pose), p.focal);
        for (size_t k = 0; k < x1.size(); ++k)
            acc.add_residual(sampson_residual(F, x1[k], x2[k]));
        return acc.residual();
    }
    void jacobian(Normal &acc, const ImagePair &p) { // :512-575
        const M3 R = p.pose.R();
        const M3 E = essential_from_motion(p.pose);
        const double focal = p.focal;
        const M3 F = focal_fundamental_lr(E, focal);
        const V3 t = p.pose.t; // relative.h:63-83
        const V3 ex{1, 0, 0}, ey{0, 1, 0}, ez{0, 0, 1};
        if (std::abs(t.x) < std::abs(t.y))
            tb0 = normalized(cross(t, (std::abs(t.x) < std::abs(t.z)) ? ex : ez));
        else
            tb0 = normalized(cross(t, (std::abs(t.y) < std::abs(t.z)) ? ey : ez));
        tb1 = normalized(cross(tb0, t));
        double dR[9][3], dt[9][2]; // relative.h:39-61
        const V3 e0 = E.col(0), e1 = E.col(1), e2 = E.col(2);
        auto put = [](double (*M)[3], int r0, int c, const V3 &v) {
            M[r0][c] = v.x;
            M[r0 + 1][c] = v.y;
            M[r0 + 2][c] = v.z;
        };
        const V3 zero{0, 0, 0};
        put(dR, 0, 0, zero), put(dR, 0, 1, -e2), put(dR, 0, 2, e1);
        put(dR, 3, 0, e2), put(dR, 3, 1, zero), put(dR, 3, 2, -e0);
        put(dR, 6, 0, -e1), put(dR, 6, 1, e0), put(dR, 6, 2, zero);
        for (int c = 0; c < 3; ++c) {
            const V3 a = cross(tb0, R.col(c)), b = cross(tb1, R.col(c));
            dt[3 * c][0] = a.x, dt[3 * c + 1][0] = a.y, dt[3 * c + 2][0] = a.z;
            dt[3 * c][1] = b.x, dt[3 * c + 1][1] = b.y, dt[3 * c + 2][1] = b.z;
        }
        const double ff = focal * focal; // :527-537
        const int once[4] = {2, 5, 6, 7};
        for (int m = 0; m < 4; ++m) {
            for (int c = 0; c < 3; ++c)
                dR[once[m]][c] *= focal;
            for (int c = 0; c < 2; ++c)
                dt[once[m]][c] *= focal;
        }
        for (int c = 0; c < 3; ++c)
            dR[8][c] *= ff;
        for (int c = 0; c < 2; ++c)
            dt[8][c] *= ff;
        const double df[9] = {0.0, 0.0, E.m[2][0], 0.0, 0.0, E.m[2][1], E.m[0][2], E.m[1][2], 2 * E.m[2][2] * focal}; // :540
        for (size_t k = 0; k < x1.size(); ++k) {
            double dF[9];
            const double r = sampson_residual_and_grad(F, x1[k], x2[k], dF);
            double J[KMAX];
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int m = 0; m < 9; ++m)
                    s += dF[m] * dR[m][c];
                J[c] = s;
            }
            for (int c = 0; c < 2; ++c) {
                double s = 0;
                for (int m = 0; m < 9; ++m)
                    s += dF[m] * dt[m][c];
                J[3 + c] = s;
            }
            double s = 0;
            for (int m = 0; m < 9; ++m)
                s += dF[m] * df[m];
            J[5] = s;
            acc.add_jacobian(r, J);
        }
    }
    ImagePair step(const double *dp, const ImagePair &p) const { // :577-585
        ImagePair out;
        out.pose.q = quat_step_post(p.pose.q, V3{dp[0], dp[1], dp[2]});
        out.pose.t = V3{p.pose.t.x + (tb0.x * dp[3] + tb1.x * dp[4]), p.pose.t.y + (tb0.y * dp[3] + tb1.y * dp[4]),
                        p.pose.t.z + (tb0.z * dp[3] + tb1.z * dp[4])};
        out.focal = p.focal + dp[5];
        return out;
    }
};

struct HomProblem { // optim/homography.h:46-178 : symmetric transfer error, first 8 entries of H (column-major)
    static constexpr int K = 8;
    int k() const { return K; }
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    static M3 adjugate(const M3 &H) {
        M3 A;
        A.m[0][0] = H.m[1][1] * H.m[2][2] - H.m[1][2] * H.m[2][1];
        A.m[0][1] = H.m[0][2] * H.m[2][1] - H.m[0][1] * H.m[2][2];
        A.m[0][2] = H.m[0][1] * H.m[1][2] - H.m[0][2] * H.m[1][1];
        A.m[1][0] = H.m[1][2] * H.m[2][0] - H.m[1][0] * H.m[2][2];
        A.m[1][1] = H.m[0][0] * H.m[2][2] - H.m[0][2] * H.m[2][0];
        A.m[1][2] = H.m[0][2] * H.m[1][0] - H.m[0][0] * H.m[1][2];
        A.m[2][0] = H.m[1][0] * H.m[2][1] - H.m[1][1] * H.m[2][0];
        A.m[2][1] = H.m[0][1] * H.m[2][0] - H.m[0][0] * H.m[2][1];
        A.m[2][2] = H.m[0][0] * H.m[1][1] - H.m[0][1] * H.m[1][0];
        return A;
    }
    static void transfer(const M3 &H, const V2 &a, double &z0, double &z1, double &inv) {
        const double h0 = H.m[0][0] * a.x + H.m[0][1] * a.y + H.m[0][2];
        const double h1 = H.m[1][0] * a.x + H.m[1][1] * a.y + H.m[1][2];
        inv = 1.0 / (H.m[2][0] * a.x + H.m[2][1] * a.y + H.m[2][2]);
        z0 = h0 * inv;
        z1 = h1 * inv;
    }
    double residual(Normal &acc, const M3 &H) const {
        const M3 G = adjugate(H);
        for (size_t k = 0; k < x1.size(); ++k) {
            double z0, z1, inv;
            transfer(H, x1[k], z0, z1, inv);
            acc.add_residual(z0 - x2[k].x, z1 - x2[k].y);
            transfer(G, x2[k], z0, z1, inv);
            acc.add_residual(z0 - x1[k].x, z1 - x1[k].y);
        }
        return acc.residual();
    }
    void jacobian(Normal &acc, const M3 &H) const {
        const M3 G = adjugate(H);
        const double H00 = H.m[0][0], H01 = H.m[0][1], H02 = H.m[0][2];
        const double H10 = H.m[1][0], H11 = H.m[1][1], H12 = H.m[1][2];
        const double H20 = H.m[2][0], H21 = H.m[2][1], H22 = H.m[2][2];
        for (size_t k = 0; k < x1.size(); ++k) {
            const double a0 = x1[k].x, a1 = x1[k].y, b0 = x2[k].x, b1 = x2[k].y;
            double z0, z1, inv;
            transfer(H, x1[k], z0, z1, inv);
            double J[2][KMAX] = {{a0, 0.0, -a0 * z0, a1, 0.0, -a1 * z0, 1.0, 0.0},
                                 {0.0, a0, -a0 * z1, 0.0, a1, -a1 * z1, 0.0, 1.0}};
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 8; ++c)
                    J[r][c] = J[r][c] * inv;
            acc.add_jacobian(z0 - b0, z1 - b1, J);

            double y0, y1, ginv;
            transfer(G, x2[k], y0, y1, ginv);
            const double y0b1 = y0 * b1, y0b0 = y0 * b0, y1b1 = y1 * b1, y1b0 = y1 * b0;
            double Jb[2][KMAX] = {
                {H21 * y0b1 - H11 * y0, H01 * y0 - H21 * y0b0, H11 * y0b0 - H01 * y0b1,
                 H12 - H22 * b1 + H10 * y0 - H20 * y0b1, H22 * b0 - H02 - H00 * y0 + H20 * y0b0,
                 H02 * b1 - H12 * b0 + H00 * y0b1 - H10 * y0b0, H21 * b1 - H11, H01 - H21 * b0},
                {H22 * b1 - H12 - H11 * y1 + H21 * y1b1, H02 - H22 * b0 + H01 * y1 - H21 * y1b0,
                 H12 * b0 - H02 * b1 - H01 * y1b1 + H11 * y1b0, H10 * y1 - H20 * y1b1, H20 * y1b0 - H00 * y1,
                 H00 * y1b1 - H10 * y1b0, H10 - H20 * b1, H20 * b0 - H00}};
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 8; ++c)
                    Jb[r][c] = Jb[r][c] * ginv;
            acc.add_jacobian(y0 - a0, y1 - a1, Jb);
        }
    }
    M3 step(const double *dp, const M3 &H) const {
        M3 out = H;
        for (int e = 0; e < 8; ++e) // column-major position e  <->  (row e%3, col e/3)
            out.m[e % 3][e / 3] += dp[e];
        return out;
    }
};

struct FactF { // optim_utils.h:57-82 (Bartoli-Sturm factorisation)
    V4 qU, qV;
    double sigma = 0;
    M3 F() const {
        const M3 U = quat_to_rotmat(qU), V = quat_to_rotmat(qV);
        M3 out;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                out.m[i][j] = U.m[i][0] * V.m[j][0] + (sigma * U.m[i][1]) * V.m[j][1];
        return out;
    }
};

struct FundProblem { // optim/fundamental.h:41-121
    static constexpr int K = 7;
    int k() const { return K; }
    const std::vector<V2> &x1;
    const std::vector<V2> &x2;
    double residual(Normal &acc, const FactF &ff) const {
        const M3 F = ff.F();
        for (size_t k = 0; k < x1.size(); ++k)
            acc.add_residual(sampson_residual(F, x1[k], x2[k]));
        return acc.residual();
    }
    void jacobian(Normal &acc, const FactF &ff) const {
        const M3 F = ff.F();
        const M3 U = quat_to_rotmat(ff.qU), V = quat_to_rotmat(ff.qV);
        // d vec(F) / d(wU, wV, sigma):  columns 0-2 = vec([e_k]x F), 3-5 = vec(F [e_k]x^T), 6 = vec(u1 v1^T)
        double D[9][7];
        for (int c = 0; c < 3; ++c) {
            const V3 f = F.col(c);
            const V3 axes[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            for (int k = 0; k < 3; ++k) {
                const V3 d = cross(axes[k], f);
                D[3 * c][k] = d.x, D[3 * c + 1][k] = d.y, D[3 * c + 2][k] = d.z;
            }
        }
        for (int r = 0; r < 3; ++r) {
            const V3 f = F.row(r);
            const V3 axes[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            for (int k = 0; k < 3; ++k) {
                const V3 d = cross(axes[k], f); // (F [e_k]x^T)(r, :) = e_k x F(r,:)
                D[r][3 + k] = d.x, D[3 + r][3 + k] = d.y, D[6 + r][3 + k] = d.z;
            }
        }
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i)
                D[3 * j + i][6] = U.m[i][1] * V.m[j][1];
        for (size_t k = 0; k < x1.size(); ++k) {
            double dF[9];
            const double r = sampson_residual_and_grad(F, x1[k], x2[k], dF);
            double J[KMAX];
            for (int c = 0; c < 7; ++c) {
                double s = 0;
                for (int m = 0; m < 9; ++m)
                    s += dF[m] * D[m][c];
                J[c] = s;
            }
            acc.add_jacobian(r, J);
        }
    }
    FactF step(const double *dp, const FactF &f) const {
        FactF out;
        out.qU = quat_step_pre(f.qU, V3{dp[0], dp[1], dp[2]});
        out.qV = quat_step_pre(f.qV, V3{dp[3], dp[4], dp[5]});
        out.sigma = f.sigma + dp[6];
        return out;
    }
};

} // namespace

// ------------------------------------------------------------------------------------ SVD
void svd3(const M3 &Ain, M3 &U, double s[3], M3 &V) {
    // Eigen::JacobiSVD<Matrix3d>(F, ComputeFullU | ComputeFullV) of optim_utils.h:60: the two-sided Jacobi iteration
    // in Eigen's operation order - the same routine the shim's JacobiSVD (reference sources) runs; the sign of the
    // refined F depends on it (JacobiSVD3x3.h).
    double sv[3];
    eigen_shim_detail::jacobi_svd3(Ain.m, U.m, sv, V.m);
    s[0] = sv[0], s[1] = sv[1], s[2] = sv[2];
}

// ------------------------------------------------------------------------------------ entry points
BundleStats bundle_adjust(const std::vector<V2> &x, const std::vector<V3> &X, Image *image, const BundleOptions &opt) {
    // bundle.cc:99-103: the camera parameters named by the options are refined along with the pose
    AbsProblem prob{x, X, image->camera.refinement_idx(opt.refine_focal_length, opt.refine_principal_point, opt.refine_extra_params)};
    return levenberg_marquardt(prob, image, opt);
}
BundleStats bundle_adjust(const std::vector<V2> &x, const std::vector<V3> &X, Pose *pose, const BundleOptions &opt) {
    Image im; // bundle.cc:84-92 : identity ("NULL") camera
    im.pose = *pose;
    im.camera.model_id = CAM_NULL;
    const BundleStats st = bundle_adjust(x, X, &im, opt);
    *pose = im.pose;
    return st;
}
BundleStats refine_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, Pose *pose, const BundleOptions &opt) {
    RelProblem prob{x1, x2, {}, {}};
    return levenberg_marquardt(prob, pose, opt);
}
BundleStats refine_shared_focal_relpose(const std::vector<V2> &x1, const std::vector<V2> &x2, ImagePair *pair,
                                        const BundleOptions &opt) { // bundle.cc:281-297
    SharedFocalProblem prob{x1, x2, {}, {}};
    return levenberg_marquardt(prob, pair, opt);
}
BundleStats refine_homography(const std::vector<V2> &x1, const std::vector<V2> &x2, M3 *H, const BundleOptions &opt) {
    HomProblem prob{x1, x2};
    return levenberg_marquardt(prob, H, opt);
}
BundleStats refine_fundamental(const std::vector<V2> &x1, const std::vector<V2> &x2, M3 *F, const BundleOptions &opt) {
    M3 U, V;
    double s[3];
    svd3(*F, U, s, V);
    if (det(U) < 0)
        U = U * -1.0;
    if (det(V) < 0)
        V = V * -1.0;
    FactF ff;
    ff.qU = rotmat_to_quat(U);
    ff.qV = rotmat_to_quat(V);
    ff.sigma = s[1] / s[0];
    FundProblem prob{x1, x2};
    const BundleStats st = levenberg_marquardt(prob, &ff, opt);
    *F = ff.F();
    return st;
}

} // namespace orc
