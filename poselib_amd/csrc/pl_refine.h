// poselib_amd — Levenberg-Marquardt building blocks for local optimisation (LO) and the final
// polish.  The per-correspondence residual / Jacobian functions and the LM control logic are
// PL_HD: the device kernel (kernels.hip, one workgroup per refinement task, reductions through
// DPP + LDS) and the host-side unit test build share this file.
//
// Reference semantics followed (citations relative to /root/reference/PoseLib):
//   robust/optim/lm_impl.h:56-140            LM driver (Nielsen / fixed-factor lambda, stop rules)
//   robust/optim/jacobian_accumulator.h      normal equations; NOTE the single residual_count that
//                                            both passes overwrite (:62-64,78-82,166) — reproduced.
//   robust/robust_loss.h:59-157              losses
//   robust/optim/absolute.h:40-171           reprojection refiner (no intrinsics refined)
//   robust/optim/relative.h:39-166           Sampson refiner on (R, t) with a tangent basis for t
//   robust/optim/homography.h:46-178         symmetric transfer error, 8 parameters
//   robust/optim/fundamental.h:41-121 + optim_utils.h:57-82   Sampson on the Bartoli-Sturm factorisation
//   misc/camera_models.cc:668-755,919-1032,2704-2726  projections with Jacobians (NULL, SIMPLE_PINHOLE,
//                                            PINHOLE, OPENCV)
#pragma once
#include "pl_math.h"
#include "pl_score.h"

namespace pl {

enum LossType : int { LOSS_TRIVIAL = 0, LOSS_TRUNCATED, LOSS_HUBER, LOSS_CAUCHY, LOSS_TRUNCATED_CAUCHY, LOSS_TRUNCATED_LE_ZACH };
enum CameraId : int { CAM_NULL = -1, CAM_SIMPLE_PINHOLE = 0, CAM_PINHOLE = 1, CAM_OPENCV = 4 };

struct LMOptions {
    uint32_t max_iterations;
    int32_t loss_type, lambda_update, damping;
    double loss_scale, gradient_tol, step_tol, relative_cost_tol, initial_lambda, min_lambda, max_lambda, lambda_factor;
};

// ------------------------------------------------------------------------------------ losses
struct Loss {
    int type;
    double thr, sq, inv_sq, max_loss, mu;
};
PL_HD Loss make_loss(int type, double scale) {
    Loss l;
    l.type = type;
    l.thr = scale;
    l.sq = scale * scale;
    l.inv_sq = 1.0 / l.sq;
    l.max_loss = l.sq * log1p(1.0);
    l.mu = 0.5;
    return l;
}
PL_HD double loss_value(const Loss &l, double r2) {
    switch (l.type) {
    case LOSS_TRUNCATED:
    case LOSS_TRUNCATED_LE_ZACH:
        return fmin(r2, l.sq);
    case LOSS_HUBER: {
        const double r = sqrt(r2);
        return (r <= l.thr) ? r2 : l.thr * (2.0 * r - l.thr);
    }
    case LOSS_CAUCHY:
        return l.sq * log1p(r2 * l.inv_sq);
    case LOSS_TRUNCATED_CAUCHY:
        return (r2 > l.sq) ? l.max_loss : l.sq * log1p(r2 * l.inv_sq);
    default:
        return r2;
    }
}
PL_HD double loss_weight(const Loss &l, double r2) {
    const double dmin = 2.2250738585072014e-308;
    switch (l.type) {
    case LOSS_TRUNCATED:
        return (r2 < l.sq) ? 1.0 : 0.0;
    case LOSS_TRUNCATED_LE_ZACH: {
        const double rh = r2 / l.sq;
        const double zstar = fmin(rh, 1.0);
        if (rh < 1.0)
            return 0.5;
        const double m1 = rh - 1.0;
        const double rho = (2.0 * m1 + sqrt(4.0 * m1 * m1 * l.mu * l.mu + 2 * l.mu * m1)) / l.mu;
        const double a = (rh + l.mu * rho * zstar - 0.5 * rho) / (1 + l.mu * rho);
        const double zbar = fmax(0.0, fmin(a, 1.0));
        return (zstar - zbar) / rho;
    }
    case LOSS_HUBER: {
        const double r = sqrt(r2);
        return (r <= l.thr) ? 1.0 : l.thr / r;
    }
    case LOSS_CAUCHY:
        return fmax(dmin, 1.0 / (1.0 + r2 * l.inv_sq));
    case LOSS_TRUNCATED_CAUCHY:
        return (r2 > l.sq) ? 0.0 : fmax(dmin, 1.0 / (1.0 + r2 * l.inv_sq));
    default:
        return 1.0;
    }
}

// ------------------------------------------------------------------------------------ camera
struct CameraParams {
    int32_t model_id;
    int32_t num_params;
    double p[12];
};

PL_HD void opencv_distort(double k1, double k2, double p1, double p2, double u, double v, double &du, double &dv,
                          double *J /*2x2 row-major*/) {
    const double u2 = u * u, uv = u * v, v2 = v * v;
    const double r2 = u * u + v * v;
    J[0] = k2 * r2 * r2 + 6 * p2 * u + 2 * p1 * v + u * (2 * k1 * u + 4 * k2 * u * r2) + k1 * r2 + 1.0;
    J[1] = 2 * p1 * u + 2 * p2 * v + v * (2 * k1 * u + 4 * k2 * u * r2);
    J[2] = 2 * p1 * u + 2 * p2 * v + u * (2 * k1 * v + 4 * k2 * v * r2);
    J[3] = k2 * r2 * r2 + 2 * p2 * u + 6 * p1 * v + v * (2 * k1 * v + 4 * k2 * v * r2) + k1 * r2 + 1.0;
    const double alpha = 1.0 + k1 * r2 + k2 * r2 * r2;
    du = alpha * u + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
    dv = alpha * v + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
}

// pixel -> normalised image plane (camera_models.h:98-102: unit bearing first, then hnormalized)
PL_HD void camera_unproject(const CameraParams &c, double px, double py, double &ox, double &oy) {
    double u, v;
    switch (c.model_id) {
    case CAM_SIMPLE_PINHOLE:
        u = (px - c.p[1]) / c.p[0];
        v = (py - c.p[2]) / c.p[0];
        break;
    case CAM_PINHOLE:
        u = (px - c.p[2]) / c.p[0];
        v = (py - c.p[3]) / c.p[1];
        break;
    case CAM_OPENCV: {
        const double tx = (px - c.p[2]) / c.p[0], ty = (py - c.p[3]) / c.p[1];
        u = tx, v = ty;
        for (int it = 0; it < 100; ++it) { // camera_models.cc:972-990
            double du, dv, J[4];
            opencv_distort(c.p[4], c.p[5], c.p[6], c.p[7], u, v, du, dv, J);
            J[0] += 1e-8;
            J[3] += 1e-8;
            const double r0 = du - tx, r1 = dv - ty;
            if (sqrt(r0 * r0 + r1 * r1) < 1e-10)
                break;
            const double dt = J[0] * J[3] - J[2] * J[1];
            const double id = 1.0 / dt;
            const double s0 = (J[3] * id) * r0 + (-J[1] * id) * r1;
            const double s1 = (-J[2] * id) * r0 + (J[0] * id) * r1;
            u = u - s0;
            v = v - s1;
        }
        break;
    }
    default: // CAM_NULL: bearing is (x, y, 1) un-normalised
        ox = px / 1.0;
        oy = py / 1.0;
        return;
    }
    const Vec3 b = normalized(v3(u, v, 1.0));
    ox = b.x / b.z;
    oy = b.y / b.z;
}

PL_HD void camera_project(const CameraParams &c, Vec3 Z, double &ox, double &oy) {
    switch (c.model_id) {
    case CAM_SIMPLE_PINHOLE:
        ox = c.p[0] * Z.x / Z.z + c.p[1];
        oy = c.p[0] * Z.y / Z.z + c.p[2];
        return;
    case CAM_PINHOLE:
        ox = c.p[0] * Z.x / Z.z + c.p[2];
        oy = c.p[1] * Z.y / Z.z + c.p[3];
        return;
    case CAM_OPENCV: {
        double du, dv, J[4];
        opencv_distort(c.p[4], c.p[5], c.p[6], c.p[7], Z.x / Z.z, Z.y / Z.z, du, dv, J);
        ox = c.p[0] * du + c.p[2];
        oy = c.p[1] * dv + c.p[3];
        return;
    }
    default:
        ox = Z.x / Z.z;
        oy = Z.y / Z.z;
    }
}
// projection + d(xp)/dZ (2x3 row-major)
PL_HD void camera_project_jac(const CameraParams &c, Vec3 Z, double &ox, double &oy, double *J) {
    switch (c.model_id) {
    case CAM_SIMPLE_PINHOLE:
    case CAM_PINHOLE: {
        const bool simple = c.model_id == CAM_SIMPLE_PINHOLE;
        const double fx = c.p[0], fy = simple ? c.p[0] : c.p[1];
        const double cx = simple ? c.p[1] : c.p[2], cy = simple ? c.p[2] : c.p[3];
        const double zi = 1.0 / Z.z;
        const double px = fx * Z.x * zi, py = fy * Z.y * zi;
        J[0] = fx * zi, J[1] = 0.0, J[2] = -px * zi;
        J[3] = 0.0, J[4] = fy * zi, J[5] = -py * zi;
        ox = px + cx;
        oy = py + cy;
        return;
    }
    case CAM_OPENCV: {
        const double u = Z.x / Z.z, v = Z.y / Z.z;
        double du, dv, Jd[4];
        opencv_distort(c.p[4], c.p[5], c.p[6], c.p[7], u, v, du, dv, Jd);
        const double P[6] = {1.0 / Z.z, 0.0, -u / Z.z, 0.0, 1.0 / Z.z, -v / Z.z};
        PL_UNROLL
        for (int a = 0; a < 2; ++a)
            PL_UNROLL
            for (int b = 0; b < 3; ++b)
                J[3 * a + b] = Jd[2 * a] * P[b] + Jd[2 * a + 1] * P[3 + b];
        PL_UNROLL
        for (int b = 0; b < 3; ++b) {
            J[b] *= c.p[0];
            J[3 + b] *= c.p[1];
        }
        ox = c.p[0] * du + c.p[2];
        oy = c.p[1] * dv + c.p[3];
        return;
    }
    default: {
        ox = Z.x / Z.z;
        oy = Z.y / Z.z;
        const double zi = 1.0 / Z.z;
        J[0] = zi, J[1] = 0.0, J[2] = -ox * zi;
        J[3] = 0.0, J[4] = zi, J[5] = -oy * zi;
    }
    }
}

// ------------------------------------------------------------------------------------ accumulators
// Per-thread partial of the normal equations:  lower triangle of J^T J (row-major packed:
// (i,j), j<=i at i*(i+1)/2 + j) followed by J^T r.  Size K*(K+1)/2 + K.
template <int K> struct NormalSize {
    static constexpr int kTri = K * (K + 1) / 2;
    static constexpr int kTotal = kTri + K;
};

template <int K> PL_HD void accumulate2(double *acc, const Loss &loss, double r0, double r1, const double *J /*2xK*/,
                                        uint32_t &count) {
    const double w = 1.0 * loss_weight(loss, r0 * r0 + r1 * r1);
    if (w == 0)
        return;
    int o = 0;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        PL_UNROLL
        for (int j = 0; j <= i; ++j)
            acc[o++] += w * (J[i] * J[j] + J[K + i] * J[K + j]);
    const double wr0 = w * r0, wr1 = w * r1;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        acc[o + i] += J[i] * wr0 + J[K + i] * wr1;
    count++;
}
template <int K> PL_HD void accumulate1(double *acc, const Loss &loss, double r, const double *J /*K*/,
                                        uint32_t &count) {
    const double w = 1.0 * loss_weight(loss, r * r);
    if (w == 0)
        return;
    int o = 0;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        PL_UNROLL
        for (int j = 0; j <= i; ++j)
            acc[o++] += w * (J[i] * J[j]);
    const double wr = w * r;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        acc[o + i] += wr * J[i];
    count++;
}

// The same terms as accumulate2 / accumulate1, handed to `store(entry, value)` instead of added to an accumulator: k_lm's producers
// write them as a row of the term ring and the consumer adds row after row (the reference's `+=` per entry and correspondence,
// jacobian_accumulator.h:82-97).  `acc += v` with the value written here equals the reference's `acc += v` bit for bit; a
// correspondence with weight zero contributes nothing there (early return) and a row of zeros here (x + 0.0 = x).
template <int K, class Store> PL_HD void terms2(const Loss &loss, double r0, double r1, const double *J /*2xK*/, uint32_t &count, Store store) {
    const double w = 1.0 * loss_weight(loss, r0 * r0 + r1 * r1);
    if (w == 0) {
        PL_UNROLL
        for (int o = 0; o < NormalSize<K>::kTotal; ++o)
            store(o, 0.0);
        return;
    }
    int o = 0;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        PL_UNROLL
        for (int j = 0; j <= i; ++j)
            store(o++, w * (J[i] * J[j] + J[K + i] * J[K + j]));
    const double wr0 = w * r0, wr1 = w * r1;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        store(o + i, J[i] * wr0 + J[K + i] * wr1);
    count++;
}
template <int K, class Store> PL_HD void terms1(const Loss &loss, double r, const double *J /*K*/, uint32_t &count, Store store) {
    const double w = 1.0 * loss_weight(loss, r * r);
    if (w == 0) {
        PL_UNROLL
        for (int o = 0; o < NormalSize<K>::kTotal; ++o)
            store(o, 0.0);
        return;
    }
    int o = 0;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        PL_UNROLL
        for (int j = 0; j <= i; ++j)
            store(o++, w * (J[i] * J[j]));
    const double wr = w * r;
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        store(o + i, wr * J[i]);
    count++;
}

// (2 rho - 1)^3 of the Nielsen update (lm_impl.h:124: std::pow(2.0 * rho - 1.0, 3)), see the note at lm_update
// The reference calls the host's libm here - glibc's pow, whose result for the exponent 3 is the correctly rounded cube
// for 99.9 % of the arguments (measured: 183 exceptions in 2*10^5; the device library's pow differs from it for 23 %).  The
// device therefore forms the (nearly) correctly rounded cube itself: x^2 = hi + lo and hi x = p + e exactly (FMA residuals), then
// one rounding of p + (e + lo x).  The value only matters for mediocre steps (factor = 1 - cube > 1/3, i.e. rho < 0.94).
// (the FMA form is a plain function so that the host test build can compare it with glibc's pow: tests/test_libm_vs_glibc.py)
PL_HD double lm_cube_fma(double x) {
    if (!(fabs(x) < 1e100) || fabs(x) < 1e-100)
        return x * x * x; // inf / NaN / overflow / underflow: as pow
    const double hi = x * x;
    const double lo = __builtin_fma(x, x, -hi);
    const double p = hi * x;
    const double e = __builtin_fma(hi, x, -p);
    return p + (e + lo * x); // (lo x and the inner sum round once each: NEARLY correctly rounded, see above)
}
PL_HD double lm_cube(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return lm_cube_fma(x);
#else
    return pow(x, 3);
#endif
}

// ------------------------------------------------------------------------------------ LM control
// Parameter block of a refinement task: 16 doubles.
//   absolute : [0..3] q, [4..6] t
//   relative : [0..3] q, [4..6] t, [7..9] tangent basis 0, [10..12] tangent basis 1
//   homography : [0..8] H row-major
//   fundamental: [0..3] qU, [4..7] qV, [8] sigma
constexpr int kParamDoubles = 16;

struct LMControl {
    LMOptions opt;
    Loss loss;
    double cost, initial_cost, lambda, nu, step_norm, grad_norm;
    uint32_t iterations, invalid_steps;
    uint32_t count; // the accumulator's single residual counter (see header comment)
    int32_t rejac;
    int32_t done;
    double sol[16]; // (up to 14: pose + 8 camera parameters, pl_refine_cam.h)
};

PL_HD double lm_scale(uint32_t count) { return 1.0 / fmax(1.0, (double)count); }

PL_HD void lm_begin(LMControl &c, const LMOptions &opt, double racc, uint32_t count) {
    c.opt = opt;
    c.count = count;
    c.cost = racc * lm_scale(count);
    c.initial_cost = c.cost;
    c.grad_norm = -1;
    c.step_norm = -1;
    c.invalid_steps = 0;
    c.lambda = opt.initial_lambda;
    c.nu = 2.0;
    c.rejac = 1;
    c.iterations = 0;
    c.done = (opt.max_iterations == 0) ? 1 : 0;
}

// After (optionally) a fresh Jacobian pass: gradient test, damped solve, step-size test.
// `normal` = reduced [tri | Jtr]; `jac_count` is the counter after the Jacobian pass (ignored when
// the Jacobian was not recomputed).  Sets c.done or leaves c.sol for the trial step.
template <int K> PL_HD void lm_solve(LMControl &c, const double *normal, bool fresh_jacobian, uint32_t jac_count) {
    constexpr int T = NormalSize<K>::kTri;
    if (fresh_jacobian) {
        c.count = jac_count;
        double s = 0;
        PL_UNROLL
        for (int i = 0; i < K; ++i)
            s += normal[T + i] * normal[T + i];
        c.grad_norm = lm_scale(c.count) * sqrt(s);
        if (c.grad_norm < c.opt.gradient_tol) {
            c.done = 1;
            return;
        }
    }
    const double sc = lm_scale(c.count);
    double A[K * K], b[K];
    PL_UNROLL
    for (int i = 0; i < K; ++i) {
        PL_UNROLL
        for (int j = 0; j <= i; ++j)
            A[i * K + j] = sc * normal[i * (i + 1) / 2 + j];
        b[i] = -(sc * normal[T + i]);
    }
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        A[i * K + i] += (c.opt.damping == 1) ? fmax(A[i * K + i] * c.lambda, 1e-8) : c.lambda;
    // Cholesky with the operation order of Eigen's unblocked LLT (squared norm / dot product summed first, then
    // subtracted), forward solve by column-oriented updates, backward solve by row dot product then subtract
    PL_UNROLL
    for (int col = 0; col < K; ++col) {
        double d = A[col * K + col];
        if (col > 0) {
            double sq = A[col * K] * A[col * K];
            PL_UNROLL
            for (int m = 1; m < col; ++m)
                sq += A[col * K + m] * A[col * K + m];
            d -= sq;
        }
        if (d <= 0)
            break;
        d = sqrt(d);
        A[col * K + col] = d;
        PL_UNROLL
        for (int r = col + 1; r < K; ++r) {
            double s = A[r * K + col];
            if (col > 0) {
                double dot = A[r * K] * A[col * K];
                PL_UNROLL
                for (int m = 1; m < col; ++m)
                    dot += A[r * K + m] * A[col * K + m];
                s -= dot;
            }
            A[r * K + col] = s / d;
        }
    }
    // (the solves run on a local copy: c lives in LDS on the device, and a read-modify-write chain on c.sol was a chain of LDS round
    // trips - cycle counters in k_lm: solve + step 29 % of an LM iteration of a small problem)
    double x[K];
    PL_UNROLL
    for (int i = 0; i < K; ++i)
        x[i] = b[i];
    PL_UNROLL
    for (int i = 0; i < K; ++i) {
        x[i] /= A[i * K + i];
        PL_UNROLL
        for (int r = i + 1; r < K; ++r)
            x[r] -= x[i] * A[r * K + i];
    }
    PL_UNROLL
    for (int i = K - 1; i >= 0; --i) {
        if (i + 1 < K) {
            double dot = A[(i + 1) * K + i] * x[i + 1];
            PL_UNROLL
            for (int j = i + 2; j < K; ++j)
                dot += A[j * K + i] * x[j];
            x[i] -= dot;
        }
        x[i] /= A[i * K + i];
    }
    double sn = 0;
    PL_UNROLL
    for (int i = 0; i < K; ++i) {
        c.sol[i] = x[i];
        sn += x[i] * x[i];
    }
    c.step_norm = sqrt(sn);
    if (c.step_norm < c.opt.step_tol)
        c.done = 1;
}

// After the residual pass on the trial parameters.  Returns true when the step is accepted.
template <int K> PL_HD bool lm_update(LMControl &c, const double *normal, double racc, uint32_t res_count) {
    constexpr int T = NormalSize<K>::kTri;
    c.count = res_count;
    const double cost_new = racc * lm_scale(res_count);
    bool accepted = false;
    bool stop = false;
    if (cost_new < c.cost) {
        const double decrease = c.cost - cost_new;
        accepted = true;
        c.cost = cost_new;
        c.rejac = 1;
        if (c.opt.lambda_update == 0) {
            const double sc = lm_scale(c.count);
            double s = 0;
            PL_UNROLL
            for (int i = 0; i < K; ++i)
                s += c.sol[i] * (c.lambda * c.sol[i] + sc * normal[T + i]);
            const double pred = -s;
            if (pred > 0) {
                const double rho = decrease / pred;
                const double factor = 1.0 - lm_cube(2.0 * rho - 1.0);
                c.lambda *= fmax(1.0 / 3.0, factor);
            } else {
                c.lambda *= 1.0 / 3.0;
            }
            c.nu = 2.0;
        } else {
            c.lambda /= c.opt.lambda_factor;
        }
        c.lambda = fmax(c.opt.min_lambda, c.lambda);
        if (c.cost > 0 && decrease / c.cost < c.opt.relative_cost_tol)
            stop = true;
    } else {
        c.invalid_steps++;
        c.rejac = 0;
        if (c.opt.lambda_update == 0) {
            c.lambda *= c.nu;
            c.nu *= 2.0;
        } else {
            c.lambda *= c.opt.lambda_factor;
        }
        c.lambda = fmin(c.opt.max_lambda, c.lambda);
    }
    if (stop) {
        c.done = 1;
    } else {
        if (c.opt.loss_type == LOSS_TRUNCATED_LE_ZACH)
            c.loss.mu *= 1.5; // bundle.cc:52-75 callback
        c.iterations++;
        if (c.iterations >= c.opt.max_iterations)
            c.done = 1;
    }
    return accepted;
}

// ------------------------------------------------------------------------------------ refiners
// Uniform (per-task) context derived from the current parameters.
struct RefineCtx {
    double M[9];    // R (abs), E (rel), H (hom), F (fund)    row-major
    double G[9];    // adjugate(H) (hom)
    double D[9 * 7]; // d vec(E or F) / d params  (column-major vec index m, D[m*7 + c])
};

template <int EST> struct Refiner;

// ---- absolute pose ----
template <> struct Refiner<EST_ABS> {
    static constexpr int K = 6;
    PL_HD static void prepare(const double *p, RefineCtx &c) {
        Quat q;
        q.w = p[0], q.x = p[1], q.y = p[2], q.z = p[3];
        const Mat3 R = quat_to_rotmat(q);
        PL_UNROLL
        for (int i = 0; i < 9; ++i)
            c.M[i] = R.m[i];
    }
    // returns false when the point is skipped (behind the camera)
    PL_HD static bool residual(const double *p, const RefineCtx &c, const CameraParams &cam, double x, double y,
                               double X, double Y, double Z, double &r0, double &r1) {
        const double *R = c.M;
        const Vec3 Zc = v3(R[0] * X + R[1] * Y + R[2] * Z + p[4], R[3] * X + R[4] * Y + R[5] * Z + p[5],
                           R[6] * X + R[7] * Y + R[8] * Z + p[6]);
        if (Zc.z < 0)
            return false;
        double px, py;
        camera_project(cam, Zc, px, py);
        r0 = px - x;
        r1 = py - y;
        return true;
    }
    PL_HD static bool jacobian(const double *p, const RefineCtx &c, const CameraParams &cam, double x, double y,
                               double X, double Y, double Z, double &r0, double &r1, double *J /*2x6*/) {
        const double *R = c.M;
        const Vec3 Zc = v3(R[0] * X + R[1] * Y + R[2] * Z + p[4], R[3] * X + R[4] * Y + R[5] * Z + p[5],
                           R[6] * X + R[7] * Y + R[8] * Z + p[6]);
        if (Zc.z < 0)
            return false;
        double px, py, Jp[6];
        camera_project_jac(cam, Zc, px, py, Jp);
        r0 = px - x;
        r1 = py - y;
        PL_UNROLL
        for (int a = 0; a < 2; ++a) {
            const double d0 = Jp[3 * a] * R[0] + Jp[3 * a + 1] * R[3] + Jp[3 * a + 2] * R[6];
            const double d1 = Jp[3 * a] * R[1] + Jp[3 * a + 1] * R[4] + Jp[3 * a + 2] * R[7];
            const double d2 = Jp[3 * a] * R[2] + Jp[3 * a + 1] * R[5] + Jp[3 * a + 2] * R[8];
            J[6 * a + 0] = -Z * d1 + Y * d2;
            J[6 * a + 1] = Z * d0 - X * d2;
            J[6 * a + 2] = -Y * d0 + X * d1;
            J[6 * a + 3] = d0;
            J[6 * a + 4] = d1;
            J[6 * a + 5] = d2;
        }
        return true;
    }
    PL_HD static void step(const double *__restrict__ p, const RefineCtx &, const double *__restrict__ dp, double *__restrict__ out) { // (cur, the LM step, trial: never the same storage)
        Quat q;
        q.w = p[0], q.x = p[1], q.y = p[2], q.z = p[3];
        const Quat qn = quat_step_post(q, v3(dp[0], dp[1], dp[2]));
        const Vec3 dt = quat_rotate(q, v3(dp[3], dp[4], dp[5]));
        PL_UNROLL
        for (int i = 0; i < kParamDoubles; ++i)
            out[i] = p[i];
        out[0] = qn.w, out[1] = qn.x, out[2] = qn.y, out[3] = qn.z;
        out[4] = p[4] + dt.x, out[5] = p[5] + dt.y, out[6] = p[6] + dt.z;
    }
};

// Sampson residual and its gradient w.r.t. vec(E) (column-major), shared by E and F refiners
// (optim/relative.h:98-140).
PL_HD double sampson_residual(const double *E, double a0, double a1, double b0, double b1) {
    const double Ea0 = E[0] * a0 + E[1] * a1 + E[2];
    const double Ea1 = E[3] * a0 + E[4] * a1 + E[5];
    const double Ea2 = E[6] * a0 + E[7] * a1 + E[8];
    const double C = b0 * Ea0 + b1 * Ea1 + Ea2;
    const double Eb0 = E[0] * b0 + E[3] * b1 + E[6];
    const double Eb1 = E[1] * b0 + E[4] * b1 + E[7];
    const double n2 = (Ea0 * Ea0 + Ea1 * Ea1) + (Eb0 * Eb0 + Eb1 * Eb1);
    return C / sqrt(n2);
}
PL_HD double sampson_residual_grad(const double *E, double a0, double a1, double b0, double b1, double *dF) {
    const double Ea0 = E[0] * a0 + E[1] * a1 + E[2];
    const double Ea1 = E[3] * a0 + E[4] * a1 + E[5];
    const double Ea2 = E[6] * a0 + E[7] * a1 + E[8];
    const double C = b0 * Ea0 + b1 * Ea1 + Ea2;
    const double J0 = E[0] * b0 + E[3] * b1 + E[6];
    const double J1 = E[1] * b0 + E[4] * b1 + E[7];
    const double J2 = Ea0, J3 = Ea1;
    const double nJ = sqrt(J0 * J0 + J1 * J1 + J2 * J2 + J3 * J3);
    const double inv = 1.0 / nJ;
    const double r = C * inv;
    dF[0] = a0 * b0, dF[1] = a0 * b1, dF[2] = a0;
    dF[3] = a1 * b0, dF[4] = a1 * b1, dF[5] = a1;
    dF[6] = b0, dF[7] = b1, dF[8] = 1.0;
    const double s = C * inv * inv;
    dF[0] -= s * (J2 * a0 + J0 * b0);
    dF[1] -= s * (J3 * a0 + J0 * b1);
    dF[2] -= s * (J0);
    dF[3] -= s * (J2 * a1 + J1 * b0);
    dF[4] -= s * (J3 * a1 + J1 * b1);
    dF[5] -= s * (J1);
    dF[6] -= s * (J2);
    dF[7] -= s * (J3);
    PL_UNROLL
    for (int i = 0; i < 9; ++i)
        dF[i] *= inv;
    return r;
}

// ---- relative pose ----
template <> struct Refiner<EST_REL> {
    static constexpr int K = 5;
    // prepare() also refreshes the tangent basis in the parameter block when `for_jacobian`
    // (the reference sets it up inside compute_jacobian, relative.h:113, and step() reads it).
    PL_HD static void prepare_params(double *p) {
        const Vec3 t = v3(p[4], p[5], p[6]);
        Vec3 tb0;
        if (fabs(t.x) < fabs(t.y))
            tb0 = normalized(cross(t, (fabs(t.x) < fabs(t.z)) ? v3(1, 0, 0) : v3(0, 0, 1)));
        else
            tb0 = normalized(cross(t, (fabs(t.y) < fabs(t.z)) ? v3(0, 1, 0) : v3(0, 0, 1)));
        const Vec3 tb1 = normalized(cross(tb0, t));
        p[7] = tb0.x, p[8] = tb0.y, p[9] = tb0.z;
        p[10] = tb1.x, p[11] = tb1.y, p[12] = tb1.z;
    }
    PL_HD static void prepare(const double *p, RefineCtx &c) {
        Quat q;
        q.w = p[0], q.x = p[1], q.y = p[2], q.z = p[3];
        const Mat3 R = quat_to_rotmat(q);
        const Vec3 t = v3(p[4], p[5], p[6]);
        const Mat3 E = essential_from_motion(R, t);
        PL_UNROLL
        for (int i = 0; i < 9; ++i)
            c.M[i] = E.m[i];
        // D[m][0..2] = d vec(E)/d rot, D[m][3..4] = d vec(E)/d tangent   (relative.h:39-61)
        const Vec3 e0 = col(E, 0), e1 = col(E, 1), e2 = col(E, 2);
        const Vec3 zero = v3(0, 0, 0);
        const Vec3 blocks[3][3] = {{zero, -e2, e1}, {e2, zero, -e0}, {-e1, e0, zero}};
        const Vec3 tb0 = v3(p[7], p[8], p[9]), tb1 = v3(p[10], p[11], p[12]);
        PL_UNROLL
        for (int cb = 0; cb < 3; ++cb) {
            PL_UNROLL
            for (int k = 0; k < 3; ++k) {
                const Vec3 v = blocks[cb][k];
                c.D[(3 * cb + 0) * 7 + k] = v.x;
                c.D[(3 * cb + 1) * 7 + k] = v.y;
                c.D[(3 * cb + 2) * 7 + k] = v.z;
            }
            const Vec3 a = cross(tb0, col(R, cb)), b = cross(tb1, col(R, cb));
            c.D[(3 * cb + 0) * 7 + 3] = a.x, c.D[(3 * cb + 1) * 7 + 3] = a.y, c.D[(3 * cb + 2) * 7 + 3] = a.z;
            c.D[(3 * cb + 0) * 7 + 4] = b.x, c.D[(3 * cb + 1) * 7 + 4] = b.y, c.D[(3 * cb + 2) * 7 + 4] = b.z;
        }
    }
    PL_HD static double residual(const RefineCtx &c, double a0, double a1, double b0, double b1) {
        return sampson_residual(c.M, a0, a1, b0, b1);
    }
    PL_HD static double jacobian(const RefineCtx &c, double a0, double a1, double b0, double b1, double *J) {
        double dF[9];
        const double r = sampson_residual_grad(c.M, a0, a1, b0, b1, dF);
        PL_UNROLL
        for (int k = 0; k < 5; ++k) {
            double s = 0;
            PL_UNROLL
            for (int m = 0; m < 9; ++m)
                s += dF[m] * c.D[m * 7 + k];
            J[k] = s;
        }
        return r;
    }
    PL_HD static void step(const double *__restrict__ p, const RefineCtx &, const double *__restrict__ dp, double *__restrict__ out) { // (cur, the LM step, trial: never the same storage)
        Quat q;
        q.w = p[0], q.x = p[1], q.y = p[2], q.z = p[3];
        const Quat qn = quat_step_post(q, v3(dp[0], dp[1], dp[2]));
        PL_UNROLL
        for (int i = 0; i < kParamDoubles; ++i)
            out[i] = p[i];
        out[0] = qn.w, out[1] = qn.x, out[2] = qn.y, out[3] = qn.z;
        out[4] = p[4] + (p[7] * dp[3] + p[10] * dp[4]);
        out[5] = p[5] + (p[8] * dp[3] + p[11] * dp[4]);
        out[6] = p[6] + (p[9] * dp[3] + p[12] * dp[4]);
    }
};

// ---- homography ----
template <> struct Refiner<EST_HOM> {
    static constexpr int K = 8;
    PL_HD static void prepare(const double *p, RefineCtx &c) {
        const double *H = p;
        PL_UNROLL
        for (int i = 0; i < 9; ++i)
            c.M[i] = H[i];
        c.G[0] = H[4] * H[8] - H[5] * H[7];
        c.G[1] = H[2] * H[7] - H[1] * H[8];
        c.G[2] = H[1] * H[5] - H[2] * H[4];
        c.G[3] = H[5] * H[6] - H[3] * H[8];
        c.G[4] = H[0] * H[8] - H[2] * H[6];
        c.G[5] = H[2] * H[3] - H[0] * H[5];
        c.G[6] = H[3] * H[7] - H[4] * H[6];
        c.G[7] = H[1] * H[6] - H[0] * H[7];
        c.G[8] = H[0] * H[4] - H[1] * H[3];
    }
    PL_HD static void transfer(const double *H, double a0, double a1, double &z0, double &z1, double &inv) {
        const double h0 = H[0] * a0 + H[1] * a1 + H[2];
        const double h1 = H[3] * a0 + H[4] * a1 + H[5];
        inv = 1.0 / (H[6] * a0 + H[7] * a1 + H[8]);
        z0 = h0 * inv;
        z1 = h1 * inv;
    }
    // forward residual (f0,f1) and backward residual (g0,g1)
    PL_HD static void residual(const RefineCtx &c, double a0, double a1, double b0, double b1, double &f0, double &f1,
                               double &g0, double &g1) {
        double z0, z1, inv;
        transfer(c.M, a0, a1, z0, z1, inv);
        f0 = z0 - b0, f1 = z1 - b1;
        transfer(c.G, b0, b1, z0, z1, inv);
        g0 = z0 - a0, g1 = z1 - a1;
    }
    PL_HD static void jacobian(const RefineCtx &c, double a0, double a1, double b0, double b1, double &f0, double &f1,
                               double *Jf /*2x8*/, double &g0, double &g1, double *Jb /*2x8*/) {
        const double *H = c.M;
        double z0, z1, inv;
        transfer(H, a0, a1, z0, z1, inv);
        f0 = z0 - b0, f1 = z1 - b1;
        const double jf[16] = {a0, 0.0, -a0 * z0, a1, 0.0, -a1 * z0, 1.0, 0.0, 0.0, a0, -a0 * z1, 0.0, a1, -a1 * z1, 0.0, 1.0};
        PL_UNROLL
        for (int i = 0; i < 16; ++i)
            Jf[i] = jf[i] * inv;
        double y0, y1, ginv;
        transfer(c.G, b0, b1, y0, y1, ginv);
        g0 = y0 - a0, g1 = y1 - a1;
        const double H00 = H[0], H01 = H[1], H02 = H[2], H10 = H[3], H11 = H[4], H12 = H[5], H20 = H[6], H21 = H[7],
                     H22 = H[8];
        const double y0b1 = y0 * b1, y0b0 = y0 * b0, y1b1 = y1 * b1, y1b0 = y1 * b0;
        const double jb[16] = {H21 * y0b1 - H11 * y0,
                               H01 * y0 - H21 * y0b0,
                               H11 * y0b0 - H01 * y0b1,
                               H12 - H22 * b1 + H10 * y0 - H20 * y0b1,
                               H22 * b0 - H02 - H00 * y0 + H20 * y0b0,
                               H02 * b1 - H12 * b0 + H00 * y0b1 - H10 * y0b0,
                               H21 * b1 - H11,
                               H01 - H21 * b0,
                               H22 * b1 - H12 - H11 * y1 + H21 * y1b1,
                               H02 - H22 * b0 + H01 * y1 - H21 * y1b0,
                               H12 * b0 - H02 * b1 - H01 * y1b1 + H11 * y1b0,
                               H10 * y1 - H20 * y1b1,
                               H20 * y1b0 - H00 * y1,
                               H00 * y1b1 - H10 * y1b0,
                               H10 - H20 * b1,
                               H20 * b0 - H00};
        PL_UNROLL
        for (int i = 0; i < 16; ++i)
            Jb[i] = jb[i] * ginv;
    }
    PL_HD static void step(const double *__restrict__ p, const RefineCtx &, const double *__restrict__ dp, double *__restrict__ out) { // (cur, the LM step, trial: never the same storage)
        PL_UNROLL
        for (int i = 0; i < kParamDoubles; ++i)
            out[i] = p[i];
        // parameters = first 8 entries of column-major H: e -> (row e%3, col e/3)
        PL_UNROLL
        for (int e = 0; e < 8; ++e)
            out[3 * (e % 3) + (e / 3)] = p[3 * (e % 3) + (e / 3)] + dp[e];
    }
};

// ---- fundamental matrix (factorised) ----
PL_HD void factorized_F(const double *p, double *F) { // optim_utils.h:73-77
    Quat qU, qV;
    qU.w = p[0], qU.x = p[1], qU.y = p[2], qU.z = p[3];
    qV.w = p[4], qV.x = p[5], qV.y = p[6], qV.z = p[7];
    const Mat3 U = quat_to_rotmat(qU), V = quat_to_rotmat(qV);
    const double sigma = p[8];
    PL_UNROLL
    for (int i = 0; i < 3; ++i)
        PL_UNROLL
        for (int j = 0; j < 3; ++j)
            F[3 * i + j] = U.m[3 * i] * V.m[3 * j] + (sigma * U.m[3 * i + 1]) * V.m[3 * j + 1];
}
template <> struct Refiner<EST_FUND> {
    static constexpr int K = 7;
    PL_HD static void prepare(const double *p, RefineCtx &c) {
        factorized_F(p, c.M);
        Quat qU, qV;
        qU.w = p[0], qU.x = p[1], qU.y = p[2], qU.z = p[3];
        qV.w = p[4], qV.x = p[5], qV.y = p[6], qV.z = p[7];
        const Mat3 U = quat_to_rotmat(qU), V = quat_to_rotmat(qV);
        Mat3 F;
        PL_UNROLL
        for (int i = 0; i < 9; ++i)
            F.m[i] = c.M[i];
        const Vec3 axes[3] = {v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
        PL_UNROLL
        for (int cb = 0; cb < 3; ++cb) {
            const Vec3 f = col(F, cb);
            PL_UNROLL
            for (int k = 0; k < 3; ++k) {
                const Vec3 d = cross(axes[k], f);
                c.D[(3 * cb + 0) * 7 + k] = d.x, c.D[(3 * cb + 1) * 7 + k] = d.y, c.D[(3 * cb + 2) * 7 + k] = d.z;
            }
        }
        PL_UNROLL
        for (int r = 0; r < 3; ++r) {
            const Vec3 f = row(F, r);
            PL_UNROLL
            for (int k = 0; k < 3; ++k) {
                const Vec3 d = cross(axes[k], f);
                c.D[(0 + r) * 7 + 3 + k] = d.x, c.D[(3 + r) * 7 + 3 + k] = d.y, c.D[(6 + r) * 7 + 3 + k] = d.z;
            }
        }
        PL_UNROLL
        for (int j = 0; j < 3; ++j)
            PL_UNROLL
            for (int i = 0; i < 3; ++i)
                c.D[(3 * j + i) * 7 + 6] = U.m[3 * i + 1] * V.m[3 * j + 1];
    }
    PL_HD static double residual(const RefineCtx &c, double a0, double a1, double b0, double b1) {
        return sampson_residual(c.M, a0, a1, b0, b1);
    }
    PL_HD static double jacobian(const RefineCtx &c, double a0, double a1, double b0, double b1, double *J) {
        double dF[9];
        const double r = sampson_residual_grad(c.M, a0, a1, b0, b1, dF);
        PL_UNROLL
        for (int k = 0; k < 7; ++k) {
            double s = 0;
            PL_UNROLL
            for (int m = 0; m < 9; ++m)
                s += dF[m] * c.D[m * 7 + k];
            J[k] = s;
        }
        return r;
    }
    PL_HD static void step(const double *__restrict__ p, const RefineCtx &, const double *__restrict__ dp, double *__restrict__ out) { // (cur, the LM step, trial: never the same storage)
        Quat qU, qV;
        qU.w = p[0], qU.x = p[1], qU.y = p[2], qU.z = p[3];
        qV.w = p[4], qV.x = p[5], qV.y = p[6], qV.z = p[7];
        const Quat u = quat_step_pre(qU, v3(dp[0], dp[1], dp[2]));
        const Quat v = quat_step_pre(qV, v3(dp[3], dp[4], dp[5]));
        PL_UNROLL
        for (int i = 0; i < kParamDoubles; ++i)
            out[i] = p[i];
        out[0] = u.w, out[1] = u.x, out[2] = u.y, out[3] = u.z;
        out[4] = v.w, out[5] = v.x, out[6] = v.y, out[7] = v.z;
        out[8] = p[8] + dp[6];
    }
};

// Record (what the scorers consume) of a refined parameter block: pose (q, t) for the absolute / relative problems,
// H row-major, or the Bartoli-Sturm factorisation of F (optim_utils.h:73-77).  est: pl_score.h Estimator.
PL_HD void record_from_lm_params(int est, const double *params, double *rec) {
    if (est == 0 || est == 1) {
        Quat q;
        q.w = params[0], q.x = params[1], q.y = params[2], q.z = params[3];
        store_pose_model_q(rec, q, v3(params[4], params[5], params[6]), est == 1);
    } else if (est == 3) {
        Mat3 H;
        for (int i = 0; i < 9; ++i)
            H.m[i] = params[i];
        store_matrix_model(rec, H);
    } else {
        Mat3 F;
        factorized_F(params, F.m);
        store_matrix_model(rec, F);
    }
}

} // namespace pl
