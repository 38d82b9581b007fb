// poselib_amd — device math primitives (gfx950).  Everything here is PL_HD so that the same
// source is compiled (a) by hipcc into the kernels and (b) by the test-only host build
// (tests/hostmath) that checks it against the oracle on CPU.  fp64 throughout; compiled with
// -ffp-contract=off so that mul/add sequences round exactly like the reference's SSE2 build.
//
// Reference semantics followed: PoseLib/misc/quaternion.h:36-104 (R<->q through Eigen's
// Quaterniond, real part first), PoseLib/camera_pose.h:40-68.
#pragma once
#include "pl_defs.h"
#include "pl_libm.h"

namespace pl {

// Layout of one hypothesis ("model record") in HBM: 24 doubles = 192 B, pulled into SGPRs by
// wave-uniform scalar loads.
//   fp64 part (kModelDoubles = 16):
//     [0..3]  q (w,x,y,z)        [4..6] t        [7..15] 3x3 matrix, row-major:
//     R(q) for absolute pose, E = [t]x R(q) for relative pose, H or F for the projective models.
//   fp32 shadow (doubles 16..23 viewed as 16 floats), used ONLY by the conservative pre-filter of the
//   scoring kernel (never by a result): f[0..8] = (float)matrix, f[9..11] = (float)t,
//   f[12] = upper bound of max|t_i|.
constexpr int kModelStride = 24;
constexpr int kModelDoubles = 16;
constexpr int kShadowOff = 16;
constexpr int kMatOff = 7;

struct Vec3 {
    double x, y, z;
};

PL_HD Vec3 v3(double x, double y, double z) {
    Vec3 r;
    r.x = x, r.y = y, r.z = z;
    return r;
}
PL_HD Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PL_HD Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PL_HD Vec3 operator-(Vec3 a) { return v3(-a.x, -a.y, -a.z); }
PL_HD Vec3 operator*(Vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
PL_HD Vec3 operator*(double s, Vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PL_HD Vec3 operator/(Vec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
PL_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PL_HD Vec3 cross(Vec3 a, Vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PL_HD Vec3 normalized(Vec3 a) { return a / sqrt(dot(a, a)); }
// 2-D image point -> unit bearing  (x.homogeneous().normalized())
PL_HD Vec3 bearing(double x, double y) { return normalized(v3(x, y, 1.0)); }

// Row-major 3x3
struct Mat3 {
    double m[9];
    PL_HD double &operator()(int r, int c) { return m[3 * r + c]; }
    PL_HD double operator()(int r, int c) const { return m[3 * r + c]; }
};
PL_HD Vec3 col(const Mat3 &A, int c) { return v3(A.m[c], A.m[3 + c], A.m[6 + c]); }
PL_HD Vec3 row(const Mat3 &A, int r) { return v3(A.m[3 * r], A.m[3 * r + 1], A.m[3 * r + 2]); }
PL_HD void set_col(Mat3 &A, int c, Vec3 v) { A.m[c] = v.x, A.m[3 + c] = v.y, A.m[6 + c] = v.z; }
PL_HD void set_row(Mat3 &A, int r, Vec3 v) { A.m[3 * r] = v.x, A.m[3 * r + 1] = v.y, A.m[3 * r + 2] = v.z; }
PL_HD Vec3 mul(const Mat3 &A, Vec3 v) {
    return v3(A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
              A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z);
}
PL_HD Vec3 mul_t(const Mat3 &A, Vec3 v) { // A^T v
    return v3(A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z,
              A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z);
}
PL_HD Mat3 mul(const Mat3 &A, const Mat3 &B) {
    Mat3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
PL_HD double det3(const Mat3 &A) {
    return A.m[0] * (A.m[4] * A.m[8] - A.m[5] * A.m[7]) - A.m[1] * (A.m[3] * A.m[8] - A.m[5] * A.m[6]) +
           A.m[2] * (A.m[3] * A.m[7] - A.m[4] * A.m[6]);
}
PL_HD Mat3 inverse3(const Mat3 &A) { // cofactors / determinant
    Mat3 C;
    C.m[0] = A.m[4] * A.m[8] - A.m[5] * A.m[7];
    C.m[1] = A.m[2] * A.m[7] - A.m[1] * A.m[8];
    C.m[2] = A.m[1] * A.m[5] - A.m[2] * A.m[4];
    C.m[3] = A.m[5] * A.m[6] - A.m[3] * A.m[8];
    C.m[4] = A.m[0] * A.m[8] - A.m[2] * A.m[6];
    C.m[5] = A.m[2] * A.m[3] - A.m[0] * A.m[5];
    C.m[6] = A.m[3] * A.m[7] - A.m[4] * A.m[6];
    C.m[7] = A.m[1] * A.m[6] - A.m[0] * A.m[7];
    C.m[8] = A.m[0] * A.m[4] - A.m[1] * A.m[3];
    const double d = A.m[0] * C.m[0] + A.m[1] * C.m[3] + A.m[2] * C.m[6];
    const double inv = 1.0 / d;
    for (int i = 0; i < 9; ++i)
        C.m[i] = C.m[i] * inv;
    return C;
}

struct Quat {
    double w, x, y, z;
};

PL_HD Mat3 quat_to_rotmat(Quat q) { // quaternion.h:36-38 (Eigen toRotationMatrix)
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    Mat3 R;
    R.m[0] = 1.0 - (tyy + tzz);
    R.m[1] = txy - twz;
    R.m[2] = txz + twy;
    R.m[3] = txy + twz;
    R.m[4] = 1.0 - (txx + tzz);
    R.m[5] = tyz - twx;
    R.m[6] = txz - twy;
    R.m[7] = tyz + twx;
    R.m[8] = 1.0 - (txx + tyy);
    return R;
}

PL_HD Quat rotmat_to_quat(const Mat3 &R) { // quaternion.h:45-51 (Eigen Quaterniond(R), then normalise)
    double q[4];                           // w x y z
    double t = R.m[0] + R.m[4] + R.m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R.m[7] - R.m[5]) * t;
        q[2] = (R.m[2] - R.m[6]) * t;
        q[3] = (R.m[3] - R.m[1]) * t;
    } else {
        // branch-free selection of the largest diagonal entry (ties -> lower index) so that the
        // per-lane code does not index a register array dynamically
        const double d0 = R.m[0], d1 = R.m[4], d2 = R.m[8];
        int i = 0;
        if (d1 > d0)
            i = 1;
        if (d2 > (i == 0 ? d0 : d1))
            i = 2;
        if (i == 0) {
            t = sqrt(d0 - d1 - d2 + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (R.m[7] - R.m[5]) * t;
            q[2] = (R.m[3] + R.m[1]) * t;
            q[3] = (R.m[6] + R.m[2]) * t;
        } else if (i == 1) {
            t = sqrt(d1 - d2 - d0 + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (R.m[2] - R.m[6]) * t;
            q[3] = (R.m[7] + R.m[5]) * t;
            q[1] = (R.m[1] + R.m[3]) * t;
        } else {
            t = sqrt(d2 - d0 - d1 + 1.0);
            q[3] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (R.m[3] - R.m[1]) * t;
            q[1] = (R.m[2] + R.m[6]) * t;
            q[2] = (R.m[5] + R.m[7]) * t;
        }
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    Quat r;
    r.w = q[0] / n, r.x = q[1] / n, r.y = q[2] / n, r.z = q[3] / n;
    return r;
}

PL_HD Quat quat_mul(Quat a, Quat b) { // quaternion.h:52-59
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w - a.x * b.z + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
PL_HD Vec3 quat_rotate(Quat q, Vec3 p) { // quaternion.h:61-70
    const double a = -p.x * q.x - p.y * q.y - p.z * q.z;
    const double b = p.x * q.w - p.y * q.z + p.z * q.y;
    const double c = p.y * q.w + p.x * q.z - p.z * q.x;
    const double d = p.y * q.x - p.x * q.y + p.z * q.w;
    return v3(b * q.w - a * q.x - c * q.z + d * q.y, c * q.w - a * q.y + b * q.z - d * q.x,
              c * q.x - b * q.y - a * q.z + d * q.w);
}
PL_HD Quat quat_exp(Vec3 w) { // quaternion.h:73-96
    const double th2 = dot(w, w);
    const double th = sqrt(th2);
    double re, im;
    if (th > 1e-6) {
        double sn; // (pl_libm.h pl_sincos: cos and sin of one argument are ONE sincos() call in the reference's build)
        pl_sincos(0.5 * th, sn, re);
        im = sn / th;
    } else {
        const double th4 = th2 * th2;
        re = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
        im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        const double s = sqrt(re * re + im * im * th2);
        re /= s;
        im /= s;
    }
    Quat r;
    r.w = re, r.x = im * w.x, r.y = im * w.y, r.z = im * w.z;
    return r;
}
PL_HD Quat quat_step_pre(Quat q, Vec3 w) { return quat_mul(quat_exp(w), q); }
PL_HD Quat quat_step_post(Quat q, Vec3 w) { return quat_mul(q, quat_exp(w)); }

// E = [t]x R      (PoseLib/misc/essential.cc:35-38)
PL_HD Mat3 essential_from_motion(const Mat3 &R, Vec3 t) {
    Mat3 T;
    T.m[0] = 0.0, T.m[1] = -t.z, T.m[2] = t.y;
    T.m[3] = t.z, T.m[4] = 0.0, T.m[5] = -t.x;
    T.m[6] = -t.y, T.m[7] = t.x, T.m[8] = 0.0;
    return mul(T, R);
}

// Two-view cheirality test for unit bearings (essential.cc:40-57).
// the two (scaled) depths and the factor of the depth bound
PL_HD void cheirality_depths(Quat q, Vec3 t, Vec3 x1, Vec3 x2, double &l1, double &l2, double &a) {
    const Vec3 Rx1 = quat_rotate(q, x1);
    a = -dot(Rx1, x2);
    const double b1 = -dot(Rx1, t);
    const double b2 = dot(x2, t);
    l1 = b1 - a * b2;
    l2 = -a * b1 + b2;
}
PL_HD bool check_cheirality(Quat q, Vec3 t, Vec3 x1, Vec3 x2, double min_depth) {
    double l1, l2, a;
    cheirality_depths(q, t, x1, x2, l1, l2, a);
    min_depth = min_depth * (1 - a * a);
    return l1 > min_depth && l2 > min_depth;
}

// fp32 shadow of the record's matrix / translation (see the layout comment above).
// max-abs entry of the 3x3 matrix as fp32 (padded up), or +inf when it lies outside the range in which the fp32
// pre-filters keep their relative accuracy (pl_prefilter.h)
PL_HD float model_scale_f32(const double *M9) {
    double m = 0.0;
    for (int i = 0; i < 9; ++i) {
        const double a = fabs(M9[i]);
        m = a > m ? a : m;
    }
    if (!(m >= 1e-18 && m <= 1e18))
        return __builtin_huge_valf();
    return (float)m * 1.000001f + 1e-30f;
}

PL_HD bool store_shadow(double *rec) { // returns the record's NaN flag
    float *f = reinterpret_cast<float *>(rec + kShadowOff);
    for (int i = 0; i < 9; ++i)
        f[i] = (float)rec[kMatOff + i];
    float tmax = 0.f;
    for (int i = 0; i < 3; ++i) {
        f[9 + i] = (float)rec[4 + i];
        const float a = (float)fabs(rec[4 + i]);
        tmax = a > tmax ? a : tmax;
    }
    f[12] = tmax * 1.000001f + 1e-30f; // rounded-to-nearest conversions, padded upwards
    // f[13] != 0: some entry of t or of the matrix is NaN.  Every residual of the four scores then is NaN (each
    // point's residual involves every entry, and 0 * NaN = NaN), no comparison r2 < thr2 succeeds, and the
    // reference counts zero inliers (utils.cc:37-130): scorers may return (0, 0) for such a model unseen.  The
    // reference's P3P does emit such poses for inconsistent samples (about 13 % at 70 % outliers).
    bool any_nan = false;
    for (int i = 4; i < kModelDoubles; ++i)
        any_nan = any_nan || (rec[i] != rec[i]);
    f[13] = any_nan ? 1.f : 0.f;
    f[14] = model_scale_f32(rec + kMatOff);
    f[15] = 0.f;
    return any_nan;
}

// Write a pose hypothesis (rotation given as matrix from a solver) into a 16-double record:
// R -> q (normalised) -> R(q), exactly the round trip CameraPose(R,t) + pose.R() makes in the
// reference (camera_pose.h:51, utils.cc:40).  `essential` selects E=[t]xR(q) for the matrix slot.
PL_HD bool store_pose_model(double *rec, const Mat3 &Rsolver, Vec3 t, bool essential) {
    const Quat q = rotmat_to_quat(Rsolver);
    const Mat3 Rq = quat_to_rotmat(q);
    rec[0] = q.w, rec[1] = q.x, rec[2] = q.y, rec[3] = q.z;
    rec[4] = t.x, rec[5] = t.y, rec[6] = t.z;
    const Mat3 M = essential ? essential_from_motion(Rq, t) : Rq;
    for (int i = 0; i < 9; ++i)
        rec[kMatOff + i] = M.m[i];
    return store_shadow(rec);
}
PL_HD bool store_pose_model_q(double *rec, Quat q, Vec3 t, bool essential) {
    const Mat3 Rq = quat_to_rotmat(q);
    rec[0] = q.w, rec[1] = q.x, rec[2] = q.y, rec[3] = q.z;
    rec[4] = t.x, rec[5] = t.y, rec[6] = t.z;
    const Mat3 M = essential ? essential_from_motion(Rq, t) : Rq;
    for (int i = 0; i < 9; ++i)
        rec[kMatOff + i] = M.m[i];
    return store_shadow(rec);
}
PL_HD bool store_matrix_model(double *rec, const Mat3 &M) {
    for (int i = 0; i < 7; ++i)
        rec[i] = 0.0;
    for (int i = 0; i < 9; ++i)
        rec[kMatOff + i] = M.m[i];
    return store_shadow(rec);
}

} // namespace pl
