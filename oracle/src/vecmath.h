// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Tiny fixed-size linear algebra used by the CPU restatement of the PoseLib hot path.
// The reference leans on Eigen for these primitives (not available in this image); the
// evaluation orders below follow SURVEY.md Appendix A:  dot/products are summed left to
// right, `normalized()` divides by sqrt(squaredNorm) (no reciprocal), 3x3 inverse is
// cofactor / determinant.  Eigen-internal association order cannot be verified here, so
// results agree with real PoseLib at tolerance level (~1e-15 rel.), not bit level.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace orc {

struct V2 {
    double x = 0, y = 0;
};
struct V3 {
    double x = 0, y = 0, z = 0;
    double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V4 {
    double a[4] = {0, 0, 0, 0};
    double &operator[](int i) { return a[i]; }
    double operator[](int i) const { return a[i]; }
};
// Row-major 3x3:  m[r][c]
struct M3 {
    double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    static M3 identity() {
        M3 I;
        I.m[0][0] = I.m[1][1] = I.m[2][2] = 1.0;
        return I;
    }
    V3 col(int c) const { return V3{m[0][c], m[1][c], m[2][c]}; }
    V3 row(int r) const { return V3{m[r][0], m[r][1], m[r][2]}; }
    void set_col(int c, const V3 &v) {
        m[0][c] = v.x;
        m[1][c] = v.y;
        m[2][c] = v.z;
    }
    void set_row(int r, const V3 &v) {
        m[r][0] = v.x;
        m[r][1] = v.y;
        m[r][2] = v.z;
    }
};

inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, double s) { return {a.x * s, a.y * s}; }
inline V2 operator/(V2 a, double s) { return {a.x / s, a.y / s}; }
inline double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
inline double norm(V2 a) { return std::sqrt(dot(a, a)); }

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double sqnorm(V3 a) { return dot(a, a); }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { return a / norm(a); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// [x, y] -> unit bearing [x, y, 1] / |.|   (x.homogeneous().normalized())
inline V3 bearing(V2 p) { return normalized(V3{p.x, p.y, 1.0}); }

inline V3 operator*(const M3 &A, V3 v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline M3 operator*(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
inline M3 operator*(const M3 &A, double s) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[i][j] = A.m[i][j] * s;
    return C;
}
inline M3 transpose(const M3 &A) {
    M3 T;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T.m[i][j] = A.m[j][i];
    return T;
}
inline double det(const M3 &A) {
    return A.m[0][0] * (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) -
           A.m[0][1] * (A.m[1][0] * A.m[2][2] - A.m[1][2] * A.m[2][0]) +
           A.m[0][2] * (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]);
}
// Cofactor / determinant inverse (what Eigen does for fixed 3x3).
inline M3 inverse(const M3 &A) {
    M3 C;
    C.m[0][0] = A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1];
    C.m[0][1] = A.m[0][2] * A.m[2][1] - A.m[0][1] * A.m[2][2];
    C.m[0][2] = A.m[0][1] * A.m[1][2] - A.m[0][2] * A.m[1][1];
    C.m[1][0] = A.m[1][2] * A.m[2][0] - A.m[1][0] * A.m[2][2];
    C.m[1][1] = A.m[0][0] * A.m[2][2] - A.m[0][2] * A.m[2][0];
    C.m[1][2] = A.m[0][2] * A.m[1][0] - A.m[0][0] * A.m[1][2];
    C.m[2][0] = A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0];
    C.m[2][1] = A.m[0][1] * A.m[2][0] - A.m[0][0] * A.m[2][1];
    C.m[2][2] = A.m[0][0] * A.m[1][1] - A.m[0][1] * A.m[1][0];
    const double d = A.m[0][0] * C.m[0][0] + A.m[0][1] * C.m[1][0] + A.m[0][2] * C.m[2][0];
    const double inv = 1.0 / d;
    return C * inv;
}
inline double frob(const M3 &A) {
    double s = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            s += A.m[i][j] * A.m[i][j];
    return std::sqrt(s);
}

// ---------------------------------------------------------------------------------------
// Quaternions, (w,x,y,z) order.  Follows PoseLib/misc/quaternion.h:36-104 which defers to
// Eigen::Quaterniond for R<->q (formulas restated from SURVEY.md Appendix A).
// ---------------------------------------------------------------------------------------
inline M3 quat_to_rotmat(const V4 &q) { // quaternion.h:36-38
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3 R;
    R.m[0][0] = 1.0 - (tyy + tzz);
    R.m[0][1] = txy - twz;
    R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;
    R.m[1][1] = 1.0 - (txx + tzz);
    R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;
    R.m[2][1] = tyz + twx;
    R.m[2][2] = 1.0 - (txx + tyy);
    return R;
}
inline V4 rotmat_to_quat(const M3 &R) { // quaternion.h:45-51
    V4 q;                                // q[0]=w, q[1..3]=xyz
    double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R.m[2][1] - R.m[1][2]) * t;
        q[2] = (R.m[0][2] - R.m[2][0]) * t;
        q[3] = (R.m[1][0] - R.m[0][1]) * t;
    } else {
        int i = 0;
        if (R.m[1][1] > R.m[0][0])
            i = 1;
        if (R.m[2][2] > R.m[i][i])
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R.m[k][j] - R.m[j][k]) * t;
        q[1 + j] = (R.m[j][i] + R.m[i][j]) * t;
        q[1 + k] = (R.m[k][i] + R.m[i][k]) * t;
    }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; ++c)
        q[c] = q[c] / n;
    return q;
}
inline V4 quat_mul(const V4 &a, const V4 &b) { // quaternion.h:52-59
    V4 r;
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] - a[1] * b[3] + a[3] * b[1];
    r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    return r;
}
inline V3 quat_rotate(const V4 &q, const V3 &p) { // quaternion.h:61-70
    const double q1 = q[0], q2 = q[1], q3 = q[2], q4 = q[3];
    const double a = -p.x * q2 - p.y * q3 - p.z * q4;
    const double b = p.x * q1 - p.y * q4 + p.z * q3;
    const double c = p.y * q1 + p.x * q4 - p.z * q2;
    const double d = p.y * q2 - p.x * q3 + p.z * q1;
    return {b * q1 - a * q2 - c * q4 + d * q3, c * q1 - a * q3 + b * q4 - d * q2, c * q2 - b * q3 - a * q4 + d * q1};
}
inline V4 quat_conj(const V4 &q) {
    V4 r;
    r[0] = q[0];
    r[1] = -q[1];
    r[2] = -q[2];
    r[3] = -q[3];
    return r;
}
inline V4 quat_exp(const V3 &w) { // quaternion.h:73-96
    const double th2 = dot(w, w);
    const double th = std::sqrt(th2);
    double re, im;
    if (th > 1e-6) {
        re = std::cos(0.5 * th);
        im = std::sin(0.5 * th) / th;
    } else {
        const double th4 = th2 * th2;
        re = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
        im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        const double s = std::sqrt(re * re + im * im * th2);
        re /= s;
        im /= s;
    }
    V4 r;
    r[0] = re;
    r[1] = im * w.x;
    r[2] = im * w.y;
    r[3] = im * w.z;
    return r;
}
inline V4 quat_step_pre(const V4 &q, const V3 &w) { return quat_mul(quat_exp(w), q); }
inline V4 quat_step_post(const V4 &q, const V3 &w) { return quat_mul(q, quat_exp(w)); }

// CameraPose{q,t}: PoseLib/camera_pose.h:40-68
struct Pose {
    V4 q;
    V3 t;
    Pose() { q[0] = 1.0; }
    Pose(const V4 &qq, const V3 &tt) : q(qq), t(tt) {}
    Pose(const M3 &R, const V3 &tt) : q(rotmat_to_quat(R)), t(tt) {}
    M3 R() const { return quat_to_rotmat(q); }
    V3 rotate(const V3 &p) const { return quat_rotate(q, p); }
};

} // namespace orc
