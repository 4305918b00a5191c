// ORACLE -- test infrastructure, NOT the product.  CPU restatement used only as the checker
// (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Parity unpinned at the
// third-party (OpenCV 3.1 / g2o / Sophus) boundary: see DESIGN.md "Oracle".
//
// o_math.h: small fixed-size linear algebra + SE(3) with the Sophus conventions the reference
// uses (tangent = [translation(3), rotation(3)], include/myslam/g2o_types.h:46,:56-60;
// SE3::log used at src/frontend.cpp:343-344,:355-356).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double& operator[](int i) { return (&x)[i]; }
    double operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { double n = norm(a); return {a.x / n, a.y / n, a.z / n}; }

struct M3 {
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
    double& operator()(int r, int c) { return m[r * 3 + c]; }
    double operator()(int r, int c) const { return m[r * 3 + c]; }
    static M3 zero() { M3 a; for (double& v : a.m) v = 0; return a; }
};
inline M3 operator*(const M3& a, const M3& b) {
    M3 c = M3::zero();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) c(i, j) += a(i, k) * b(k, j);
    return c;
}
inline V3 operator*(const M3& a, V3 v) {
    return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
            a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z};
}
inline M3 transpose(const M3& a) { M3 t; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t(i, j) = a(j, i); return t; }
inline M3 hat(V3 w) { M3 h = M3::zero(); h(0, 1) = -w.z; h(0, 2) = w.y; h(1, 0) = w.z; h(1, 2) = -w.x; h(2, 0) = -w.y; h(2, 1) = w.x; return h; }

inline M3 so3_exp(V3 w) {
    double th2 = dot(w, w), th = std::sqrt(th2), A, B;
    if (th < 1e-8) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; }
    else { A = std::sin(th) / th; B = (1.0 - std::cos(th)) / th2; }
    M3 W = hat(w), W2 = W * W, R;
    for (int i = 0; i < 9; ++i) R.m[i] = (i % 4 == 0 ? 1.0 : 0.0) + A * W.m[i] + B * W2.m[i];
    return R;
}

// rotation matrix -> unit quaternion (x,y,z,w), w >= 0 not enforced
inline void rot_to_quat(const M3& R, double q[4]) {
    double tr = R(0, 0) + R(1, 1) + R(2, 2);
    if (tr > 0) { double s = std::sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R(2, 1) - R(1, 2)) / s; q[1] = (R(0, 2) - R(2, 0)) / s; q[2] = (R(1, 0) - R(0, 1)) / s; }
    else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { double s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2; q[3] = (R(2, 1) - R(1, 2)) / s; q[0] = 0.25 * s; q[1] = (R(0, 1) + R(1, 0)) / s; q[2] = (R(0, 2) + R(2, 0)) / s; }
    else if (R(1, 1) > R(2, 2)) { double s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2; q[3] = (R(0, 2) - R(2, 0)) / s; q[0] = (R(0, 1) + R(1, 0)) / s; q[1] = 0.25 * s; q[2] = (R(1, 2) + R(2, 1)) / s; }
    else { double s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2; q[3] = (R(1, 0) - R(0, 1)) / s; q[0] = (R(0, 2) + R(2, 0)) / s; q[1] = (R(1, 2) + R(2, 1)) / s; q[2] = 0.25 * s; }
}

inline V3 so3_log(const M3& R) {
    double q[4]; rot_to_quat(R, q);
    if (q[3] < 0) for (double& v : q) v = -v;
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), k;
    if (n < 1e-10) k = 2.0 / q[3] - 2.0 * n * n / (3.0 * q[3] * q[3] * q[3]);
    else k = 2.0 * std::atan2(n, q[3]) / n;
    return {k * q[0], k * q[1], k * q[2]};
}

struct SE3 {
    M3 R; V3 t;
    SE3() {}
    SE3(const M3& r, V3 tt) : R(r), t(tt) {}
    V3 operator*(V3 p) const { return R * p + t; }
    SE3 operator*(const SE3& o) const { return SE3(R * o.R, R * o.t + t); }
    SE3 inverse() const { M3 Rt = transpose(R); return SE3(Rt, -1.0 * (Rt * t)); }
    // tangent order [upsilon(3), omega(3)] (Sophus / g2o_types.h:46)
    static SE3 exp(const double d[6]) {
        V3 u(d[0], d[1], d[2]), w(d[3], d[4], d[5]);
        double th2 = dot(w, w), th = std::sqrt(th2), B, C;
        if (th < 1e-8) { B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
        else { B = (1.0 - std::cos(th)) / th2; C = (th - std::sin(th)) / (th2 * th); }
        M3 W = hat(w), W2 = W * W, V;
        for (int i = 0; i < 9; ++i) V.m[i] = (i % 4 == 0 ? 1.0 : 0.0) + B * W.m[i] + C * W2.m[i];
        return SE3(so3_exp(w), V * u);
    }
    void log(double d[6]) const {
        V3 w = so3_log(R);
        double th2 = dot(w, w), th = std::sqrt(th2), D;
        if (th < 1e-8) D = 1.0 / 12.0 + th2 / 720.0;
        else D = (1.0 - th * std::cos(0.5 * th) / (2.0 * std::sin(0.5 * th))) / th2;
        M3 W = hat(w), W2 = W * W, Vi;
        for (int i = 0; i < 9; ++i) Vi.m[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * W.m[i] + D * W2.m[i];
        V3 u = Vi * t;
        d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = w.x; d[4] = w.y; d[5] = w.z;
    }
    void to12(double o[12]) const { std::memcpy(o, R.m, 72); o[9] = t.x; o[10] = t.y; o[11] = t.z; }
    static SE3 from12(const double o[12]) { SE3 T; std::memcpy(T.R.m, o, 72); T.t = V3(o[9], o[10], o[11]); return T; }
};

// Cholesky solve of an n x n SPD system stored dense row-major (n <= 256). Returns false if not PD.
inline bool chol_solve(int n, double* A, double* b) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
    return true;
}

}  // namespace orc
