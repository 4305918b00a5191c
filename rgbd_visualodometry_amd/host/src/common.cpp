// common.cpp -- SE3 exponential/logarithm (Sophus conventions: tangent = [translation, rotation],
// as the reference relies on at include/myslam/g2o_types.h:46,:59 and src/frontend.cpp:343-358).
#include "myslam/common_include.h"

namespace myslam {

static Matrix3d skew(const Vector3d& w) {
    Matrix3d S;
    S.m[0] = 0; S.m[1] = -w[2]; S.m[2] = w[1];
    S.m[3] = w[2]; S.m[4] = 0; S.m[5] = -w[0];
    S.m[6] = -w[1]; S.m[7] = w[0]; S.m[8] = 0;
    return S;
}

static Matrix3d lincomb(double a, const Matrix3d& A, double b, const Matrix3d& B) {   // I + aA + bB
    Matrix3d R;
    for (int i = 0; i < 9; ++i) R.m[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * A.m[i] + b * B.m[i];
    return R;
}

SE3 SE3::exp(const Vector6d& d) {
    const Vector3d u(d[0], d[1], d[2]), w(d[3], d[4], d[5]);
    const double t2 = w.dot(w), t = std::sqrt(t2);
    double a, b, c;                         // sin t / t, (1-cos t)/t^2, (t-sin t)/t^3
    if (t < 1e-8) { a = 1 - t2 / 6; b = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; }
    else { a = std::sin(t) / t; b = (1 - std::cos(t)) / t2; c = (t - std::sin(t)) / (t2 * t); }
    const Matrix3d W = skew(w), W2 = W * W;
    return SE3(lincomb(a, W, b, W2), lincomb(b, W, c, W2) * u);
}

void SE3::quaternion(double q[4]) const {
    const Matrix3d& R = R_;
    const double tr = R(0, 0) + R(1, 1) + R(2, 2);
    if (tr > 0) { double s = 2 * std::sqrt(tr + 1); q[3] = s / 4; q[0] = (R(2, 1) - R(1, 2)) / s; q[1] = (R(0, 2) - R(2, 0)) / s; q[2] = (R(1, 0) - R(0, 1)) / s; }
    else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { double s = 2 * std::sqrt(1 + R(0, 0) - R(1, 1) - R(2, 2)); q[3] = (R(2, 1) - R(1, 2)) / s; q[0] = s / 4; q[1] = (R(0, 1) + R(1, 0)) / s; q[2] = (R(0, 2) + R(2, 0)) / s; }
    else if (R(1, 1) > R(2, 2)) { double s = 2 * std::sqrt(1 + R(1, 1) - R(0, 0) - R(2, 2)); q[3] = (R(0, 2) - R(2, 0)) / s; q[0] = (R(0, 1) + R(1, 0)) / s; q[1] = s / 4; q[2] = (R(1, 2) + R(2, 1)) / s; }
    else { double s = 2 * std::sqrt(1 + R(2, 2) - R(0, 0) - R(1, 1)); q[3] = (R(1, 0) - R(0, 1)) / s; q[0] = (R(0, 2) + R(2, 0)) / s; q[1] = (R(1, 2) + R(2, 1)) / s; q[2] = s / 4; }
}

Vector6d SE3::log() const {
    double q[4];
    quaternion(q);
    if (q[3] < 0) for (double& v : q) v = -v;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double k = n < 1e-10 ? 2.0 / q[3] - 2.0 * n * n / (3.0 * q[3] * q[3] * q[3]) : 2.0 * std::atan2(n, q[3]) / n;
    const Vector3d w(k * q[0], k * q[1], k * q[2]);
    const double t2 = w.dot(w), t = std::sqrt(t2);
    const double dcoef = t < 1e-8 ? 1.0 / 12 + t2 / 720 : (1 - t * std::cos(t / 2) / (2 * std::sin(t / 2))) / t2;
    const Matrix3d W = skew(w), W2 = W * W;
    const Vector3d u = lincomb(-0.5, W, dcoef, W2) * t_;
    return Vector6d{u[0], u[1], u[2], w[0], w[1], w[2]};
}

Image Image::clone(int bpp) const {
    Image o = *this;
    if (on_device || !data) return o;
    o.owned = std::make_shared<std::vector<uint8_t>>((size_t)rows * cols * bpp);
    for (int r = 0; r < rows; ++r) std::memcpy(o.owned->data() + (size_t)r * cols * bpp, (const uint8_t*)data + (size_t)r * stride, (size_t)cols * bpp);
    o.data = o.owned->data(); o.stride = cols * bpp;
    return o;
}

}  // namespace myslam
