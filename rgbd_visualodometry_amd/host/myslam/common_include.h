// common_include.h -- value types of the libmyslam-compatible host layer.
// The reference builds on Eigen / Sophus / OpenCV value types (include/myslam/common_include.h:26-47);
// none of those exist here, so this header supplies minimal own types with the same roles and
// the member names the reference's code uses (SE3::rotationMatrix/translation/inverse/exp/log,
// KeyPoint::pt, Point2f, Mat-like Image).
#ifndef MYSLAM_COMMON_INCLUDE_H
#define MYSLAM_COMMON_INCLUDE_H

#include <array>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "vo_hip.h"

namespace myslam {

struct Vector2d { double x = 0, y = 0; Vector2d() {} Vector2d(double a, double b) : x(a), y(b) {} };

struct Vector3d {
    double v[3] = {0, 0, 0};
    Vector3d() {}
    Vector3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    Vector3d operator+(const Vector3d& o) const { return {v[0] + o[0], v[1] + o[1], v[2] + o[2]}; }
    Vector3d operator-(const Vector3d& o) const { return {v[0] - o[0], v[1] - o[1], v[2] - o[2]}; }
    Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    double dot(const Vector3d& o) const { return v[0] * o[0] + v[1] * o[1] + v[2] * o[2]; }
    double norm() const { return std::sqrt(dot(*this)); }
    Vector3d normalized() const { double n = norm(); return {v[0] / n, v[1] / n, v[2] / n}; }
    static Vector3d Zero() { return {}; }
};
typedef Vector3d Vec3;

struct Matrix3d {
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
    Vector3d operator*(const Vector3d& p) const {
        return {m[0] * p[0] + m[1] * p[1] + m[2] * p[2], m[3] * p[0] + m[4] * p[1] + m[5] * p[2], m[6] * p[0] + m[7] * p[1] + m[8] * p[2]};
    }
    Matrix3d operator*(const Matrix3d& o) const {
        Matrix3d r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += m[3 * i + k] * o.m[3 * k + j]; r.m[3 * i + j] = s; }
        return r;
    }
    Matrix3d transpose() const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = m[3 * j + i]; return r; }
};

typedef std::array<double, 6> Vector6d;

// Rigid transform with Sophus::SE3d's interface subset; tangent = [translation, rotation].
class SE3 {
public:
    SE3() {}
    SE3(const Matrix3d& R, const Vector3d& t) : R_(R), t_(t) {}
    const Matrix3d& rotationMatrix() const { return R_; }
    const Vector3d& translation() const { return t_; }
    Vector3d operator*(const Vector3d& p) const { return R_ * p + t_; }
    SE3 operator*(const SE3& o) const { return SE3(R_ * o.R_, R_ * o.t_ + t_); }
    SE3 inverse() const { Matrix3d Rt = R_.transpose(); return SE3(Rt, (Rt * t_) * -1.0); }
    static SE3 exp(const Vector6d& d);
    Vector6d log() const;
    void matrix3x4(double out[12]) const {   // row-major [R|t]
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out[4 * r + c] = R_(r, c); out[4 * r + 3] = t_[r]; }
    }
    // C-ABI layout: R row-major (9) then t (3)
    void to12(double o[12]) const { std::memcpy(o, R_.m, 72); o[9] = t_[0]; o[10] = t_[1]; o[11] = t_[2]; }
    static SE3 from12(const double o[12]) { SE3 T; std::memcpy(T.R_.m, o, 72); T.t_ = Vector3d(o[9], o[10], o[11]); return T; }
    void quaternion(double q[4]) const;      // (x, y, z, w)
private:
    Matrix3d R_;
    Vector3d t_;
};

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point3f { float x = 0, y = 0, z = 0; };

// cv::KeyPoint's fields + index of the keypoint in its frame and the raw depth sample.
struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
    int index = -1;         // position in the frame's keypoint list (identity for set membership)
    int depth_raw = 0;      // Frame::GetDepth's raw sample (0 = none)
};

// Stand-in for cv::Mat as run_vo uses it: an 8UC3 colour or 16UC1 depth image, host or device resident.
struct Image {
    const void* data = nullptr; int rows = 0, cols = 0, stride = 0; bool on_device = false;
    std::shared_ptr<std::vector<uint8_t>> owned;     // set by clone()
    Image clone(int bytes_per_px) const;
    bool empty() const { return data == nullptr; }
};
typedef Image Mat;

typedef std::array<uint8_t, 32> Descriptor;

}  // namespace myslam

#endif
