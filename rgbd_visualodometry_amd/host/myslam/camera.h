// camera.h -- pinhole RGB-D camera (reference include/myslam/camera.h:38-66, src/camera.cpp:27-86).
#ifndef MYSLAM_CAMERA_H
#define MYSLAM_CAMERA_H
#include "myslam/common_include.h"

namespace myslam {
class Camera {
public:
    typedef std::shared_ptr<Camera> Ptr;
    Camera();                                                     // reads camera.* from Config (float, camera.cpp:29-33)
    Camera(float fx, float fy, float cx, float cy, float depth_scale) : fx_(fx), fy_(fy), cx_(cx), cy_(cy), depthScale_(depth_scale) {}
    float GetFx() const { return fx_; }
    float GetFy() const { return fy_; }
    float GetCx() const { return cx_; }
    float GetCy() const { return cy_; }
    float GetDepthScale() const { return depthScale_; }
    Vector3d World2Camera(const Vector3d& p_w, const SE3& T_c_w) const { return T_c_w * p_w; }
    Vector3d Camera2World(const Vector3d& p_c, const SE3& T_c_w) const { return T_c_w.inverse() * p_c; }
    Vector2d Camera2Pixel(const Vector3d& p_c) const { return Vector2d(fx_ * p_c[0] / p_c[2] + cx_, fy_ * p_c[1] / p_c[2] + cy_); }
    Vector3d Pixel2Camera(const Vector2d& p_p, double depth = 1) const { return Vector3d((p_p.x - cx_) * depth / fx_, (p_p.y - cy_) * depth / fy_, depth); }
    Vector3d Pixel2Camera(const Point2f& p_p, double depth = 1) const { return Pixel2Camera(Vector2d(p_p.x, p_p.y), depth); }
    Vector2d World2Pixel(const Vector3d& p_w, const SE3& T_c_w) const { return Camera2Pixel(World2Camera(p_w, T_c_w)); }
    Vector3d Pixel2World(const Vector2d& p_p, const SE3& T_c_w, double depth = 1) const { return Camera2World(Pixel2Camera(p_p, depth), T_c_w); }
    Vector3d Pixel2World(const KeyPoint& kp, const SE3& T_c_w, double depth = 1) const { return Pixel2World(Vector2d(kp.pt.x, kp.pt.y), T_c_w, depth); }
private:
    float fx_, fy_, cx_, cy_, depthScale_;
};
}  // namespace myslam
#endif
