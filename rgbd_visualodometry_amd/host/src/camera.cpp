// camera.cpp -- intrinsics from the parameter file, stored as float (reference src/camera.cpp:27-39).
#include "myslam/camera.h"

#include "myslam/config.h"

namespace myslam {
Camera::Camera() {
    fx_ = Config::get<float>("camera.fx");
    fy_ = Config::get<float>("camera.fy");
    cx_ = Config::get<float>("camera.cx");
    cy_ = Config::get<float>("camera.cy");
    depthScale_ = Config::get<float>("camera.depth_scale");
}
}  // namespace myslam
