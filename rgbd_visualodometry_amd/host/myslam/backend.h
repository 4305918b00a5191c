// backend.h -- local bundle adjustment over the covisibility graph of a new keyframe
// (reference include/myslam/backend.h:21-37, src/backend.cpp:19-195).  The graph is flattened
// on the host and solved by vo_local_ba on the GPU.  By default the optimisation runs
// synchronously inside OptimizeCovisibleGraphOfKeyframe (deterministic; removes the
// tracker/back-end data race of the reference).
#ifndef MYSLAM_BACKEND_H
#define MYSLAM_BACKEND_H
#include "myslam/camera.h"
#include "myslam/common_include.h"
#include "myslam/frame.h"
#include "myslam/mapmanager.h"

namespace myslam {
class Backend {
public:
    typedef std::shared_ptr<Backend> Ptr;
    Backend(const Camera::Ptr camera);
    void SetContext(vo_ctx* ctx) { ctx_ = ctx; }
    void Stop() {}
    void OptimizeCovisibleGraphOfKeyframe(const Frame::Ptr keyframeCurr);
    struct Stats { int runs = 0, poses = 0, fixed = 0, points = 0, edges = 0, outliers = 0; double ms = 0, ms_build = 0, ms_solve = 0; };
    const Stats& GetStats() const { return stats_; }
private:
    Camera::Ptr camera_;
    Frame::Ptr keyframeCurr_;
    float chi2Threshold_;
    vo_ctx* ctx_ = nullptr;
    Stats stats_;
    void Optimize();
};
}  // namespace myslam
#endif
