// viewer.h -- GUI hook kept for API compatibility (reference include/myslam/viewer.h, Pangolin
// window).  Out of scope on a headless GPU server: every call is a no-op.
#ifndef MYSLAM_VIEWER_H
#define MYSLAM_VIEWER_H
#include "myslam/frame.h"
#include "myslam/util.h"
namespace myslam {
class Viewer {
public:
    typedef std::shared_ptr<Viewer> Ptr;
    void setCurrentFrame(const Frame::Ptr&, const KeyPointSet&) {}
    void updateDrawingObjects() {}
    void Close() {}
};
}  // namespace myslam
#endif
