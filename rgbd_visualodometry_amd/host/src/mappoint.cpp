// mappoint.cpp -- reference src/mappoint.cpp:17-49, include/myslam/mappoint.h:59-64.
#include "myslam/mappoint.h"

#include "myslam/mapmanager.h"

namespace myslam {
std::atomic<size_t> Mappoint::nextId_{0};      // shared by every VO system of the process: ids stay unique across threads

Mappoint::Ptr Mappoint::CreateMappoint(const Vector3d position, const Descriptor& descriptor) {
    return Mappoint::Ptr(new Mappoint(nextId_.fetch_add(1) + 1, position, descriptor));
}

Mappoint::Mappoint(const size_t id, const Vector3d position, const Descriptor& descriptor)
    : pos_(position), outlier_(false), triangulated_(false), optimized_(false), descriptor_(descriptor), id_(id), norm_(Vector3d::Zero()) {}

void Mappoint::AddObservedByKeyframe(const size_t keyframeId, const Point2f posInPixel, const Vector3d cameraCenter, Frame* keyframe) {
    std::unique_lock<std::mutex> lock(obsLock_);
    obsList_.push_back(Observation{keyframeId, posInPixel, keyframe});
    norm_ = (norm_ + (pos_ - cameraCenter).normalized()).normalized();      // running mean viewing direction
    lock.unlock();
    MarkDirty();
}

void Mappoint::RemoveObservedByKeyframe(const size_t keyframeId) {
    std::unique_lock<std::mutex> lock(obsLock_);
    for (size_t i = 0; i < obsList_.size(); ++i) if (obsList_[i].keyframeId == keyframeId) { obsList_.erase(obsList_.begin() + i); break; }
    if (obsList_.empty()) { outlier_ = true; lock.unlock(); MarkDirty(); }  // no observation left
}

void Mappoint::MarkDirty() {
    if (dirty_ || slot_ < 0) return;            // not yet inserted: InsertMappoint queues it
    dirty_ = true;
    MapManager::GetInstance().NoteDirty(this);
}
}  // namespace myslam
