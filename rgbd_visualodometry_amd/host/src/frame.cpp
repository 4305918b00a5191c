// frame.cpp -- reference src/frame.cpp:18-171.
#include "myslam/frame.h"

#include "myslam/mapmanager.h"

namespace myslam {
size_t Frame::factoryId_ = 0;

Frame::Ptr Frame::CreateFrame(const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth) {
    return Frame::Ptr(new Frame(++factoryId_, timestamp, camera, color.clone(3), depth.clone(2)));
}

Frame::Frame(const size_t id, const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth)
    : timestamp_(timestamp), camera_(camera), color_(color), depth_(depth), id_(id), T_c_w_(SE3()) {}

// The raw sample (with the 4-neighbour fallback of frame.cpp:52-63) is taken on the device next to the
// keypoint (vo_keypoint::depth_raw); only the metric conversion is left.
double Frame::GetDepth(const KeyPoint& kp) {
    if (kp.depth_raw != 0) return double(kp.depth_raw) / camera_->GetDepthScale();
    return -1.0;
}

bool Frame::IsCouldObserveMappoint(const Mappoint::Ptr& mpt) {
    Vector3d posInCam = camera_->World2Camera(mpt->GetPosition(), T_c_w_);
    if (posInCam[2] < 0) return false;
    Vector2d px = camera_->Camera2Pixel(posInCam);
    if (px.x < 0 || px.x >= color_.cols || px.y < 0 || px.y >= color_.rows) return false;
    Vector3d direction = (mpt->GetPosition() - GetCamCenter()).normalized();
    double angle = std::acos(direction.dot(mpt->GetNormDirection()));
    return !(angle > M_PI / 6);
}

void Frame::AddObservedMappoint(const size_t mappointId, const Point2f pixelPos) {
    auto mappoint = MapManager::GetInstance().GetMappoint(mappointId);
    assert(mappoint != nullptr);
    AddObservedMappoint(mappoint.get(), pixelPos);
}

void Frame::AddObservedMappoint(Mappoint* mappoint, const Point2f pixelPos) {
    std::unique_lock<std::mutex> lck(observationMutex_);
    observed_.push_back(ObservedEntry{mappoint->GetId(), mappoint, true});
    mappoint->AddObservedByKeyframe(id_, pixelPos, GetCamCenter(), this);
    for (const Mappoint::Observation& o : mappoint->ObservationList()) {     // no copy of the observation map
        Frame* otherKF = o.keyframe;
        if (otherKF == this) continue;
        assert(otherKF != nullptr);
        if (covisBatch_) { if (otherKF->covisAcc_++ == 0) covisTouched_.push_back(otherKF); continue; }
        int w = ++allCovisibleKeyframeIdToWeight_[o.keyframeId];
        if (w >= 15) activeCovisibleKeyframes_.insert(o.keyframeId);
        otherKF->UpdateCovisibleKeyframeWeight(id_, w);
    }
}

void Frame::EndCovisibilityBatch() {
    std::unique_lock<std::mutex> lck(observationMutex_);
    covisBatch_ = false;
    for (Frame* otherKF : covisTouched_) {
        const int w = (allCovisibleKeyframeIdToWeight_[otherKF->id_] += otherKF->covisAcc_);
        otherKF->covisAcc_ = 0;
        if (w >= 15) activeCovisibleKeyframes_.insert(otherKF->id_);
        otherKF->UpdateCovisibleKeyframeWeight(id_, w);
    }
    covisTouched_.clear();
}

void Frame::RemoveObservedMappoint(const size_t mappointId) {
    std::unique_lock<std::mutex> lck(observationMutex_);
    Mappoint* mappoint = nullptr;
    for (size_t i = observed_.size(); i-- > 0;)
        if (observed_[i].id == mappointId && observed_[i].alive) { observed_[i].alive = false; mappoint = observed_[i].mappoint; break; }
    assert(mappoint != nullptr);
    if (mappoint == nullptr) return;
    mappoint->RemoveObservedByKeyframe(id_);
    for (const Mappoint::Observation& o : mappoint->ObservationList()) {
        const size_t other = o.keyframeId;
        if (other == id_) continue;
        Frame* otherKF = o.keyframe;
        assert(otherKF != nullptr);
        int w = --allCovisibleKeyframeIdToWeight_[other];
        if (w == 0) allCovisibleKeyframeIdToWeight_.erase(other);
        else if (w < 15) activeCovisibleKeyframes_.erase(other);
        otherKF->UpdateCovisibleKeyframeWeight(id_, w);
    }
}

void Frame::UpdateCovisibleKeyframeWeight(const size_t id, const int weight) {
    std::unique_lock<std::mutex> lck(observationMutex_);
    if (weight == 0) { allCovisibleKeyframeIdToWeight_.erase(id); activeCovisibleKeyframes_.erase(id); }
    else if (weight >= 15) { allCovisibleKeyframeIdToWeight_[id] = weight; activeCovisibleKeyframes_.insert(id); }
    else { allCovisibleKeyframeIdToWeight_[id] = weight; activeCovisibleKeyframes_.erase(id); }
}
}  // namespace myslam
