// frame.cpp -- reference src/frame.cpp:18-171.
#include "myslam/frame.h"

#include "myslam/mapmanager.h"

namespace myslam {
std::atomic<size_t> Frame::nextId_{0};         // shared by every VO system of the process: ids stay unique across threads

Frame::Ptr Frame::CreateFrame(const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth) {
    return Frame::Ptr(new Frame(nextId_.fetch_add(1) + 1, timestamp, camera, color.clone(3), depth.clone(2)));
}

Frame::Ptr Frame::CreateFrameView(const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth) {
    return Frame::Ptr(new Frame(nextId_.fetch_add(1) + 1, timestamp, camera, color, depth));
}

Frame::Frame(const size_t id, const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth)
    : timestamp_(timestamp), camera_(camera), color_(color), depth_(depth), id_(id), pose_cw_(SE3()) {}

// The raw sample (with the 4-neighbour fallback of frame.cpp:52-63) is taken on the device next to the
// keypoint (vo_keypoint::depth_raw); only the metric conversion is left.
double Frame::GetDepth(const KeyPoint& kp) {
    if (kp.depth_raw != 0) return double(kp.depth_raw) / camera_->GetDepthScale();
    return -1.0;
}

bool Frame::IsCouldObserveMappoint(const Mappoint::Ptr& mpt) {
    Vector3d posInCam = camera_->World2Camera(mpt->GetPosition(), pose_cw_);
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
    std::unique_lock<std::mutex> guard(obsLock_);
    observed_.push_back(ObservedEntry{mappoint->GetId(), mappoint, true});
    mappoint->AddObservedByKeyframe(id_, pixelPos, GetCamCenter(), this);
    // every other keyframe that already sees this point gains one shared point with this frame (frame.cpp:104-119)
    for (const Mappoint::Observation& seenBy : mappoint->ObservationList()) {
        Frame* partner = seenBy.keyframe;
        if (partner == this) continue;
        assert(partner != nullptr);
        if (covisBatch_) {                                  // counted now, written once in EndCovisibilityBatch
            if (partner->covisPending_++ == 0) covisTouched_.push_back(partner);
            continue;
        }
        partner->UpdateCovisibleKeyframeWeight(id_, covis_.add(seenBy.keyframeId, +1));
    }
}

void Frame::EndCovisibilityBatch() {
    std::unique_lock<std::mutex> guard(obsLock_);
    covisBatch_ = false;
    for (Frame* partner : covisTouched_) {
        const int gained = partner->covisPending_;
        partner->covisPending_ = 0;
        partner->UpdateCovisibleKeyframeWeight(id_, covis_.add(partner->id_, gained));
    }
    covisTouched_.clear();
}

void Frame::RemoveObservedMappoint(const size_t mappointId) {
    std::unique_lock<std::mutex> guard(obsLock_);
    Mappoint* mappoint = nullptr;
    for (size_t i = observed_.size(); i-- > 0;)
        if (observed_[i].id == mappointId && observed_[i].alive) { observed_[i].alive = false; mappoint = observed_[i].mappoint; break; }
    assert(mappoint != nullptr);
    if (mappoint == nullptr) return;
    mappoint->RemoveObservedByKeyframe(id_);
    // the remaining observers lose one shared point with this frame (frame.cpp:133-150)
    for (const Mappoint::Observation& seenBy : mappoint->ObservationList()) {
        Frame* partner = seenBy.keyframe;
        if (partner == this) continue;
        assert(partner != nullptr);
        partner->UpdateCovisibleKeyframeWeight(id_, covis_.add(seenBy.keyframeId, -1));
    }
}

// the partner's side of a ledger update: it records the weight this frame computed (frame.cpp:157-171)
void Frame::UpdateCovisibleKeyframeWeight(const size_t id, const int weight) {
    std::unique_lock<std::mutex> guard(obsLock_);
    covis_.set(id, weight);
}
}  // namespace myslam
