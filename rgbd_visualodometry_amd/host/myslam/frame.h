// frame.h -- one RGB-D frame and, once promoted, a keyframe with its observations and
// covisibility links (reference include/myslam/frame.h:23-92, src/frame.cpp:18-171).
#ifndef MYSLAM_FRAME_H
#define MYSLAM_FRAME_H
#include "myslam/camera.h"
#include "myslam/common_include.h"
#include "myslam/mappoint.h"

namespace myslam {
class Frame {
public:
    typedef std::shared_ptr<Frame> Ptr;
    typedef std::unordered_map<size_t, int> CovisibleKeyframeIdToWeight;

    double      timestamp_;
    Camera::Ptr camera_;
    Mat         color_, depth_;     // 8UC3 BGR, 16UC1 depth

    // Host images are deep-copied (frame.cpp:28-29); device-resident images are referenced.
    static Frame::Ptr CreateFrame(const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth);
    // The same without the deep copy, for callers that keep their buffers alive until the frame has been consumed (the C wrapper's
    // contract, include/myslam_c.h): the frame's pixels are read exactly once, by the upload into its frame slot (the depth samples
    // the reference takes from Frame::depth_ later, src/frame.cpp:43-67, are taken on the device beside the keypoints).
    static Frame::Ptr CreateFrameView(const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth);

    size_t GetId() const { return id_; }
    SE3 GetPose() { std::unique_lock<std::mutex> lck(poseLock_); return pose_cw_; }
    void SetPose(const SE3 pose) { std::unique_lock<std::mutex> lck(poseLock_); pose_cw_ = pose; }
    double GetDepth(const KeyPoint& kp);                  // metres, -1 if none (frame.cpp:43-67)
    Vector3d GetCamCenter() const { return pose_cw_.inverse().translation(); }
    bool IsCouldObserveMappoint(const Mappoint::Ptr& mpt);
    void AddObservedMappoint(const size_t mappointId, const Point2f pixelPos);
    void AddObservedMappoint(Mappoint* mappoint, const Point2f pixelPos);      // same, without the id lookup
    void RemoveObservedMappoint(const size_t mappointId);
    // The observation set of the reference (frame.h:82, an unordered_set of ids) is stored as one flat, insertion-
    // ordered list: iteration is deterministic and hash-free; a removed observation stays as a dead entry.
    struct ObservedEntry { size_t id; Mappoint* mappoint; bool alive; };
    const std::vector<ObservedEntry>& Observed() const { return observed_; }
    std::unordered_set<size_t> GetObservedMappointIds() {
        std::unique_lock<std::mutex> lck(obsLock_);
        std::unordered_set<size_t> ids;
        for (const ObservedEntry& e : observed_) if (e.alive) ids.insert(e.id);
        return ids;
    }
    bool IsObservedMappoint(const size_t id) {
        std::unique_lock<std::mutex> lck(obsLock_);
        for (size_t i = observed_.size(); i-- > 0;) if (observed_[i].id == id) return observed_[i].alive;
        return false;
    }
    void UpdateCovisibleKeyframeWeight(const size_t id, const int weight);
    // Between Begin and End the covisibility weights raised by AddObservedMappoint are counted per partner keyframe
    // and written once (same final weights and active sets as one update per observation).
    void BeginCovisibilityBatch() { covisBatch_ = true; }
    void EndCovisibilityBatch();
    // Device-resident bookkeeping (FrontEnd with device_keyframes): the weights come counted from the device tables; both sides of the ledger are
    // written here (the two halves of src/frame.cpp:104-119 / :133-150 for one partner)
    void SetCovisibleWeightBoth(Frame* partner, int weight) { covis_.set(partner->id_, weight); partner->covis_.set(id_, weight); }
    void AddCovisibleWeightBoth(Frame* partner, int delta) { partner->covis_.set(id_, covis_.add(partner->id_, delta)); }
    // the observation set rebuilt from the device tables (MapManager::MaterializeFromTables)
    void RestoreObservedClear() { observed_.clear(); }
    void RestoreObserved(Mappoint* mp) { observed_.push_back(ObservedEntry{mp->GetId(), mp, true}); }
    std::unordered_set<size_t> GetCovisibleKeyframes() { std::unique_lock<std::mutex> lck(obsLock_); return covis_.strong; }
    CovisibleKeyframeIdToWeight GetCovisibleKeyframeWeights() { std::unique_lock<std::mutex> lck(obsLock_); return covis_.count; }     // allCovisibleKeyframeIdToWeight_ (frame.h:94)

    int slot_ = -1;                 // vo_ctx frame slot holding this frame's ORB results (-1: none)
    int kfIndex_ = -1;              // dense keyframe number (insertion order) in the device-resident keyframe table; -1: not a keyframe there
    uint64_t baStamp_ = 0; int baIndex_ = -1;    // scratch of Backend::Build
    bool orb_done_ = false;

private:
    // Covisibility ledger: shared-point counts per partner keyframe; a partner is "covisible" (returned by
    // GetCovisibleKeyframes) exactly while its count is >= kCovisibleMin.  One routine keeps both containers in step.
    static constexpr int kCovisibleMin = 15;
    struct CovisLedger {
        CovisibleKeyframeIdToWeight count;
        std::unordered_set<size_t> strong;
        int add(size_t kf, int delta) { return set(kf, count[kf] + delta); }
        int set(size_t kf, int w) {
            if (w <= 0) { count.erase(kf); strong.erase(kf); return 0; }
            count[kf] = w;
            if (w >= kCovisibleMin) strong.insert(kf); else strong.erase(kf);
            return w;
        }
    };
    static std::atomic<size_t> nextId_;
    size_t id_;
    std::mutex poseLock_;
    SE3 pose_cw_;                                           // world -> camera
    std::mutex obsLock_;
    std::vector<ObservedEntry> observed_;
    CovisLedger covis_;
    bool covisBatch_ = false; int covisPending_ = 0;        // batch state (this frame) / pending count (partner frame)
    std::vector<Frame*> covisTouched_;
    Frame(const size_t id, const double timestamp, const Camera::Ptr camera, const Mat color, const Mat depth);
};
}  // namespace myslam
#endif
