// mappoint.h -- 3-D landmark (reference include/myslam/mappoint.h:36-84, src/mappoint.cpp:17-49).
#ifndef MYSLAM_MAPPOINT_H
#define MYSLAM_MAPPOINT_H
#include "myslam/common_include.h"

namespace myslam {
class Frame;
class Mappoint {
public:
    typedef std::shared_ptr<Mappoint> Ptr;
    typedef std::unordered_map<size_t, Point2f> ObservedByKeyframeIdtoPixelPos;
    struct Observation { size_t keyframeId; Point2f pixel; Frame* keyframe; };

private:
    // Hot fields first, in ONE cache line: the graph cut, the local-map query and the BA merge walk tens of thousands of
    // map points per keyframe and touch exactly these (position, observation list, visit / index scratch, device slot).
    Vector3d pos_;
    std::vector<Observation> obsList_;
public:
    uint64_t baStamp_ = 0; int baIndex_ = -1;    // scratch of Backend::Build
    int  slot_ = -1;                // device-map bookkeeping (slot in the vo_ctx map, set by MapManager::InsertMappoint)
    uint64_t visitStamp_ = 0;       // scratch for de-duplicated traversals
    bool        outlier_;           // no observation left / rejected
    bool        triangulated_;      // refined by the front-end's triangulation
    bool        optimized_;         // touched by the back-end
    bool dirty_ = false;            // host copy newer than the device copy (queued in MapManager's dirty list)

    Descriptor  descriptor_;        // 256-bit rBRIEF descriptor used for matching

    static Mappoint::Ptr CreateMappoint(const Vector3d position, const Descriptor& descriptor);

    Vector3d GetPosition() { std::unique_lock<std::mutex> lock(posLock_); return pos_; }
    void SetPosition(const Vector3d pos) { { std::unique_lock<std::mutex> lock(posLock_); pos_ = pos; } MarkDirty(); }
    // position already mirrored on the device by the caller (bulk upsert): no dirty marking
    void SetPositionSynced(const Vector3d pos) { std::unique_lock<std::mutex> lock(posLock_); pos_ = pos; }
    // Bulk paths of the map-owning thread (BA graph cut / merge touch every point of the local map): no lock per point.
    const Vector3d& PositionUnlocked() const { return pos_; }
    void SetPositionSyncedUnlocked(const Vector3d& pos) { pos_ = pos; }
    size_t GetId() const { return id_; }
    Vector3d GetNormDirection() { std::unique_lock<std::mutex> lock(obsLock_); return norm_; }

    void AddObservedByKeyframe(const size_t keyframeId, const Point2f posInPixel, const Vector3d cameraCenter, Frame* keyframe = nullptr);
    const std::vector<Observation>& ObservationList() const { return obsList_; }     // insertion (= keyframe id) order; single-threaded callers
    void RemoveObservedByKeyframe(const size_t keyframeId);
    // the reference's container (mappoint.h:71), assembled on request: the flat observation list is the storage
    ObservedByKeyframeIdtoPixelPos GetObservedByKeyframesMap() {
        std::unique_lock<std::mutex> lock(obsLock_);
        ObservedByKeyframeIdtoPixelPos m;
        for (const Observation& o : obsList_) m[o.keyframeId] = o.pixel;
        return m;
    }

    void MarkDirty();
    // state taken over from the device tables (MapManager::MaterializeFromTables): no dirty marking, no recomputation
    void RestoreState(const Vector3d& pos, const Vector3d& norm, bool outlier, bool triangulated, bool optimized) { pos_ = pos; norm_ = norm; outlier_ = outlier; triangulated_ = triangulated; optimized_ = optimized; }
    void RestoreObservationsClear() { obsList_.clear(); }
    void RestoreObservation(const size_t keyframeId, const Point2f pixel, Frame* keyframe) { obsList_.push_back(Observation{keyframeId, pixel, keyframe}); }

private:
    static std::atomic<size_t> nextId_;
    size_t id_;
    Vector3d norm_;
    std::mutex posLock_;
    std::mutex obsLock_;
    Mappoint(const size_t id, const Vector3d position, const Descriptor& descriptor);
};
}  // namespace myslam
#endif
