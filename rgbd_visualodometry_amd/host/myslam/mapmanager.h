// mapmanager.h -- owner of all keyframes and map points (reference include/myslam/mapmanager.h:23-62,
// src/mapmanager.cpp:14-38).  GetInstance() returns the map bound to the calling thread's VO
// system (one process may drive several independent streams); without a binding it is the
// process-wide singleton of the reference.
#ifndef MYSLAM_MAPMANAGER_H
#define MYSLAM_MAPMANAGER_H
#include <algorithm>
#include "myslam/common_include.h"
#include "myslam/frame.h"
#include "myslam/mappoint.h"

namespace myslam {
class MapManager {
public:
    typedef std::shared_ptr<MapManager> Ptr;
    typedef std::unordered_map<size_t, Mappoint::Ptr> MappointIdToPtr;
    typedef std::unordered_map<size_t, Frame::Ptr> KeyframeIdToPtr;

    static MapManager& GetInstance();
    static void BindToThread(MapManager* m);       // nullptr -> process-wide singleton

    void InsertKeyframe(const Frame::Ptr& frame) { std::unique_lock<std::mutex> lck(tableLock_); keyframesById_[frame->GetId()] = frame; }
    Frame::Ptr GetKeyframe(const size_t id) { std::unique_lock<std::mutex> lck(tableLock_); auto it = keyframesById_.find(id); return it == keyframesById_.end() ? nullptr : it->second; }
    KeyframeIdToPtr GetAllKeyframes() { std::unique_lock<std::mutex> lck(tableLock_); return keyframesById_; }
    void InsertMappoint(const Mappoint::Ptr& map_point);
    Mappoint::Ptr GetMappoint(const size_t id) { std::unique_lock<std::mutex> lck(tableLock_); auto it = pointsById_.find(id); return it == pointsById_.end() ? nullptr : it->second; }
    MappointIdToPtr GetAllMappoints() { std::unique_lock<std::mutex> lck(tableLock_); return pointsById_; }
    MappointIdToPtr GetMappointsAroundKeyframe(const Frame::Ptr& keyframe);

    Mappoint* FindMappoint(const size_t id) { auto it = pointsById_.find(id); return it == pointsById_.end() ? nullptr : it->second.get(); }   // no lock, no copy
    size_t MappointCount() { std::unique_lock<std::mutex> lck(tableLock_); return std::max(pointsById_.size(), (size_t)nextSlot_); }
    // Device-resident map (FrontEnd with device_keyframes): map points are created ON the device (vo_keyframe_commit); the host only hands out their slots.
    int NextSlot() const { return nextSlot_; }
    void ReserveSlots(int n) { std::unique_lock<std::mutex> lck(tableLock_); nextSlot_ += n; }
    // Mappoint objects, observation lists and observation sets rebuilt from the device tables (inspection, viewers, tests): everything the
    // reference's containers hold (include/myslam/mapmanager.h:23-58), as of now.  Objects of slots that already exist are updated in place.
    void MaterializeFromTables(vo_ctx* ctx);
    // map points whose host state is newer than the device copy (drained by the front-end once per frame)
    void NoteDirty(Mappoint* mp) { dirty_.push_back(mp); }
    std::vector<Mappoint*> TakeDirty() { std::vector<Mappoint*> d; d.swap(dirty_); return d; }
    // all map points in insertion (= id) order, and the de-duplicated local map of a keyframe as vectors
    const std::vector<Mappoint::Ptr>& AllMappointsOrdered() const { return order_; }
    std::vector<Mappoint*> CollectMappointsAroundKeyframe(const Frame::Ptr& keyframe);

private:
    std::mutex tableLock_;
    MappointIdToPtr pointsById_;
    KeyframeIdToPtr keyframesById_;
    int nextSlot_ = 0;
    std::vector<Mappoint*> dirty_;
    std::vector<Mappoint::Ptr> order_;
    uint64_t stamp_ = 0;
public:
    // Device-resident bookkeeping: keyframes by their dense number and the observation registry -- observation id
    // (append order of vo_obs_append) -> (keyframe, map point), so that a device-side cull list can be applied to the host structures.
    std::vector<Frame*> kfByIndex_;
    struct ObsRef { Frame* keyframe; Mappoint* mappoint; };
    std::vector<ObsRef> obsRegistry_;
    Mappoint* MappointBySlot(int slot) const { return (slot >= 0 && (size_t)slot < order_.size()) ? order_[slot].get() : nullptr; }
};
}  // namespace myslam
#endif
