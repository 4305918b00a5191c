// mapmanager.cpp -- reference src/mapmanager.cpp:14-38, include/myslam/mapmanager.h:23-58.
#include "myslam/mapmanager.h"

#include <algorithm>

namespace myslam {
namespace { thread_local MapManager* t_bound = nullptr; }

MapManager& MapManager::GetInstance() {
    if (t_bound) return *t_bound;
    static MapManager map_;
    return map_;
}
void MapManager::BindToThread(MapManager* m) { t_bound = m; }

void MapManager::InsertMappoint(const Mappoint::Ptr& mp) {
    std::unique_lock<std::mutex> lck(tableLock_);
    if (mp->slot_ < 0) { mp->slot_ = nextSlot_++; order_.push_back(mp); mp->dirty_ = true; dirty_.push_back(mp.get()); }
    pointsById_[mp->GetId()] = mp;
}

MapManager::MappointIdToPtr MapManager::GetMappointsAroundKeyframe(const Frame::Ptr& keyframe) {
    std::unique_lock<std::mutex> lck(tableLock_);
    auto ids = keyframe->GetCovisibleKeyframes();
    ids.insert(keyframe->GetId());
    MappointIdToPtr local;
    for (auto& kfId : ids) {
        auto kf = keyframesById_.find(kfId);
        assert(kf != keyframesById_.end());
        for (auto& mpId : kf->second->GetObservedMappointIds()) {
            auto mp = pointsById_.find(mpId);
            if (mp == pointsById_.end() || mp->second->outlier_) continue;
            local[mpId] = mp->second;
        }
    }
    return local;
}

// Same set as GetMappointsAroundKeyframe (mapmanager.cpp:14-38) as a vector: keyframes in id order, each
// keyframe's observations in insertion order, de-duplicated with a visit stamp (no hash-map copies).
std::vector<Mappoint*> MapManager::CollectMappointsAroundKeyframe(const Frame::Ptr& keyframe) {
    std::unique_lock<std::mutex> lck(tableLock_);
    auto ids = keyframe->GetCovisibleKeyframes();
    ids.insert(keyframe->GetId());
    std::vector<size_t> kfs(ids.begin(), ids.end());
    std::sort(kfs.begin(), kfs.end());
    std::vector<Mappoint*> out;
    const uint64_t stamp = ++stamp_;
    for (size_t kfId : kfs) {
        auto kf = keyframesById_.find(kfId);
        assert(kf != keyframesById_.end());
        for (const Frame::ObservedEntry& e : kf->second->Observed()) {
            Mappoint& mp = *e.mappoint;
            if (!e.alive || mp.visitStamp_ == stamp || mp.outlier_) continue;
            mp.visitStamp_ = stamp;
            out.push_back(&mp);
        }
    }
    return out;
}
}  // namespace myslam
