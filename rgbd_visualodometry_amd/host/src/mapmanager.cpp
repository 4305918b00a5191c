// mapmanager.cpp -- reference src/mapmanager.cpp:14-38, include/myslam/mapmanager.h:23-58.
#include "myslam/mapmanager.h"

namespace myslam {
namespace { thread_local MapManager* t_bound = nullptr; }

MapManager& MapManager::GetInstance() {
    if (t_bound) return *t_bound;
    static MapManager map_;
    return map_;
}
void MapManager::BindToThread(MapManager* m) { t_bound = m; }

void MapManager::InsertMappoint(const Mappoint::Ptr& mp) {
    std::unique_lock<std::mutex> lck(dataMutex_);
    if (mp->slot_ < 0) mp->slot_ = nextSlot_++;
    mappointsDict_[mp->GetId()] = mp;
}

MapManager::MappointIdToPtr MapManager::GetMappointsAroundKeyframe(const Frame::Ptr& keyframe) {
    std::unique_lock<std::mutex> lck(dataMutex_);
    auto ids = keyframe->GetCovisibleKeyframes();
    ids.insert(keyframe->GetId());
    MappointIdToPtr local;
    for (auto& kfId : ids) {
        auto kf = keyframesDict_.find(kfId);
        assert(kf != keyframesDict_.end());
        for (auto& mpId : kf->second->GetObservedMappointIds()) {
            auto mp = mappointsDict_.find(mpId);
            if (mp == mappointsDict_.end() || mp->second->outlier_) continue;
            local[mpId] = mp->second;
        }
    }
    return local;
}

std::vector<Mappoint::Ptr> MapManager::TakeDirtyMappoints() { std::unique_lock<std::mutex> lck(dataMutex_); std::vector<Mappoint::Ptr> d; d.swap(dirty_); return d; }
void MapManager::MarkDirty(const Mappoint::Ptr& mp) { std::unique_lock<std::mutex> lck(dataMutex_); dirty_.push_back(mp); }
}  // namespace myslam
