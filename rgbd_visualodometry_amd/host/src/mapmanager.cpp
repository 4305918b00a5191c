// mapmanager.cpp -- reference src/mapmanager.cpp:14-38, include/myslam/mapmanager.h:23-58.
#include "myslam/mapmanager.h"

#include <algorithm>
#include <stdexcept>

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
void MapManager::MaterializeFromTables(vo_ctx* ctx) {
    std::unique_lock<std::mutex> lck(tableLock_);
    int64_t no = 0; int32_t na = 0;
    int rc = vo_tables_fetch(ctx, 0, 0, nullptr, nullptr, nullptr, nullptr, &no, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &na);
    if (rc != VO_OK) throw std::runtime_error(std::string("vo_tables_fetch failed: ") + vo_strerror(rc));
    const size_t nm = (size_t)nextSlot_;
    std::vector<int32_t> okf((size_t)no), omp((size_t)no); std::vector<float> ouv(2 * (size_t)no); std::vector<uint8_t> oal((size_t)no);
    std::vector<double> xyz(3 * nm), nrm(3 * nm); std::vector<uint8_t> desc(32 * nm), fl(nm);
    rc = vo_tables_fetch(ctx, 0, no, okf.data(), omp.data(), ouv.data(), oal.data(), &no, 0, (int32_t)nm, xyz.data(), nrm.data(), desc.data(), fl.data(), nullptr, 0, &na);
    if (rc != VO_OK) throw std::runtime_error(std::string("vo_tables_fetch failed: ") + vo_strerror(rc));
    for (size_t k = 0; k < nm; ++k) {
        Descriptor d; std::memcpy(d.data(), &desc[32 * k], 32);
        if (k >= order_.size()) {
            Mappoint::Ptr mp = Mappoint::CreateMappoint(Vector3d(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]), d);
            mp->slot_ = (int)k; order_.push_back(mp); pointsById_[mp->GetId()] = mp;
        }
        Mappoint& mp = *order_[k];
        mp.descriptor_ = d;
        mp.RestoreState(Vector3d(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]), Vector3d(nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]), fl[k] & VO_MAP_FLAG_OUTLIER, fl[k] & VO_MAP_FLAG_TRIANGULATED, fl[k] & VO_MAP_FLAG_OPTIMIZED);
        mp.RestoreObservationsClear();
    }
    for (Frame* f : kfByIndex_) if (f) f->RestoreObservedClear();
    for (size_t o = 0; o < (size_t)no; ++o) {
        if (!oal[o] || (size_t)omp[o] >= nm || (size_t)okf[o] >= kfByIndex_.size() || !kfByIndex_[okf[o]]) continue;
        Frame* f = kfByIndex_[okf[o]]; Mappoint* mp = order_[omp[o]].get();
        f->RestoreObserved(mp);
        mp->RestoreObservation(f->GetId(), Point2f(ouv[2 * o], ouv[2 * o + 1]), f);
    }
}
}  // namespace myslam
