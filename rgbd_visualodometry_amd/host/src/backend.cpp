// backend.cpp -- local BA over the covisibility graph (reference src/backend.cpp:19-195).  The
// graph is flattened into the vo_ba_problem arrays; the LM/Schur numerics run in vo_local_ba.
#include "myslam/backend.h"

#include <algorithm>
#include <chrono>
#include <stdexcept>

#include "myslam/config.h"

namespace myslam {

Backend::Backend(const Camera::Ptr camera) : camera_(camera) {
    chi2Threshold_ = Config::has("chi2_th") ? Config::get<float>("chi2_th") : 1.0f;       // backend.h:24
}

void Backend::OptimizeCovisibleGraphOfKeyframe(const Frame::Ptr keyframeCurr) {
    keyframeCurr_ = keyframeCurr;
    Optimize();
}

void Backend::Optimize() {
    if (!ctx_) throw std::runtime_error("Backend has no compute context (FrontEnd::SetBackend binds it)");
    auto t0 = std::chrono::steady_clock::now();
    MapManager& map = MapManager::GetInstance();

    // free poses: covisible keyframes + the current one (backend.cpp:36-59), id order
    auto covis = keyframeCurr_->GetCovisibleKeyframes();
    covis.insert(keyframeCurr_->GetId());
    std::vector<size_t> freeIds(covis.begin(), covis.end());
    std::sort(freeIds.begin(), freeIds.end());
    std::vector<Frame::Ptr> poseFrames;
    std::unordered_map<size_t, int> poseIndex;
    std::vector<Mappoint::Ptr> points;
    std::unordered_map<size_t, int> pointIndex;
    for (size_t id : freeIds) {
        auto kf = map.GetKeyframe(id);
        if (kf == nullptr) continue;
        poseIndex[id] = (int)poseFrames.size();
        poseFrames.push_back(kf);
    }
    const int nFree = (int)poseFrames.size();
    // points: every non-outlier map point observed by a free keyframe (backend.cpp:62-81)
    for (int j = 0; j < nFree; ++j) {
        auto obs = poseFrames[j]->GetObservedMappointIds();
        std::vector<size_t> ids(obs.begin(), obs.end());
        std::sort(ids.begin(), ids.end());
        for (size_t mpId : ids) {
            if (pointIndex.count(mpId)) continue;
            auto mp = map.GetMappoint(mpId);
            if (mp == nullptr || mp->outlier_) continue;
            pointIndex[mpId] = (int)points.size();
            points.push_back(mp);
        }
    }
    // edges: every observation of those points; observers outside the free set are fixed (backend.cpp:88-135)
    std::vector<int32_t> edgePose, edgePoint; std::vector<float> edgeUv;
    std::vector<std::pair<Frame::Ptr, Mappoint::Ptr>> edgeOwner;
    for (size_t k = 0; k < points.size(); ++k) {
        auto obs = points[k]->GetObservedByKeyframesMap();
        std::vector<size_t> kfIds;
        for (auto& kv : obs) kfIds.push_back(kv.first);
        std::sort(kfIds.begin(), kfIds.end());
        for (size_t kfId : kfIds) {
            auto kf = map.GetKeyframe(kfId);
            if (kf == nullptr) continue;
            auto it = poseIndex.find(kfId);
            int pj;
            if (it != poseIndex.end()) pj = it->second;
            else { pj = (int)poseFrames.size(); poseIndex[kfId] = pj; poseFrames.push_back(kf); }
            edgePose.push_back(pj); edgePoint.push_back((int)k);
            edgeUv.push_back(obs[kfId].x); edgeUv.push_back(obs[kfId].y);
            edgeOwner.emplace_back(kf, points[k]);
        }
    }
    if (edgePose.empty() || nFree == 0) return;

    std::vector<double> poses(12 * poseFrames.size()), pts(3 * points.size());
    for (size_t j = 0; j < poseFrames.size(); ++j) poseFrames[j]->GetPose().to12(&poses[12 * j]);
    for (size_t k = 0; k < points.size(); ++k) { Vector3d p = points[k]->GetPosition(); pts[3 * k] = p[0]; pts[3 * k + 1] = p[1]; pts[3 * k + 2] = p[2]; }
    vo_ba_problem prob;
    prob.n_poses = (int)poseFrames.size(); prob.n_free = nFree; prob.n_points = (int)points.size(); prob.n_edges = (int)edgePose.size();
    prob.poses = poses.data(); prob.points = pts.data(); prob.edge_pose = edgePose.data(); prob.edge_point = edgePoint.data(); prob.edge_uv = edgeUv.data();
    prob.huber_delta = std::sqrt(7.815); prob.chi2_th = chi2Threshold_; prob.it_robust = 10; prob.it_plain = 10;
    std::vector<double> posesOut(12 * (size_t)nFree), ptsOut(3 * points.size());
    std::vector<uint8_t> flags(edgePose.size());
    vo_ba_result res;
    std::memset(&res, 0, sizeof(res));
    res.poses = posesOut.data(); res.points = ptsOut.data(); res.edge_flags = flags.data();
    auto t1 = std::chrono::steady_clock::now();
    int rc = vo_local_ba(ctx_, &prob, &res);
    auto t2 = std::chrono::steady_clock::now();
    stats_.ms_build += std::chrono::duration<double, std::milli>(t1 - t0).count(); stats_.ms_solve += std::chrono::duration<double, std::milli>(t2 - t1).count();
    if (rc != VO_OK) throw std::runtime_error(std::string("vo_local_ba failed: ") + vo_strerror(rc));

    int outlierCnt = 0;
    for (size_t e = 0; e < flags.size(); ++e) {                                 // backend.cpp:144-172
        if (flags[e] & 3) {
            auto& kf = edgeOwner[e].first; auto& mp = edgeOwner[e].second;
            if (kf->IsObservedMappoint(mp->GetId())) kf->RemoveObservedMappoint(mp->GetId());
            ++outlierCnt;
        }
        edgeOwner[e].second->optimized_ = true;
    }
    for (int j = 0; j < nFree; ++j) poseFrames[j]->SetPose(SE3::from12(&posesOut[12 * (size_t)j]));     // backend.cpp:183-187
    for (size_t k = 0; k < points.size(); ++k)                                                           // backend.cpp:188-194
        if (!points[k]->outlier_) points[k]->SetPosition(Vector3d(ptsOut[3 * k], ptsOut[3 * k + 1], ptsOut[3 * k + 2]));

    stats_.runs++; stats_.poses = nFree; stats_.fixed = (int)poseFrames.size() - nFree; stats_.points = (int)points.size();
    stats_.edges = (int)edgePose.size(); stats_.outliers = outlierCnt;
    stats_.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace myslam
