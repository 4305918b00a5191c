// backend.cpp -- local BA over the covisibility graph (reference src/backend.cpp:19-195).  The graph is
// flattened into the vo_ba_problem arrays; the LM/Schur numerics run in vo_local_ba.
#include <cstdio>
#include <cstdlib>
#include "myslam/backend.h"
#include "myslam/rccl_exchange.h"

#include <algorithm>
#include <chrono>
#include <stdexcept>

#include "myslam/config.h"
#include "myslam/util.h"

namespace myslam {

static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

Backend::Backend(const Camera::Ptr camera) : camera_(camera) {
    chi2Threshold_ = Config::has("chi2_th") ? Config::get<float>("chi2_th") : 1.0f;       // backend.h:24
    if (Config::has("backend_lag_frames")) lag_ = std::max(0, Config::get<int>("backend_lag_frames"));
    if (Config::has("ba_max_free_keyframes")) maxFree_ = std::max(1, Config::get<int>("ba_max_free_keyframes"));
    if (Config::has("ba_device_graph")) deviceGraph_ = Config::get<int>("ba_device_graph") != 0;
    // experiment key (not a reference setting): the reference never fixes a vertex (backend.cpp:49-59: setFixed(id == 0), ids start at 1), so
    // every local BA floats in its 6-dof gauge.  With this key the OLDEST keyframe of the free set is treated as fixed.
    if (Config::has("ba_fix_oldest_free_keyframe")) fixOldest_ = Config::get<int>("ba_fix_oldest_free_keyframe") != 0;
    if (const char* e = std::getenv("VO_TEST_FAIL_CUT_AT")) testFailAt_ = std::atoi(e);
}

Backend::~Backend() { Stop(); if (ctxOwn_) vo_ctx_destroy(ctxOwn_); if (rcclComm_) myslam_rccl_comm_destroy(rcclComm_); }

void Backend::ShardOverRanks(int rank, int world, const std::string& idFile) {
    if (world <= 1) return;
    char id[MYSLAM_RCCL_ID_BYTES];
    if (myslam_rccl_id_via_file(idFile.c_str(), rank, 120, id) || myslam_rccl_comm_create(id, rank, world, &rcclComm_))
        throw std::runtime_error(std::string("RCCL: ") + myslam_rccl_last_error());
    shardRank_ = rank; shardWorld_ = world;
    deviceGraph_ = false; deviceKeyframes_ = false;     // the sharded solve takes explicit problems: the graph is cut on the host
    if (ctx_) (void)vo_set_ba_shard_stream(ctx_, rank, world, myslam_rccl_allreduce_f64, rcclComm_);
    if (ctxOwn_) (void)vo_set_ba_shard_stream(ctxOwn_, rank, world, myslam_rccl_allreduce_f64, rcclComm_);
}

void Backend::SetContext(vo_ctx* ctx, int device) { ctx_ = ctx; device_ = device; if (ctx_ && lag_ > 0) EnsureWorker(); }

void Backend::EnsureWorker() {
    if (!ctxOwn_) {                                  // small private context: BA needs scratch + a stream only
        vo_params p; vo_default_params(&p);
        p.fx = camera_->GetFx(); p.fy = camera_->GetFy(); p.cx = camera_->GetCx(); p.cy = camera_->GetCy();
        p.n_features = 64; p.max_frames = 1; p.map_capacity = 64; p.max_hypotheses = 1;
        p.stream_priority = -1;                         // problem preparation (graph cut, uploads, pair lists) in the lowest class: a pool of hardware queues of its own, away from
                                                        // the trackers' streams (default class) and from the pace-setting chains (highest class: BA engines, group chains)
        int rc = vo_ctx_create(&p, device_, &ctxOwn_);
        if (rc != VO_OK) throw std::runtime_error(std::string("vo_ctx_create (backend) failed: ") + vo_strerror(rc));
        if (shardWorld_ > 1) (void)vo_set_ba_shard_stream(ctxOwn_, shardRank_, shardWorld_, myslam_rccl_allreduce_f64, rcclComm_);
    }
    if (!worker_.joinable()) worker_ = std::thread(&Backend::WorkerLoop, this);
}

void Backend::Stop() {
    if (job_) Finish();
    if (worker_.joinable()) {
        { std::unique_lock<std::mutex> lk(mu_); quit_ = true; workSeq_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        worker_.join();
    }
}

// poll `seq` (it changes when the other side has something for us) for up to ~300 us with the lock released, then wait on the condition variable
template <typename Pred> void Backend::SpinThenWait(std::unique_lock<std::mutex>& lk, std::atomic<int>& seq, int seen, Pred pred) {
    if (!pred()) {
        lk.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; seq.load(std::memory_order_acquire) == seen; ++i) {
            __builtin_ia32_pause();
            if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
        }
        lk.lock();
    }
    cv_.wait(lk, pred);
}

void Backend::WorkerLoop() {
    int seenWork = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(mu_);
        SpinThenWait(lk, workSeq_, seenWork, [&] { return quit_ || hasWork_; });
        seenWork = workSeq_.load(std::memory_order_acquire);
        if (quit_) return;
        hasWork_ = false;
        Job* j = job_.get();
        lk.unlock();
        if (j->resident) {                                   // cut on the device from the tracker's tables, then solve
            auto t0 = std::chrono::steady_clock::now();
            { VO_SCOPE("bw.cut");
            j->rc = vo_local_ba_resident_cut(ctxOwn_, ctx_, j->freeKf.data(), (int)j->freeKf.size(), std::sqrt(7.815), chi2Threshold_, &j->nPoints, &j->nFixed, &j->nEdges); }
            if (testFailAt_ > 0 && ++nCuts_ == testFailAt_ && j->rc == VO_OK) j->rc = VO_E_UNSUPPORTED;
            lk.lock(); j->cutDone = true; cutSeq_.fetch_add(1, std::memory_order_release); cv_.notify_all(); lk.unlock();
            if (j->rc == VO_OK) { VO_SCOPE("bw.solve"); SolveResident(*j, ctxOwn_); }
            j->solveMs = ms_since(t0);
        } else Solve(*j, ctxOwn_);
        j->tDone = std::chrono::steady_clock::now();
        lk.lock();
        j->done = true;
        doneSeq_.fetch_add(1, std::memory_order_release);
        cv_.notify_all();
    }
}

bool Backend::Poll(size_t frameIndex) {
    frameIndex_ = frameIndex;
    if (job_ && frameIndex >= job_->frameIndex + (size_t)lag_) { Finish(); return true; }
    return false;
}

void Backend::Finish(bool deferTail) {
    auto t0 = std::chrono::steady_clock::now();
    bool waited = false;
    if (lag_ > 0) { VO_SCOPE("ba.wait_done"); const int seen = doneSeq_.load(std::memory_order_acquire); std::unique_lock<std::mutex> lk(mu_); waited = !job_->done; SpinThenWait(lk, doneSeq_, seen, [&] { return job_->done; }); }
    stats_.ms_wait += ms_since(t0);
    const auto tWake = std::chrono::steady_clock::now();
    if (waited) { stats_.ms_wake += std::chrono::duration<double, std::milli>(tWake - job_->tDone).count(); ++stats_.waited; }
    std::unique_ptr<Job> j = std::move(job_);
    if (j->rc != VO_OK) {                            // a failed solve must not end the stream: the map keeps its un-optimised state
        if (j->resident && deviceGraph_ && (j->rc == VO_E_UNSUPPORTED || j->rc == VO_E_OVERFLOW || j->rc == VO_E_NOMEM)) {
            std::cerr << "[myslam] device graph cut unavailable (" << vo_strerror(j->rc) << "): local BA graphs are cut on the host from here on" << std::endl;
            if (deviceKeyframes_ && onDeviceFallback_) onDeviceFallback_();      // the host objects the host cut walks are rebuilt from the device tables first
            deviceGraph_ = false; deviceKeyframes_ = false;
        }
        if (stats_.failed++ == 0) std::cerr << "[myslam] vo_local_ba failed (" << vo_strerror(j->rc) << "): this local BA is skipped, tracking continues" << std::endl;
        return;
    }
    if (!j->resident) { Apply(*j); return; }
    // Device-cut BA (reference src/backend.cpp:144-194 in two halves).  Now: the culled observations leave the host ledger (the next
    // free-keyframe set depends on the covisibility counts they change) and the result is merged into the device tables ON the device
    // (vo_local_ba_resident_merge: no trip to the host and back before the next graph cut may start).  Later (FinishTail): the host
    // objects receive their copy.
    FinishTail();                                    // an earlier tail, if its owner never came back for it
    MapManager& map = MapManager::GetInstance();
    if (j->nPoints == 0 || j->nEdges == 0) return;
    if (deviceKeyframes_) { FinishOnDevice(*j, lag_ > 0 ? ctxOwn_ : ctx_); if (waited) stats_.ms_to_merge += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tWake).count(); return; }
    { VO_SCOPE("ba.culled");
    for (int i = 0; i < j->nCulled; ++i) {
        const MapManager::ObsRef& o = map.obsRegistry_[(size_t)j->culled[i]];
        if (o.keyframe->IsObservedMappoint(o.mappoint->GetId())) o.keyframe->RemoveObservedMappoint(o.mappoint->GetId());
    } }
    vo_ctx* solver = lag_ > 0 ? ctxOwn_ : ctx_;
    int rc;
    { VO_SCOPE("ba.merge_call"); rc = vo_local_ba_resident_merge(solver, ctx_); }
    if (rc != VO_OK) throw std::runtime_error(std::string("vo_local_ba_resident_merge failed: ") + vo_strerror(rc));
    if (waited) stats_.ms_to_merge += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tWake).count();
    tail_ = std::move(j); tailCtx_ = solver;
    if (!deferTail) FinishTail();
}

// The write-back of reference src/backend.cpp:144-194 when the map lives on the device: positions, flags, poses and culled observations are merged there
// (vo_local_ba_resident_merge_ledger); the host applies the covisibility decrements to its ledger (src/frame.cpp:133-150) and takes the free poses.
void Backend::FinishOnDevice(Job& j, vo_ctx* solver) {
    VO_SCOPE("ba.merge_ledger");
    MapManager& map = MapManager::GetInstance();
    if (pairA_.size() < 16384) { pairA_.resize(16384); pairB_.resize(16384); }
    j.posesOut.resize(12 * (size_t)std::max(j.nFree, 1));
    int32_t np = 0;
    int rc = vo_local_ba_resident_merge_ledger(solver, ctx_, pairA_.data(), pairB_.data(), (int)pairA_.size(), &np, j.posesOut.data(), j.nFree);
    if (rc == VO_E_OVERFLOW && np > (int)pairA_.size()) {       // more decrements than the arrays hold (a noisy graph: culled observations x co-observers): the call says how many, the repeated call delivers them
        pairA_.resize((size_t)np + 1024); pairB_.resize((size_t)np + 1024);
        rc = vo_local_ba_resident_merge_ledger(solver, ctx_, pairA_.data(), pairB_.data(), (int)pairA_.size(), &np, j.posesOut.data(), j.nFree);
    }
    if (rc != VO_OK) throw std::runtime_error(std::string("vo_local_ba_resident_merge_ledger failed: ") + vo_strerror(rc));
    for (int i = 0; i < np; ++i) {
        Frame* a = (size_t)pairA_[i] < map.kfByIndex_.size() ? map.kfByIndex_[pairA_[i]] : nullptr;
        Frame* b = (size_t)pairB_[i] < map.kfByIndex_.size() ? map.kfByIndex_[pairB_[i]] : nullptr;
        if (a && b && a != b) a->AddCovisibleWeightBoth(b, -1);
    }
    for (int p = 0; p < j.nFree; ++p) j.poseFrames[p]->SetPose(SE3::from12(&j.posesOut[12 * (size_t)p]));
    stats_.runs++; stats_.poses = j.nFree; stats_.fixed = j.nFixed; stats_.points = j.nPoints; stats_.edges = j.nEdges; stats_.outliers = j.nCulled; stats_.ms_solve += j.solveMs;
    { const double D = 6.0 * j.nFree; stats_.sum_d3 += D * D * D; stats_.sum_d2 += D * D; stats_.sum_edges += j.nEdges; stats_.sum_points += j.nPoints; stats_.sum_pairs += j.nPairs; }
    if (j.culled.capacity() > culledSpare_.capacity()) culledSpare_.swap(j.culled);      // the list's buffer goes on to the next job
}

void Backend::FinishTail() {
    if (!tail_) return;
    VO_SCOPE("ba.apply");
    std::unique_ptr<Job> jp = std::move(tail_);
    Job& j = *jp;
    j.posesOut.resize(12 * (size_t)std::max(j.nFree, 1)); j.pointSlots.resize((size_t)j.nPoints); j.ptsOut.resize(3 * (size_t)j.nPoints);
    vo_ba_resident_result r;
    std::memset(&r, 0, sizeof(r));
    r.poses = j.posesOut.data(); r.point_slots = j.pointSlots.data(); r.points = j.ptsOut.data(); r.culled_obs = j.culled.data();
    r.cap_points = j.nPoints; r.cap_culled = (int)j.culled.size();
    int rc = vo_local_ba_resident_fetch(tailCtx_, &r);
    if (rc != VO_OK) throw std::runtime_error(std::string("vo_local_ba_resident_fetch failed: ") + vo_strerror(rc));
    MapManager& map = MapManager::GetInstance();
    for (int p = 0; p < j.nFree; ++p) j.poseFrames[p]->SetPose(SE3::from12(&j.posesOut[12 * (size_t)p]));
    const size_t np = (size_t)j.nPoints;
    for (size_t k = 0; k < np; ++k) {
        if (k + 8 < np) __builtin_prefetch(map.MappointBySlot(j.pointSlots[k + 8]), 1);
        Mappoint* mp = map.MappointBySlot(j.pointSlots[k]);
        if (!mp) continue;
        mp->optimized_ = true;
        if (mp->outlier_) continue;                  // (a point that lost its last observation to this BA's culls: its flag travels with the next dirty-list flush)
        const double* x = &j.ptsOut[3 * k];
        mp->SetPositionSyncedUnlocked(Vector3d(x[0], x[1], x[2]));
    }
    stats_.runs++; stats_.poses = j.nFree; stats_.fixed = j.nFixed; stats_.points = j.nPoints; stats_.edges = j.nEdges; stats_.outliers = j.nCulled; stats_.ms_solve += j.solveMs;
    { const double D = 6.0 * j.nFree; stats_.sum_d3 += D * D * D; stats_.sum_d2 += D * D; stats_.sum_edges += j.nEdges; stats_.sum_points += j.nPoints; stats_.sum_pairs += j.nPairs; }
    if (j.culled.capacity() > culledSpare_.capacity()) culledSpare_.swap(j.culled);
}

void Backend::OptimizeCovisibleGraphOfKeyframe(const Frame::Ptr keyframeCurr, bool deferTail) {
    if (!ctx_) throw std::runtime_error("Backend has no compute context (FrontEnd::SetBackend binds it)");
    auto t0 = std::chrono::steady_clock::now();
    if (job_) Finish(true);                          // the previous result is merged before a new graph is cut (its host-side copy follows below, beside the new solve)
    std::unique_ptr<Job> j(new Job);
    j->culled.swap(culledSpare_);                    // (the previous job's list buffer, see SolveResident)
    j->frameIndex = frameIndex_;
    // The device graph cut takes VO_BA_RESIDENT_MAX_FREE free poses (include/vo_hip.h; the same number as maxFree_'s default).  A keyframe with
    // more covisible keyframes than that hands the graph cut back to the host for the rest of the run -- the same transition as a full
    // observation table -- instead of silently solving a smaller problem than the host cut would.
    if (deviceGraph_ && !deviceKeyframes_ && keyframeCurr->kfIndex_ >= 0 && (int)keyframeCurr->GetCovisibleKeyframes().size() + 1 > std::min(maxFree_, VO_BA_RESIDENT_MAX_FREE)) {
        std::fprintf(stderr, "[myslam_amd] local BA: %zu covisible keyframes exceed the device graph cut's %d free poses: the host graph cut takes over\n",
                     keyframeCurr->GetCovisibleKeyframes().size(), VO_BA_RESIDENT_MAX_FREE);
        deviceGraph_ = false;
    }
    if (deviceGraph_ && keyframeCurr->kfIndex_ >= 0) {      // only the free keyframes' numbers go to the device; no host graph cut
        auto covis = keyframeCurr->GetCovisibleKeyframes();
        std::vector<size_t> ids(covis.begin(), covis.end());
        const int cap = std::min(maxFree_, VO_BA_RESIDENT_MAX_FREE);
        if ((int)ids.size() + 1 > cap) {             // (device keyframes only: there is no host graph to fall back to) the strongest stay free, as in Build
            auto w = keyframeCurr->GetCovisibleKeyframeWeights();
            std::sort(ids.begin(), ids.end(), [&](size_t a, size_t b) { const int wa = w[a], wb = w[b]; return wa != wb ? wa > wb : a > b; });
            ids.resize((size_t)cap - 1);
            ++stats_.capped;
        }
        ids.push_back(keyframeCurr->GetId());
        std::sort(ids.begin(), ids.end());
        if (fixOldest_ && ids.size() > 1) ids.erase(ids.begin());      // its observations still constrain the points: it joins the fixed poses
        MapManager& map = MapManager::GetInstance();
        for (size_t id : ids) { auto f = map.GetKeyframe(id); if (f && f->kfIndex_ >= 0) { j->freeKf.push_back(f->kfIndex_); j->poseFrames.push_back(f.get()); } }
        j->nFree = (int)j->freeKf.size(); j->resident = true;
        stats_.ms_build += ms_since(t0);
        job_ = std::move(j);
        if (lag_ == 0) {
            auto t1 = std::chrono::steady_clock::now();
            job_->rc = vo_local_ba_resident_cut(ctx_, ctx_, job_->freeKf.data(), job_->nFree, std::sqrt(7.815), chi2Threshold_, &job_->nPoints, &job_->nFixed, &job_->nEdges);
            job_->cutDone = true;
            if (testFailAt_ > 0 && ++nCuts_ == testFailAt_ && job_->rc == VO_OK) job_->rc = VO_E_UNSUPPORTED;
            if (job_->rc == VO_OK) SolveResident(*job_, ctx_);
            job_->solveMs = ms_since(t1);
            job_->done = true; Finish();
        } else {
            EnsureWorker();
            { std::unique_lock<std::mutex> lk(mu_); hasWork_ = true; workSeq_.fetch_add(1, std::memory_order_release); }
            cv_.notify_all();
        }
        if (!deferTail) FinishTail();
        stats_.ms += ms_since(t0);
        return;
    }
    FinishTail();
    Build(*j, keyframeCurr);
    stats_.ms_build += ms_since(t0);
    if (j->edgePose.empty() || j->nFree == 0) return;
    job_ = std::move(j);
    if (lag_ == 0) { Solve(*job_, ctx_); job_->done = true; Finish(); }
    else {
        EnsureWorker();
        { std::unique_lock<std::mutex> lk(mu_); hasWork_ = true; workSeq_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
    }
    stats_.ms += ms_since(t0);
}

// Flatten: free poses = covisible keyframes + current (backend.cpp:36-59); points = non-outlier map points
// they observe (:62-81); one edge per observation, observers outside the free set are fixed (:88-135).
void Backend::Build(Job& j, const Frame::Ptr& kf) {
    VO_SCOPE("ba.build");
    MapManager& map = MapManager::GetInstance();
    const uint64_t stamp = ++buildStamp_;            // index scratch lives in the frames / map points: no hash maps
    auto covis = kf->GetCovisibleKeyframes();
    std::vector<size_t> freeIds(covis.begin(), covis.end());
    if ((int)freeIds.size() + 1 > maxFree_) {        // more covisible keyframes than one solve takes: the strongest stay free, the others
        auto w = kf->GetCovisibleKeyframeWeights();  // still constrain the points they observe, as fixed poses (edge loop below)
        std::sort(freeIds.begin(), freeIds.end(), [&](size_t a, size_t b) { const int wa = w[a], wb = w[b]; return wa != wb ? wa > wb : a > b; });
        freeIds.resize((size_t)maxFree_ - 1);
        ++stats_.capped;
    }
    freeIds.push_back(kf->GetId());
    std::sort(freeIds.begin(), freeIds.end());
    if (fixOldest_ && freeIds.size() > 1) freeIds.erase(freeIds.begin());      // gauge anchor (experiment key): it becomes one of the fixed observers below
    for (size_t id : freeIds) {
        auto f = map.GetKeyframe(id);
        if (f == nullptr) continue;
        f->baStamp_ = stamp; f->baIndex_ = (int)j.poseFrames.size();
        j.poseFrames.push_back(f.get());
    }
    j.nFree = (int)j.poseFrames.size();
    { VO_SCOPE("ba.build.points");
    j.points.reserve(lastPoints_ + lastPoints_ / 4 + 1024);
    for (int p = 0; p < j.nFree; ++p) {
        const std::vector<Frame::ObservedEntry>& ob = j.poseFrames[p]->Observed();
        for (size_t q = 0; q < ob.size(); ++q) {                                    // insertion order: deterministic without sorting
            if (q + 8 < ob.size()) __builtin_prefetch(ob[q + 8].mappoint);
            const Frame::ObservedEntry& e = ob[q];
            Mappoint* mp = e.mappoint;
            if (!e.alive || mp->baStamp_ == stamp || mp->outlier_) continue;
            mp->baStamp_ = stamp; mp->baIndex_ = (int)j.points.size();
            j.points.push_back(mp);
        }
    }
    }
    { VO_SCOPE("ba.build.edges");
    // edges are written through raw pointers into arrays sized from the previous graph (grown geometrically when short)
    size_t cap = lastEdges_ + lastEdges_ / 4 + 4096, ne = 0;
    j.edgePose.resize(cap); j.edgePoint.resize(cap); j.edgeUv.resize(2 * cap);
    int32_t* ep = j.edgePose.data(); int32_t* el = j.edgePoint.data(); float* uv = j.edgeUv.data();
    const size_t npts = j.points.size();
    for (size_t k = 0; k < npts; ++k) {
        // the loop chases pointers (point -> its observation array -> keyframe): fetch the arrays of the points ahead
        if (k + 12 < npts) __builtin_prefetch(j.points[k + 12]);
        if (k + 6 < npts) __builtin_prefetch(j.points[k + 6]->ObservationList().data());
        const std::vector<Mappoint::Observation>& obs = j.points[k]->ObservationList();     // keyframe-id order
        if (ne + obs.size() > cap) {
            cap = 2 * (ne + obs.size());
            j.edgePose.resize(cap); j.edgePoint.resize(cap); j.edgeUv.resize(2 * cap);
            ep = j.edgePose.data(); el = j.edgePoint.data(); uv = j.edgeUv.data();
        }
        for (const Mappoint::Observation& o : obs) {
            Frame* f = o.keyframe;
            if (f == nullptr) continue;
            if (f->baStamp_ != stamp) {                                             // observer outside the free set: fixed pose
                f->baStamp_ = stamp; f->baIndex_ = (int)j.poseFrames.size();
                j.poseFrames.push_back(f);
            }
            ep[ne] = f->baIndex_; el[ne] = (int32_t)k; uv[2 * ne] = o.pixel.x; uv[2 * ne + 1] = o.pixel.y;
            ++ne;
        }
    }
    j.edgePose.resize(ne); j.edgePoint.resize(ne); j.edgeUv.resize(2 * ne);
    lastEdges_ = j.edgePose.size(); lastPoints_ = j.points.size();
    }
    VO_SCOPE("ba.build.copy");
    j.poses.resize(12 * j.poseFrames.size()); j.pts.resize(3 * j.points.size());
    for (size_t p = 0; p < j.poseFrames.size(); ++p) j.poseFrames[p]->GetPose().to12(&j.poses[12 * p]);
    for (size_t k = 0; k < j.points.size(); ++k) { if (k + 8 < j.points.size()) __builtin_prefetch(j.points[k + 8]); const Vector3d& x = j.points[k]->PositionUnlocked(); j.pts[3 * k] = x[0]; j.pts[3 * k + 1] = x[1]; j.pts[3 * k + 2] = x[2]; }
    j.posesOut.resize(12 * (size_t)std::max(j.nFree, 1)); j.ptsOut.resize(3 * std::max<size_t>(j.points.size(), 1)); j.flags.resize(std::max<size_t>(j.edgePose.size(), 1));
}

Backend::GraphView Backend::DescribeGraph(const Frame::Ptr& kf) {
    Job j;
    Build(j, kf);
    GraphView g;
    for (Frame* f : j.poseFrames) g.poseIds.push_back(f->GetId());
    g.nFree = j.nFree;
    for (Mappoint* mp : j.points) g.pointIds.push_back(mp->GetId());
    g.edgePose = j.edgePose; g.edgePoint = j.edgePoint; g.edgeUv = j.edgeUv;
    return g;
}

void Backend::Solve(Job& j, vo_ctx* ctx) {
    auto t0 = std::chrono::steady_clock::now();
    vo_ba_problem prob;
    prob.n_poses = (int)j.poseFrames.size(); prob.n_free = j.nFree; prob.n_points = (int)j.points.size(); prob.n_edges = (int)j.edgePose.size();
    prob.poses = j.poses.data(); prob.points = j.pts.data(); prob.edge_pose = j.edgePose.data(); prob.edge_point = j.edgePoint.data(); prob.edge_uv = j.edgeUv.data();
    prob.huber_delta = std::sqrt(7.815); prob.chi2_th = chi2Threshold_; prob.it_robust = 10; prob.it_plain = 10;      // backend.cpp:83,141,159
    vo_ba_result res;
    std::memset(&res, 0, sizeof(res));
    res.poses = j.posesOut.data(); res.points = j.ptsOut.data(); res.edge_flags = j.flags.data();
    j.rc = vo_local_ba(ctx, &prob, &res);
    j.solveMs = ms_since(t0);
}

void Backend::WaitGraphCut() {
    if (!job_ || !job_->resident || lag_ == 0) return;
    const int seen = cutSeq_.load(std::memory_order_acquire);
    std::unique_lock<std::mutex> lk(mu_);
    SpinThenWait(lk, cutSeq_, seen, [&] { return job_->cutDone; });
}

void Backend::SolveResident(Job& j, vo_ctx* ctx) {
    if (j.nPoints == 0 || j.nEdges == 0) { j.nCulled = 0; return; }
    // (sized once and kept: an edge-sized vector per BA is 800 KB of page faults and an munmap on the latency chain; more culled observations
    // than this come back as VO_E_OVERFLOW and the BA is skipped)
    const size_t cullCap = std::min<size_t>((size_t)j.nEdges, 1 << 16);
    if (j.culled.size() < cullCap) j.culled.resize(cullCap);
    vo_ba_resident_result r;
    std::memset(&r, 0, sizeof(r));                   // poses / point_slots / points stay NULL: the result is merged on the device (Finish) and fetched later (FinishTail)
    r.culled_obs = j.culled.data(); r.cap_culled = (int)cullCap;
    j.rc = vo_local_ba_resident_solve(ctx, 10, 10, &r);                                  // backend.cpp:141,:159
    j.nCulled = r.n_culled; j.nPairs = r.n_pairs;
}

void Backend::Apply(Job& j) {
    VO_SCOPE("ba.apply");
    int outlierCnt = 0;
    const size_t ne = j.edgePose.size();
    for (size_t e = 0; e < ne; ++e) {                                         // backend.cpp:144-172
        if (!j.flags[e]) {                                                    // flagged edges are rare: skip clean runs 8 at a time
            uint64_t w;
            while (e + 8 <= ne && (std::memcpy(&w, &j.flags[e], 8), w == 0)) e += 8;
            if (e >= ne || !j.flags[e]) continue;
        }
        if (!(j.flags[e] & 3)) continue;
        Frame& f = *j.poseFrames[j.edgePose[e]];
        Mappoint& mp = *j.points[j.edgePoint[e]];
        if (f.IsObservedMappoint(mp.GetId())) f.RemoveObservedMappoint(mp.GetId());
        ++outlierCnt;
    }
    for (int p = 0; p < j.nFree; ++p) j.poseFrames[p]->SetPose(SE3::from12(&j.posesOut[12 * (size_t)p]));       // backend.cpp:183-187
    // backend.cpp:188-194.  The optimised positions are already one flat array: they go to the tracker's device map
    // in a single vo_map_upsert (positions only) instead of through the per-point dirty list.
    VO_SCOPE("ba.apply.points");
    const size_t np = j.points.size();
    applySlots_.resize(np); applyXyz_.resize(3 * np);
    size_t m = 0;
    for (size_t k = 0; k < np; ++k) {
        if (k + 8 < np) __builtin_prefetch(j.points[k + 8], 1);
        Mappoint& mp = *j.points[k];
        mp.optimized_ = true;                                                 // every point of the graph has at least one edge
        if (mp.outlier_) continue;
        const double* x = &j.ptsOut[3 * k];
        mp.SetPositionSyncedUnlocked(Vector3d(x[0], x[1], x[2]));
        applySlots_[m] = mp.slot_; applyXyz_[3 * m] = x[0]; applyXyz_[3 * m + 1] = x[1]; applyXyz_[3 * m + 2] = x[2];
        ++m;
    }
    if (m) {
        int rc = vo_map_upsert(ctx_, applySlots_.data(), applyXyz_.data(), nullptr, nullptr, nullptr, (int)m);
        if (rc != VO_OK) throw std::runtime_error(std::string("vo_map_upsert (BA merge) failed: ") + vo_strerror(rc));
    }
    stats_.runs++; stats_.poses = j.nFree; stats_.fixed = (int)j.poseFrames.size() - j.nFree; stats_.points = (int)j.points.size();
    stats_.edges = (int)j.edgePose.size(); stats_.outliers = outlierCnt; stats_.ms_solve += j.solveMs;
    { const double D = 6.0 * j.nFree; stats_.sum_d3 += D * D * D; stats_.sum_d2 += D * D; stats_.sum_edges += (long long)j.edgePose.size(); }
}

}  // namespace myslam
