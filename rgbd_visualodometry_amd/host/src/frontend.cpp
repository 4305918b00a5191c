// frontend.cpp -- tracking front-end behind the reference's FrontEnd surface
// (reference src/frontend.cpp:29-506).  State machine, keyframe policy and map bookkeeping stay
// on the host; ORB, matching, PnP-RANSAC and pose refinement run through the C-ABI (vo_hip.h).
#include "myslam/frontend.h"
#include "myslam/rccl_exchange.h"

#include <algorithm>
#include <chrono>
#include <stdexcept>

#include "myslam/config.h"
#include "myslam/mapmanager.h"
#include "myslam/util.h"

namespace myslam {

namespace {
struct StageTimer {
    double& acc; std::chrono::steady_clock::time_point t0;
    explicit StageTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

static void vo_check(int rc, const char* what) {
    if (rc != VO_OK) throw std::runtime_error(std::string(what) + " failed: " + vo_strerror(rc));
}

template <typename T>
static T cfg_or(const std::string& key, T dflt) { return Config::has(key) ? Config::get<T>(key) : dflt; }

FrontEnd::FrontEnd() { Init(cfg_or<int>("device", 0), cfg_or<int>("image.width", 640), cfg_or<int>("image.height", 480), cfg_or<int>("max_frames_in_flight", 1)); }
FrontEnd::FrontEnd(int device, int width, int height, int max_frames) { Init(device, width, height, max_frames); }

void FrontEnd::Init(int device, int width, int height, int max_frames) {
    state_ = INITIALIZING;
    device_ = device;
    accuLostFrameNums_ = 0;                                           // (uninitialised in the reference, frontend.h:59)
    vo_default_params(&params_);
    params_.width = width; params_.height = height;
    params_.fx = cfg_or<float>("camera.fx", params_.fx); params_.fy = cfg_or<float>("camera.fy", params_.fy);
    params_.cx = cfg_or<float>("camera.cx", params_.cx); params_.cy = cfg_or<float>("camera.cy", params_.cy);
    params_.depth_scale = cfg_or<float>("camera.depth_scale", params_.depth_scale);
    params_.n_features = cfg_or<int>("number_of_features", params_.n_features);            // frontend.cpp:35
    params_.scale_factor = (float)cfg_or<double>("scale_factor", params_.scale_factor);    // :36
    params_.n_levels = cfg_or<int>("level_pyramid", params_.n_levels);                     // :37
    lookahead_ = std::max(1, max_frames);
    triangulateAll_ = cfg_or<int>("triangulate_all", 0) != 0;
    reobserveNew_ = cfg_or<int>("reobserve_new_mappoints", 0) != 0;
    deviceDescriptors_ = cfg_or<int>("map_descriptors_on_device", 0) != 0;
    deviceKeyframes_ = cfg_or<int>("device_keyframes", 0) != 0;
    params_.max_frames = lookahead_ + (reobserveNew_ ? 1 : 0);
    scratchSlot_ = reobserveNew_ ? lookahead_ : -1;
    params_.map_capacity = cfg_or<int>("map_capacity", 1 << 22);      // 4 Mi map points to start with (~100 bytes each + 49 per tracking lane); vo_keyframe_commit doubles the arrays when a keyframe may not fit
    trackBatch_ = std::max(1, std::min(16, cfg_or<int>("track_batch", 1)));
    trackAhead_ = cfg_or<int>("track_ahead", 1) != 0;
    if (const char* e = std::getenv("VO_TRACK_AHEAD")) trackAhead_ = std::atoi(e) != 0;      // experiments
    params_.max_track_batch = trackBatch_;
    vo_default_track_params(&trackParams_);
    trackParams_.match_ratio = minDisRatio_ = cfg_or<float>("match_ratio", 2.0f);          // :38
    maxLostFrames_ = (int)cfg_or<float>("max_num_lost", 10.f);                             // :39
    minInliers_ = cfg_or<int>("min_inliers", 10);                                          // :40
    keyFrameMinRot_ = cfg_or<double>("keyframe_rotation", 0.05);                           // :41
    keyFrameMinTrans_ = cfg_or<double>("keyframe_translation", 0.05);                      // :42
    trackParams_.n_hyp = cfg_or<int>("ransac_iterations", 100);                            // :240
    params_.max_hypotheses = std::max(params_.max_hypotheses, trackParams_.n_hyp);
    vo_check(vo_ctx_create(&params_, device, &ctx_), "vo_ctx_create");
    if (backend_) backend_->SetContext(ctx_, device_);
}

FrontEnd::~FrontEnd() { if (ctx_) { try { DrainAhead(); } catch (...) {} if (group_) vo_group_leave(group_, ctx_); vo_ctx_destroy(ctx_); } if (rcclComm_) myslam_rccl_comm_destroy(rcclComm_); }

void FrontEnd::ShardHypothesesOverRanks(int rank, int world, const std::string& idFile) {
    if (world <= 1) { vo_check(vo_set_hypothesis_shard_stream(ctx_, 0, 1, nullptr, nullptr), "vo_set_hypothesis_shard_stream"); return; }
    char id[MYSLAM_RCCL_ID_BYTES];
    if (myslam_rccl_id_via_file(idFile.c_str(), rank, 120, id) || myslam_rccl_comm_create(id, rank, world, &rcclComm_))
        throw std::runtime_error(std::string("RCCL: ") + myslam_rccl_last_error());
    vo_check(vo_set_hypothesis_shard_stream(ctx_, rank, world, myslam_rccl_allreduce_i32, rcclComm_), "vo_set_hypothesis_shard_stream");
}

void FrontEnd::JoinGroup(vo_group* g) { vo_check(vo_group_join(g, ctx_), "vo_group_join"); group_ = g; }

bool FrontEnd::AddFrame(const Frame::Ptr frame) {
    if (verbose_) std::cout << "Frontend status: " << (state_ == INITIALIZING ? "Initializing" : state_ == TRACKING ? "Tracking" : "Lost") << std::endl;
    frameCurr_ = frame;
    ++stats_.frames;
    if (backend_) { StageTimer t(stats_.ms_backend); if (backend_->Poll((size_t)stats_.frames)) ++epoch_; }   // deterministic merge point of an overlapped BA
    switch (state_) {
        case INITIALIZING: InitializationHandler(); break;
        case TRACKING: if (!TrackingHandler()) return false; break;
        case LOST: LostHandler(); return false;
    }
    if (viewer_ && !kfOnDevice_) { EnsureMatchLists(); viewer_->setCurrentFrame(frameCurr_, flannMatchedKptSet_); viewer_->updateDrawingObjects(); }      // (a viewer set after the first keyframe of a device_keyframes run gets no per-frame match sets: the mode is fixed there)
    return true;
}

// The mode is fixed at the first keyframe: everything it needs must be there (device graph cut, descriptors on the device, the reference's
// default triangulation policy, nobody who reads host map objects per frame).
bool FrontEnd::UseDeviceKeyframes() {
    if (!kfModeDecided_) {
        kfOnDevice_ = deviceKeyframes_ && deviceDescriptors_ && backend_ && backend_->DeviceGraph() && !reobserveNew_ && !triangulateAll_ && !viewer_;
        kfModeDecided_ = true;
        if (backend_) backend_->SetDeviceKeyframes(kfOnDevice_);
    }
    return kfOnDevice_;
}

void FrontEnd::InitializationHandler() {
    ExtractKeyPointsAndComputeDescriptors();
    if (UseDeviceKeyframes()) {
        MapManager::GetInstance().InsertKeyframe(frameCurr_);
        ++stats_.keyframes;
        curLane_ = -1;                                           // no matches yet
        CommitKeyframeOnDevice();
        state_ = TRACKING;
        framePrev_ = frameCurr_;
        keyframeRef_ = frameCurr_;
        return;
    }
    EnsureMatchLists();                                          // no matches yet: empty lists sized for this frame
    MapManager::GetInstance().InsertKeyframe(frameCurr_);       // the first frame is a keyframe
    ++stats_.keyframes;
    CreateNewMappoints();                                        // one frame suffices: depth is measured
    if (backend_ && backend_->DeviceGraph()) RegisterKeyframeOnDevice();
    state_ = TRACKING;
    framePrev_ = frameCurr_;
    keyframeRef_ = frameCurr_;
}

bool FrontEnd::TrackingHandler() {
    frameCurr_->SetPose(framePrev_->GetPose());                  // prior = pose of the last keyframe (frontend.cpp:96)
    { StageTimer t(stats_.ms_extract); ExtractKeyPointsAndComputeDescriptors(); }
    { StageTimer t(stats_.ms_track); MatchAndEstimatePose(); }   // coarse + fine (frontend.cpp:100-108)

    if (!IsGoodEstimation()) {
        accuLostFrameNums_++;
        state_ = (++accuLostFrameNums_ > maxLostFrames_) ? LOST : TRACKING;      // double increment kept (:113-114)
        ++stats_.lost;
        return false;
    }
    accuLostFrameNums_ = 0;
    if (!IsKeyframe()) return true;

    ++epoch_; lastInterval_ = framesSinceKf_; framesSinceKf_ = 0;   // map + prior change: cached speculative results are void
    if (backend_) backend_->WaitGraphCut();                        // a device-side graph cut may still be reading the tables this keyframe changes
    if (kfOnDevice_) {
        StageTimer t(stats_.ms_keyframe);
        MapManager::GetInstance().InsertKeyframe(frameCurr_);
        ++stats_.keyframes;
        { VO_SCOPE("kf.commit"); CommitKeyframeOnDevice(); }
    } else {
        StageTimer t(stats_.ms_keyframe);
        { VO_SCOPE("kf.lists"); EnsureMatchLists(); }
        { VO_SCOPE("kf.insert"); MapManager::GetInstance().InsertKeyframe(frameCurr_); }
        ++stats_.keyframes;
        { VO_SCOPE("kf.add_observations"); AddCurrentKeyframeObservations(); }
        { VO_SCOPE("kf.create_mappoints"); CreateNewMappoints(); }
        if (reobserveNew_) { VO_SCOPE("kf.reobserve"); AddNewMappointsObservationsForOldKeyframes(); }
        { VO_SCOPE("kf.triangulate"); if (triangulateAll_) TriangulateAllBatched(); else TriangulateMappointsInTrackingMap(); }
    }
    if (!kfOnDevice_ && backend_ && backend_->DeviceGraph()) { StageTimer t(stats_.ms_keyframe); VO_SCOPE("kf.register"); RegisterKeyframeOnDevice(); }
    const bool ahead = backend_ && backend_->DeviceGraph() && backend_->Lag() > 0 && trackAhead_ && !group_;
    if (!ahead) {
        if (backend_) { StageTimer t(stats_.ms_backend); backend_->OptimizeCovisibleGraphOfKeyframe(frameCurr_); }
        framePrev_ = frameCurr_;
        keyframeRef_ = frameCurr_;
        return true;
    }
    // Overlapped back-end with the graph on the device: the previous local BA is merged (host ledger + device tables), the map points that lost their
    // last observation to it are flushed before the next cut reads the tables, the next BA starts, the NEXT FRAMES' tracking chain is launched -- and
    // only then the merged result is copied into the host objects, while the GPU tracks.
    { StageTimer t(stats_.ms_backend); VO_SCOPE("kf.merge_pending"); backend_->MergePending(); }
    { StageTimer t(stats_.ms_flush); FlushDirtyMappoints(); }
    { StageTimer t(stats_.ms_backend); VO_SCOPE("kf.start_ba"); backend_->OptimizeCovisibleGraphOfKeyframe(frameCurr_, true); }
    framePrev_ = frameCurr_;
    keyframeRef_ = frameCurr_;
    { StageTimer t(stats_.ms_track); VO_SCOPE("kf.track_ahead"); LaunchTrackAhead(); }
    { StageTimer t(stats_.ms_backend); backend_->FinishTailNow(); }
    return true;
}

void FrontEnd::LostHandler() { if (verbose_) std::cout << "Tracking is lost" << std::endl; }

int FrontEnd::PrefetchFrames(const std::vector<Frame::Ptr>& frames) {
    if ((int)frames.size() > lookahead_) throw std::runtime_error("PrefetchFrames: more frames than frame slots (max_frames_in_flight)");
    const int n = (int)frames.size();
    if (n <= 0) return 0;
    DrainAhead();                                                // a chain in flight reads the slots that are rebound below
    for (const Frame::Ptr& old : prefetched_) if (old) { old->orb_done_ = false; old->slot_ = -1; }      // their slots are rebound below
    for (int i = 0; i < n; ++i) {
        const Frame::Ptr& f = frames[i];
        if (f->color_.cols != params_.width || f->color_.rows != params_.height) throw std::runtime_error("frame size differs from the context's");
        if (f->color_.on_device) vo_check(vo_frame_bind_device(ctx_, i, f->color_.data, f->color_.stride, f->depth_.data, f->depth_.stride), "vo_frame_bind_device");
        else vo_check(vo_frame_upload(ctx_, i, (const uint8_t*)f->color_.data, f->color_.stride, (const uint16_t*)f->depth_.data, f->depth_.stride), "vo_frame_upload");
        f->slot_ = i; f->orb_done_ = false;
    }
    vo_check(vo_orb_detect_describe(ctx_, 0, n), "vo_orb_detect_describe");
    for (int i = 0; i < n; ++i) frames[i]->orb_done_ = true;
    prefetched_.assign(frames.begin(), frames.begin() + n);
    spec_.clear();                                               // slots were reassigned
    nextSlot_ = 0;
    return n;
}

void FrontEnd::PreloadFrames(const std::vector<const void*>& bgr, const std::vector<const void*>& depth, int bgr_stride, int depth_stride) {
    const int n = std::min<int>((int)std::min(bgr.size(), depth.size()), lookahead_);
    if (n < 1) return;
    vo_check(vo_frames_preload(ctx_, 0, n, (const uint8_t* const*)bgr.data(), bgr_stride, (const uint16_t* const*)depth.data(), depth_stride), "vo_frames_preload");
}

void FrontEnd::ExtractKeyPointsAndComputeDescriptors() {
    VO_SCOPE("fe.fetch_keypoints");
    Frame::Ptr f = frameCurr_;
    if (!f->orb_done_) {
        std::vector<Frame::Ptr> one{f};
        PrefetchFrames(one);
    }
    const int cap = 2 * params_.n_features + 64;
    int n = 0;
    vo_check(vo_orb_fetch(ctx_, f->slot_, nullptr, nullptr, cap, &n), "vo_orb_fetch");      // count only; records on demand
    nKeypointsCurr_ = std::min(n, cap);
    keypointsBuilt_ = false; matchListsBuilt_ = false; matchesFetched_ = false; curLane_ = -1; nCurMatches_ = 0;
    stats_.last_keypoints = nKeypointsCurr_;
}

void FrontEnd::EnsureKeypoints() {
    if (keypointsBuilt_) return;
    const int cap = 2 * params_.n_features + 64;
    kpBuf_.resize(cap); descBuf_.resize((size_t)32 * cap);
    int nf = 0;
    VO_SCOPE("fe.keypoints");
    // with the descriptors kept on the device (map_descriptors_on_device) only the keypoint records come down
    vo_check(vo_orb_fetch(ctx_, frameCurr_->slot_, kpBuf_.data(), deviceDescriptors_ ? nullptr : descBuf_.data(), cap, &nf), "vo_orb_fetch");
    const int n = nKeypointsCurr_;
    keypointsCurr_.resize(n); descriptorsCurr_.resize(n);
    for (int i = 0; i < n; ++i) {
        const vo_keypoint& k = kpBuf_[i];
        KeyPoint& o = keypointsCurr_[i];
        o.pt = Point2f(k.x, k.y); o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
        o.index = i; o.depth_raw = k.depth_raw;
        if (!deviceDescriptors_) std::memcpy(descriptorsCurr_[i].data(), &descBuf_[(size_t)32 * i], 32);
    }
    keypointsBuilt_ = true;
}

// flannMatchedMptKptMap_ / pnpMatchedMptSet_ / pnpMatchedKptSet_ of the reference (frontend.cpp:208-209, :326-328)
void FrontEnd::EnsureMatchLists() {
    if (matchListsBuilt_) return;
    EnsureKeypoints();
    if (!matchesFetched_ && curLane_ >= 0 && nCurMatches_ > 0) {        // the records of this frame are still in its lane's device buffer
        if ((int)matchBuf_.size() < nCurMatches_) matchBuf_.resize(nCurMatches_);
        int got = 0;
        VO_SCOPE("fe.fetch_matches");
        vo_check(vo_track_fetch_matches(ctx_, curLane_, matchBuf_.data(), nCurMatches_, &got), "vo_track_fetch_matches");
        nCurMatches_ = got;
    }
    matchesFetched_ = true;
    flannMatchedMpt_.clear(); flannMatchedKp_.clear(); flannMatchedLm_.clear();
    flannMatchedKptSet_.reset(keypointsCurr_.size()); pnpMatchedKptSet_.reset(keypointsCurr_.size());
    pnpMatchedMpt_.clear(); pnpMatchedMptKp_.clear();
    for (int i = 0; i < nCurMatches_; ++i) {
        const vo_match& m = matchBuf_[i];
        Mappoint* mp = activeList_[activeIndexOfSlot_[m.map_index]];
        const KeyPoint& kp = keypointsCurr_[m.kp_index];
        flannMatchedMpt_.push_back(mp); flannMatchedKp_.push_back(m.kp_index);
        flannMatchedLm_.push_back((m.flags & VO_MATCH_LM_INLIER) ? 1 : 0);
        flannMatchedKptSet_.insert(kp);
        if (m.flags & VO_MATCH_LM_INLIER) { pnpMatchedMpt_.push_back(mp); pnpMatchedMptKp_.push_back(m.kp_index); pnpMatchedKptSet_.insert(kp); }
    }
    matchListsBuilt_ = true;
}

void FrontEnd::RefreshTrackingMap() {
    if (kfOnDevice_) {                                                              // the local-map query runs on the device tables (src/mapmanager.cpp:14-38, src/frontend.cpp:159-166)
        if (keyframeForTrackingMap_ == keyframeRef_) return;
        keyframeForTrackingMap_ = keyframeRef_;
        MapManager& map = MapManager::GetInstance();
        std::vector<int32_t> kfs{keyframeRef_->kfIndex_};
        for (size_t id : keyframeRef_->GetCovisibleKeyframes()) { Frame::Ptr f = map.GetKeyframe(id); if (f && f->kfIndex_ >= 0) kfs.push_back(f->kfIndex_); }
        std::sort(kfs.begin(), kfs.end());
        if (backend_) backend_->WaitGraphCut();
        int32_t n = 0;
        VO_SCOPE("fe.active_covisible");
        vo_check(vo_map_set_active_covisible(ctx_, kfs.data(), (int)kfs.size(), 100, map.NextSlot(), &n), "vo_map_set_active_covisible");
        nActive_ = n;
        return;
    }
    bool changed = false;
    if (keyframeForTrackingMap_ != keyframeRef_) {                                  // frontend.cpp:159-162
        keyframeForTrackingMap_ = keyframeRef_;
        activeList_ = MapManager::GetInstance().CollectMappointsAroundKeyframe(keyframeRef_);
        changed = true;
    }
    if (activeList_.size() < 100) {                                                 // frontend.cpp:163-166
        const auto& all = MapManager::GetInstance().AllMappointsOrdered();
        activeList_.resize(all.size());
        for (size_t i = 0; i < all.size(); ++i) activeList_[i] = all[i].get();
        changed = true;
    }
    if (!changed) return;
    std::vector<int32_t> slots(activeList_.size());
    int maxSlot = -1;
    for (auto& mp : activeList_) maxSlot = std::max(maxSlot, mp->slot_);
    activeIndexOfSlot_.assign((size_t)maxSlot + 1, -1);
    for (size_t i = 0; i < activeList_.size(); ++i) { slots[i] = activeList_[i]->slot_; activeIndexOfSlot_[slots[i]] = (int)i; }
    vo_check(vo_map_set_active(ctx_, slots.data(), (int)slots.size()), "vo_map_set_active");
}

void FrontEnd::FlushDirtyMappoints() {
    std::vector<Mappoint*> dirty = MapManager::GetInstance().TakeDirty();
    if (dirty.empty()) return;
    if (backend_) backend_->WaitGraphCut();                 // the cut reads positions and outlier flags of the map as of its keyframe
    const size_t n = dirty.size();
    upIdx_.resize(n); upXyz_.resize(3 * n); upNrm_.resize(3 * n); upDesc_.resize(32 * n); upFlags_.resize(n);
    for (size_t i = 0; i < n; ++i) {
        Mappoint* mp = dirty[i];
        mp->dirty_ = false;
        upIdx_[i] = mp->slot_;
        Vector3d p = mp->GetPosition(), nr = mp->GetNormDirection();
        for (int a = 0; a < 3; ++a) { upXyz_[3 * i + a] = p[a]; upNrm_[3 * i + a] = nr[a]; }
        std::memcpy(&upDesc_[32 * i], mp->descriptor_.data(), 32);
        upFlags_[i] = mp->outlier_ ? VO_MAP_FLAG_OUTLIER : 0;
    }
    vo_check(vo_map_upsert(ctx_, upIdx_.data(), upXyz_.data(), upNrm_.data(), deviceDescriptors_ ? nullptr : upDesc_.data(), upFlags_.data(), (int)n), "vo_map_upsert");
}

void FrontEnd::DrainAhead() {
    if (!ahead_.pending) return;
    ahead_.pending = false;
    ahead_.res.resize(ahead_.ids.size());
    vo_check(vo_track_batch_end(ctx_, ahead_.res.data()), "vo_track_batch_end");
}

// The frames after the current one (a keyframe: the prior and the map they will be tracked against are final) get their launch chain now; the batch is
// put together exactly as MatchAndEstimatePose would at the next AddFrame (same frames, same seeds), which then only collects the results.
void FrontEnd::LaunchTrackAhead() {
    if (ahead_.pending || group_) return;
    size_t pos = 0;
    while (pos < prefetched_.size() && prefetched_[pos] != frameCurr_) ++pos;
    if (pos + 1 >= prefetched_.size() || !prefetched_[pos + 1]->orb_done_) return;
    { StageTimer t(stats_.ms_refresh); RefreshTrackingMap(); }
    FlushDirtyMappoints();
    double prior[12];
    framePrev_->GetPose().to12(prior);                           // = what TrackingHandler gives the next frame as its prior (frontend.cpp:96)
    const size_t nextFrames = (size_t)stats_.frames + 1;         // AddFrame's counter when the first of these frames arrives
    const size_t nextMerge = backend_ ? backend_->NextMergeFrame() : (size_t)-1;
    int want = lastInterval_ > 0 ? std::min(trackBatch_, lastInterval_ + 1) : trackBatch_;      // framesSinceKf_ == 0 here (a frame more changed nothing: profiles/r06_spec_margin_ab.txt)
    std::vector<Frame::Ptr> batch{prefetched_[pos + 1]};
    for (size_t j = pos + 2; j < prefetched_.size() && (int)batch.size() < want; ++j) {
        if (!prefetched_[j]->orb_done_ || nextFrames + batch.size() >= nextMerge) break;
        batch.push_back(prefetched_[j]);
    }
    const int nb = (int)batch.size(), cap = ActiveCount() + 1;
    std::vector<int> slots(nb); std::vector<uint64_t> seeds(nb);
    ahead_.ids.resize(nb);
    for (int j = 0; j < nb; ++j) { slots[j] = batch[j]->slot_; seeds[j] = 0x5eed5eedull + 2 * (uint64_t)(nextFrames + j); ahead_.ids[j] = batch[j]->GetId(); }
    const int rc = vo_track_batch_begin(ctx_, nb, slots.data(), prior, &trackParams_, seeds.data(), cap);
    if (rc == VO_E_UNSUPPORTED) { trackAhead_ = false; return; }
    vo_check(rc, "vo_track_batch_begin");
    ahead_.pending = true; ahead_.epoch = epoch_;
}

void FrontEnd::MatchAndEstimatePose() {
    vo_track_result res;
    bool have = false;
    if (ahead_.pending) {                                        // the chain launched at the last keyframe
        const uint64_t ep = ahead_.epoch;
        DrainAhead();
        if (!ahead_.ids.empty() && ahead_.ids[0] == frameCurr_->GetId() && ep == epoch_) {
            spec_.clear();
            for (size_t j = 1; j < ahead_.ids.size(); ++j) { SpecResult sp; sp.frameId = ahead_.ids[j]; sp.epoch = epoch_; sp.res = ahead_.res[j]; sp.lane = (int)j; spec_.push_back(sp); }
            res = ahead_.res[0]; curLane_ = 0; have = true;
            ++stats_.track_launches;
        }
    }
    if (!have) for (auto& sp : spec_)
        if (sp.frameId == frameCurr_->GetId() && sp.epoch == epoch_) {            // tracked ahead of time with identical inputs
            res = sp.res;
            curLane_ = sp.lane;                                                   // lane buffers live until the next batch call
            have = true;
            break;
        }
    if (!have) {
        spec_.clear();
        { StageTimer t(stats_.ms_refresh); RefreshTrackingMap(); }
        { StageTimer t(stats_.ms_flush); FlushDirtyMappoints(); }
        double prior[12];
        frameCurr_->GetPose().to12(prior);
        // batch = this frame + the prefetched frames that follow it, while they see the same map: stop before a
        // scheduled BA merge and before the predicted next keyframe
        std::vector<Frame::Ptr> batch{frameCurr_};
        size_t pos = 0;
        while (pos < prefetched_.size() && prefetched_[pos] != frameCurr_) ++pos;
        const size_t nextMerge = backend_ ? backend_->NextMergeFrame() : (size_t)-1;
        // how many frames are likely to see this map: right after a keyframe the previous keyframe interval is the
        // estimate (lanes tracked beyond the next keyframe are thrown away), later the motion since the keyframe
        int want = lastInterval_ > 0 ? std::min(trackBatch_, lastInterval_ + 1) : trackBatch_;
        if (framesSinceKf_ > 0 && lastMotion_ > 0) {
            const double left = (1.0 - lastMotion_) / (lastMotion_ / framesSinceKf_);      // frames until a threshold is reached
            // (+ 2, not + 1: the estimate is a straight line through the motion so far; a chain that stops one frame short of the keyframe costs a second chain --
            // ~0.5 ms of latency -- while a lane too many costs one more workgroup per kernel: the driver's 20-step form 1762 -> 1846 frames/s, 300 steps +1 %, same poses)
            want = std::max(1, std::min(trackBatch_, (int)left + 2));
        }
        for (size_t j = pos + 1; j < prefetched_.size() && (int)batch.size() < want; ++j) {
            if (!prefetched_[j]->orb_done_ || (size_t)stats_.frames + batch.size() >= nextMerge) break;
            batch.push_back(prefetched_[j]);
        }
        const int nb = (int)batch.size(), cap = ActiveCount() + 1;
        std::vector<int> slots(nb); std::vector<uint64_t> seeds(nb); std::vector<vo_track_result> rs(nb);
        for (int j = 0; j < nb; ++j) { slots[j] = batch[j]->slot_; seeds[j] = 0x5eed5eedull + 2 * (uint64_t)(stats_.frames + j); }
        // match records are not copied back here: only keyframes (and the viewer) read them, through vo_track_fetch_matches
        ++stats_.track_launches;
        { VO_SCOPE("fe.vo_track_batch"); vo_check(vo_track_batch(ctx_, nb, slots.data(), prior, &trackParams_, seeds.data(), rs.data(), nullptr, cap), "vo_track_batch"); }
        for (int j = 1; j < nb; ++j) {
            SpecResult sp; sp.frameId = batch[j]->GetId(); sp.epoch = epoch_; sp.res = rs[j]; sp.lane = j;
            spec_.push_back(sp);
        }
        res = rs[0];
        curLane_ = 0;
    }
    if (res.status != VO_OK) throw std::runtime_error(std::string("device pipeline: ") + vo_strerror(res.status));
    matchListsBuilt_ = false; matchesFetched_ = false;
    nCurMatches_ = std::min(res.n_matches, ActiveCount() + 1);
    numInliers_ = res.n_ransac_inliers;                                             // frontend.cpp:242
    frameCurr_->SetPose(SE3::from12(res.T_cw));                                     // frontend.cpp:312
    stats_.last_candidates = res.n_candidates; stats_.last_matches = res.n_matches;
    stats_.last_ransac = res.n_ransac_inliers; stats_.last_lm = res.n_lm_inliers;
    ++stats_.tracked; stats_.sum_active += (long long)ActiveCount(); stats_.sum_cand += res.n_candidates; stats_.sum_match += res.n_matches;
    stats_.sum_ransac += res.n_ransac_inliers; stats_.sum_lm += res.n_lm_inliers; stats_.sum_lm_iters += res.lm_iters;
    if (verbose_)
        std::cout << "  tracking map " << ActiveCount() << ", candidates " << res.n_candidates << ", matches " << res.n_matches
                  << ", PnP inliers " << res.n_ransac_inliers << ", LM inliers " << res.n_lm_inliers << std::endl;
}

bool FrontEnd::IsGoodEstimation() {
    if (numInliers_ < minInliers_) return false;                                    // frontend.cpp:337-341
    SE3 T_r_c = framePrev_->GetPose() * frameCurr_->GetPose().inverse();
    Vector6d d = T_r_c.log();
    double n2 = 0;
    for (double v : d) n2 += v * v;
    return !(std::sqrt(n2) > 5.0);                                                  // frontend.cpp:345
}

bool FrontEnd::IsKeyframe() {
    SE3 T_r_c = framePrev_->GetPose() * frameCurr_->GetPose().inverse();
    Vector6d d = T_r_c.log();
    const double trans = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double rot = std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    ++framesSinceKf_;
    lastMotion_ = std::max(rot / keyFrameMinRot_, trans / keyFrameMinTrans_);   // fraction of the keyframe threshold used up
    return rot > keyFrameMinRot_ || trans > keyFrameMinTrans_;                      // frontend.cpp:359
}

int FrontEnd::PolicyFlags(const SE3& T_ref_cw, const SE3& T_cur_cw, int numInliers) {
    Frame::Ptr ref = Frame::CreateFrame(0, nullptr, Mat(), Mat()), cur = Frame::CreateFrame(0, nullptr, Mat(), Mat());
    ref->SetPose(T_ref_cw); cur->SetPose(T_cur_cw);
    std::swap(framePrev_, ref); std::swap(frameCurr_, cur);
    const int keepInl = numInliers_, keepSince = framesSinceKf_; const double keepMotion = lastMotion_;
    numInliers_ = numInliers;
    const int flags = (IsGoodEstimation() ? 1 : 0) | (IsKeyframe() ? 2 : 0);
    numInliers_ = keepInl; framesSinceKf_ = keepSince; lastMotion_ = keepMotion;
    std::swap(framePrev_, ref); std::swap(frameCurr_, cur);
    return flags;
}

void FrontEnd::AddCurrentKeyframeObservations() {
    // reference iterates an unordered_set (frontend.cpp:366-370); active-list (= id) order here, for determinism
    frameCurr_->BeginCovisibilityBatch();
    for (size_t i = 0; i < pnpMatchedMpt_.size(); ++i)
        frameCurr_->AddObservedMappoint(pnpMatchedMpt_[i], keypointsCurr_[pnpMatchedMptKp_[i]].pt);
    frameCurr_->EndCovisibilityBatch();
}

// A frame becomes a keyframe with its bookkeeping done where the data already is (vo_keyframe_commit): observations of the LM inliers, their viewing
// directions, the covisibility weights (the ledger below receives them counted), the new map points from the unmatched keypoints with depth, the
// reference's first-success triangulation, pose and observations into the device tables.  Reference src/frontend.cpp:119-131.
void FrontEnd::CommitKeyframeOnDevice() {
    MapManager& map = MapManager::GetInstance();
    if (frameCurr_->kfIndex_ < 0) { frameCurr_->kfIndex_ = (int)map.kfByIndex_.size(); map.kfByIndex_.push_back(frameCurr_.get()); }
    FlushDirtyMappoints();                                   // (scenario-built points only: tracking creates none on the host in this mode)
    if (covisKf_.size() < 4096) { covisKf_.resize(4096); covisW_.resize(4096); }
    double T[12];
    frameCurr_->GetPose().to12(T);
    vo_kf_commit_result r;
    const int rc = vo_keyframe_commit(ctx_, curLane_, frameCurr_->slot_, frameCurr_->kfIndex_, T, map.NextSlot(), covisKf_.data(), covisW_.data(), (int)covisKf_.size(), &r);
    if (rc == VO_E_OVERFLOW) throw std::runtime_error("device map / observation tables full (raise map_capacity; device_keyframes keeps no host map to fall back to)");
    vo_check(rc, "vo_keyframe_commit");
    if (r.n_covisible_total > r.n_covisible) {               // more partners than the arrays (or the library's pinned block) hold: recount this keyframe's row from the tables
        covisKf_.resize((size_t)r.n_covisible_total + 64); covisW_.resize(covisKf_.size());
        int32_t n = 0;
        vo_check(vo_kf_covisibility(ctx_, frameCurr_->kfIndex_, covisKf_.data(), covisW_.data(), (int)covisKf_.size(), &n), "vo_kf_covisibility");
        r.n_covisible = n;
    }
    map.ReserveSlots(r.n_new);
    for (int i = 0; i < r.n_covisible; ++i) {                // allCovisibleKeyframeIdToWeight_ / activeCovisibleKeyframes_ of both sides (src/frame.cpp:104-119, :157-171)
        Frame* partner = (size_t)covisKf_[i] < map.kfByIndex_.size() ? map.kfByIndex_[covisKf_[i]] : nullptr;
        if (partner && partner != frameCurr_.get()) frameCurr_->SetCovisibleWeightBoth(partner, covisW_[i]);
    }
    if (r.triangulated_slot >= 0) ++stats_.triangulated;
    if (verbose_) std::cout << "Created new mappoints: " << r.n_new << "\n  Triangulate active mappoints size: " << (r.triangulated_slot >= 0 ? 1 : 0) << std::endl;
}

void FrontEnd::MaterializeMap() {
    if (!kfOnDevice_) return;
    if (backend_) backend_->WaitGraphCut();
    DrainAhead();
    MapManager::GetInstance().MaterializeFromTables(ctx_);
}

void FrontEnd::FallBackToHostObjects() {
    if (!kfOnDevice_) return;
    std::cerr << "[myslam] device_keyframes: going back to host map objects (rebuilt from the device tables)" << std::endl;
    DrainAhead();
    MapManager::GetInstance().MaterializeFromTables(ctx_);
    kfOnDevice_ = false;
    keyframeForTrackingMap_ = nullptr;                       // the active list is built from the host objects again (and uploaded) at the next frame
    activeList_.clear(); activeIndexOfSlot_.clear();
    spec_.clear(); ++epoch_;
}

void FrontEnd::CreateNewMappoints() {
    newMappoints_.clear(); newMappointKp_.clear();
    for (size_t idx = 0; idx < keypointsCurr_.size(); ++idx) {
        const KeyPoint& kp = keypointsCurr_[idx];
        if (pnpMatchedKptSet_.count(kp)) continue;                                  // already explained by the map
        double depth = frameCurr_->GetDepth(kp);
        if (depth < 0) continue;
        Vector3d pos = frameCurr_->camera_->Pixel2World(kp, frameCurr_->GetPose(), depth);
        Mappoint::Ptr mpt = Mappoint::CreateMappoint(pos, descriptorsCurr_[idx]);
        MapManager::GetInstance().InsertMappoint(mpt);
        if (mpt->slot_ >= params_.map_capacity) throw std::runtime_error("device map capacity exceeded (raise map_capacity)");
        frameCurr_->AddObservedMappoint(mpt.get(), kp.pt);
        newMappoints_.push_back(mpt); newMappointKp_.push_back((int32_t)idx);
    }
    if (deviceDescriptors_ && !newMappoints_.empty()) {
        // the descriptor row of each new point is copied on the device from this frame's ORB results (its slot is still bound);
        // position, normal and flags travel with the next flush of the dirty list as usual
        const size_t n = newMappoints_.size();
        std::vector<int32_t> kp(n), slots(n);
        for (size_t i = 0; i < n; ++i) { kp[i] = newMappointKp_[i]; slots[i] = newMappoints_[i]->slot_; }
        vo_check(vo_map_upsert_from_frame(ctx_, frameCurr_->slot_, kp.data(), slots.data(), nullptr, nullptr, nullptr, (int)n), "vo_map_upsert_from_frame");
    }
    if (verbose_) std::cout << "Created new mappoints: " << newMappoints_.size() << std::endl;
}

// The new keyframe enters the device-resident tables the local BA's graph is cut from: its map points (positions, flags --
// the dirty list, flushed now instead of at the next frame), its pose, and one observation record per observed map point
// (Frame::AddObservedMappoint, reference src/frame.cpp:93-120); the registry maps observation ids back to host objects.
void FrontEnd::RegisterKeyframeOnDevice() {
    MapManager& map = MapManager::GetInstance();
    if (frameCurr_->kfIndex_ < 0) { frameCurr_->kfIndex_ = (int)map.kfByIndex_.size(); map.kfByIndex_.push_back(frameCurr_.get()); }
    FlushDirtyMappoints();
    const std::vector<Frame::ObservedEntry>& ob = frameCurr_->Observed();
    std::vector<int32_t> kf(ob.size(), frameCurr_->kfIndex_), mp(ob.size()); std::vector<float> uv(2 * ob.size());
    size_t n = 0;
    for (const Frame::ObservedEntry& e : ob) {
        if (!e.alive) continue;
        const Mappoint::Observation& o = e.mappoint->ObservationList().back();          // this keyframe's observation was appended last
        mp[n] = e.mappoint->slot_; uv[2 * n] = o.pixel.x; uv[2 * n + 1] = o.pixel.y;
        map.obsRegistry_.push_back(MapManager::ObsRef{frameCurr_.get(), e.mappoint});
        ++n;
    }
    int64_t first = 0;
    double T[12];
    frameCurr_->GetPose().to12(T);
    int rc = vo_obs_append(ctx_, kf.data(), mp.data(), uv.data(), (int)n, &first);
    if (rc == VO_OK) rc = vo_kf_set_pose(ctx_, &frameCurr_->kfIndex_, T, 1);
    if (rc == VO_E_OVERFLOW || rc == VO_E_INVALID || rc == VO_E_NOMEM) {
        // the tables are full (256 Mi observations / 64 Ki keyframes) or could not grow: from here on the back-end cuts its graphs on the host again;
        // a job already cut from the tables still merges through them
        std::cerr << "[myslam] device observation table full (" << vo_strerror(rc) << "): local BA graphs are cut on the host from keyframe " << frameCurr_->GetId() << " on" << std::endl;
        map.obsRegistry_.resize(map.obsRegistry_.size() - n);
        backend_->SetDeviceGraph(false);
        return;
    }
    vo_check(rc, "vo_obs_append / vo_kf_set_pose");
    if ((size_t)first + n != map.obsRegistry_.size()) throw std::runtime_error("observation registry out of step with the device table");
}

// Every eligible point of the keyframe in one batched launch (vo_triangulate_batch): the reference's loop stops after the
// first success (src/frontend.cpp:501 -- kept as the default policy above); `triangulate_all: 1` applies them all.
void FrontEnd::TriangulateAllBatched() {
    std::vector<Mappoint*> pts; std::vector<int32_t> vs{0}; std::vector<double> T, xy;
    for (Mappoint* mp : pnpMatchedMpt_) {
        if (mp->outlier_ || mp->triangulated_ || mp->optimized_) continue;
        const size_t before = xy.size();
        for (const Mappoint::Observation& o : mp->ObservationList()) {
            if (o.keyframe == nullptr) continue;
            double p12[12];
            o.keyframe->GetPose().to12(p12);
            T.insert(T.end(), p12, p12 + 12);
            const Vec3 pc = o.keyframe->camera_->Pixel2Camera(o.pixel);
            xy.push_back(pc[0]); xy.push_back(pc[1]);
        }
        if ((xy.size() - before) / 2 < 2) { xy.resize(before); T.resize(before / 2 * 12); continue; }
        pts.push_back(mp); vs.push_back((int32_t)(xy.size() / 2));
    }
    if (pts.empty()) return;
    std::vector<double> xyz(3 * pts.size()); std::vector<uint8_t> ok(pts.size());
    vo_check(vo_triangulate_batch(ctx_, (int)pts.size(), vs.data(), T.data(), xy.data(), xyz.data(), ok.data()), "vo_triangulate_batch");
    int cnt = 0;
    for (size_t i = 0; i < pts.size(); ++i)
        if (ok[i] && xyz[3 * i + 2] > 0) { pts[i]->SetPosition(Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])); pts[i]->triangulated_ = true; ++cnt; }
    stats_.triangulated += cnt;
    if (verbose_) std::cout << "  Triangulate active mappoints size: " << cnt << " (batched, " << pts.size() << " candidates)" << std::endl;
}

// The reference's disabled pass (src/frontend.cpp:408-463, call commented out at :130): every local keyframe is detected
// again and the map points just created are matched against it; like the reference it only COUNTS the matches that pass the
// distance gate (the lines that would add the observations are commented out there, :452-453).
void FrontEnd::AddNewMappointsObservationsForOldKeyframes() {
    if (newMappoints_.empty() || scratchSlot_ < 0) return;
    auto local = keyframeRef_->GetCovisibleKeyframes();
    local.insert(keyframeRef_->GetId());
    std::vector<size_t> ids(local.begin(), local.end());
    std::sort(ids.begin(), ids.end());
    FlushDirtyMappoints();                                  // the new points' descriptors and positions reach the device map
    std::vector<int32_t> slots(newMappoints_.size());
    for (size_t i = 0; i < newMappoints_.size(); ++i) slots[i] = newMappoints_[i]->slot_;
    vo_check(vo_map_set_active(ctx_, slots.data(), (int)slots.size()), "vo_map_set_active");
    long long matched = 0;
    for (size_t id : ids) {
        Frame::Ptr kf = MapManager::GetInstance().GetKeyframe(id);
        if (!kf || kf->color_.empty()) continue;
        if (kf->color_.on_device) vo_check(vo_frame_bind_device(ctx_, scratchSlot_, kf->color_.data, kf->color_.stride, kf->depth_.data, kf->depth_.stride), "vo_frame_bind_device");
        else vo_check(vo_frame_upload(ctx_, scratchSlot_, (const uint8_t*)kf->color_.data, kf->color_.stride, (const uint16_t*)kf->depth_.data, kf->depth_.stride), "vo_frame_upload");
        vo_check(vo_orb_detect_describe(ctx_, scratchSlot_, 1), "vo_orb_detect_describe");     // orb_->detectAndCompute(keyframe->color_, ...)
        double T[12];
        kf->GetPose().to12(T);
        int n = 0, ncand = 0, mind = 0;
        vo_check(vo_match_active_map(ctx_, scratchSlot_, T, trackParams_.match_ratio, trackParams_.match_floor, nullptr, 0, &n, &ncand, &mind), "vo_match_active_map");
        matched += n;
        if (verbose_) std::cout << " for keyframe " << id << " add " << n << " new observations" << std::endl;
    }
    stats_.reobserved += matched;
    keyframeForTrackingMap_ = nullptr;                      // the active list was borrowed: the next frame uploads the tracking map again
}

void FrontEnd::TriangulateMappointsInTrackingMap() {
    int triangulatedCnt = 0;
    std::vector<SE3> poses; std::vector<Vec3> points;
    const size_t nm = pnpMatchedMpt_.size();
    for (size_t i = 0; i < nm; ++i) {                                               // trackingMap_ ∩ pnpMatchedMptSet_, id order
        if (i + 8 < nm) __builtin_prefetch(pnpMatchedMpt_[i + 8]);                  // ~8000 objects, almost all skipped on their flags: the walk is cache misses
        Mappoint* mp = pnpMatchedMpt_[i];
        if (mp->outlier_ || mp->triangulated_ || mp->optimized_) continue;
        poses.clear(); points.clear();
        for (const Mappoint::Observation& o : mp->ObservationList()) {              // keyframe-id order
            if (o.keyframe == nullptr) continue;
            poses.push_back(o.keyframe->GetPose());
            points.push_back(o.keyframe->camera_->Pixel2Camera(o.pixel));
        }
        if (poses.size() >= 2) {
            Vec3 pworld = Vec3::Zero();
            if (Triangulation(poses, points, pworld) && pworld[2] > 0) {
                mp->SetPosition(pworld);
                mp->triangulated_ = true;
                triangulatedCnt++; ++stats_.triangulated;
                break;                                                              // frontend.cpp:501
            }
        }
    }
    if (verbose_) std::cout << "  Triangulate active mappoints size: " << triangulatedCnt << std::endl;
}

}  // namespace myslam
