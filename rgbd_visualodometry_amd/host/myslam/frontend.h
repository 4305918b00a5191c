// frontend.h -- tracking front-end (reference include/myslam/frontend.h:32-112,
// src/frontend.cpp).  Same public surface (AddFrame / SetViewer / SetBackend / GetState); the
// ORB, matching, PnP-RANSAC and pose-refinement numerics run behind the C-ABI of vo_hip.h.
#ifndef MYSLAM_FRONTEND_H
#define MYSLAM_FRONTEND_H
#include "myslam/backend.h"
#include "myslam/common_include.h"
#include "myslam/frame.h"
#include "myslam/mappoint.h"
#include "myslam/util.h"
#include "myslam/viewer.h"

namespace myslam {
class FrontEnd {
public:
    typedef std::shared_ptr<FrontEnd> Ptr;
    enum VOState { INITIALIZING = 0, TRACKING, LOST };

    FrontEnd();                         // parameters from Config (frontend.cpp:29-43); owns a vo_ctx on device 0
    explicit FrontEnd(int device, int width = 640, int height = 480, int max_frames = 1);
    ~FrontEnd();

    bool AddFrame(const Frame::Ptr frame);
    void SetViewer(const Viewer::Ptr viewer) { viewer_ = viewer; }
    void SetBackend(const Backend::Ptr backend) { backend_ = backend; if (backend_) { backend_->SetContext(ctx_, device_); backend_->SetFallbackHook([this]() { FallBackToHostObjects(); }); } }
    VOState GetState() const { return state_; }
    // Several processes (one per GPU) track the SAME stream: each scores the PnP-RANSAC hypotheses h % world == rank (src/frontend.cpp:238-241), one RCCL all-reduce
    // of the per-hypothesis inlier counts per pass, enqueued on the tracking chain's stream (vo_set_hypothesis_shard_stream with the native
    // exchange of myslam/rccl_exchange.h).  idFile: where rank 0 leaves the RCCL id for the other ranks.  Throws when RCCL cannot be loaded / initialised.
    void ShardHypothesesOverRanks(int rank, int world, const std::string& idFile);
    // the two per-frame decisions on explicit inputs (parity tests): bit0 IsGoodEstimation, bit1 IsKeyframe
    int PolicyFlags(const SE3& T_ref_cw, const SE3& T_cur_cw, int numInliers);

    // Look-ahead: upload (or bind) and run batched ORB for upcoming frames of this stream.
    // Frames of one stream depend on each other only from matching onwards, so detection and
    // description of up to max_frames future frames run as one batched launch chain.
    int PrefetchFrames(const std::vector<Frame::Ptr>& frames);
    // Start the host -> device copies of the frames of the NEXT PrefetchFrames call (same order, same buffers) beside the tracking of the
    // current ones: (pointer, stride) pairs of frames in page-locked host memory; no-op for anything else (vo_frames_preload)
    void PreloadFrames(const std::vector<const void*>& bgr, const std::vector<const void*>& depth, int bgr_stride, int depth_stride);

    vo_ctx* GetContext() const { return ctx_; }
    void JoinGroup(vo_group* g);         // this stream's tracking calls share launch chains with the group's other streams
    struct Stats { int frames = 0, keyframes = 0, lost = 0; int last_candidates = 0, last_matches = 0, last_ransac = 0, last_lm = 0, last_keypoints = 0;
                   double ms_extract = 0, ms_track = 0, ms_keyframe = 0, ms_backend = 0, ms_refresh = 0, ms_flush = 0;
                   long long triangulated = 0, reobserved = 0, tracked = 0, sum_active = 0, sum_cand = 0, sum_match = 0, sum_ransac = 0, sum_lm = 0, sum_lm_iters = 0, track_launches = 0; };
    const Stats& GetStats() const { return stats_; }
    bool verbose_ = false;

private:
    Viewer::Ptr  viewer_;
    Backend::Ptr backend_;
    VOState      state_;
    int          accuLostFrameNums_;
    Frame::Ptr   keyframeRef_, framePrev_, frameCurr_;
    std::unordered_map<size_t, Mappoint::Ptr> trackingMap_;
    Frame::Ptr   keyframeForTrackingMap_;
    bool         trackingMapChanged_ = true;

    vo_ctx*                 ctx_ = nullptr;
    vo_group*               group_ = nullptr;
    void*                   rcclComm_ = nullptr;     // ShardHypothesesOverRanks
    int                     device_ = 0;
    vo_params               params_;
    vo_track_params         trackParams_;
    std::vector<KeyPoint>   keypointsCurr_;
    std::vector<Descriptor> descriptorsCurr_;
    std::vector<Mappoint*> activeList_;                     // device tracking map, in matching order (owned by the MapManager)
    std::vector<int> activeIndexOfSlot_;
    // flannMatchedMptKptMap_ of the reference (frontend.h:73) as parallel arrays in match order
    std::vector<Mappoint*> flannMatchedMpt_; std::vector<int> flannMatchedKp_; std::vector<char> flannMatchedLm_;
    KeyPointSet flannMatchedKptSet_;
    std::vector<Mappoint*> pnpMatchedMpt_;                  // pnpMatchedMptSet_ (frontend.h:76), id order
    std::vector<int> pnpMatchedMptKp_;
    KeyPointSet pnpMatchedKptSet_;
    std::vector<Mappoint::Ptr> newMappoints_;
    std::vector<int32_t> newMappointKp_;                    // keypoint each of them was created from (frontend.cpp:390)
    int   numInliers_ = 0;
    float minDisRatio_; int maxLostFrames_, minInliers_; double keyFrameMinRot_, keyFrameMinTrans_;
    int   nextSlot_ = 0;
    int   lookahead_ = 1, scratchSlot_ = -1;        // frame slots [0, lookahead_) serve PrefetchFrames; the scratch slot re-detects old keyframes
    bool  triangulateAll_ = false, reobserveNew_ = false, deviceDescriptors_ = false;
    // speculative batch tracking: frames between keyframes share prior + map (frontend.cpp:96), so the frames that
    // follow the current one in the prefetch queue are tracked in the same launch chain; results are cached and
    // dropped when a keyframe / BA merge changes the inputs (epoch).
    struct SpecResult { size_t frameId; uint64_t epoch; vo_track_result res; int lane; };    // matches stay on the device (lane buffers)
    std::vector<SpecResult> spec_;
    // A launch chain started ahead of the frames it tracks (at a keyframe, before the BA write-back): the frames, the state it was started in
    struct Ahead { bool pending = false; uint64_t epoch = 0; std::vector<size_t> ids; std::vector<vo_track_result> res; } ahead_;
    bool trackAhead_ = true;
    void LaunchTrackAhead();        // start the next frames' chain now (vo_track_batch_begin)
    void DrainAhead();              // wait for a chain in flight and drop its results
    std::vector<Frame::Ptr> prefetched_;
    uint64_t epoch_ = 0;
    int trackBatch_ = 1, framesSinceKf_ = 0, lastInterval_ = 0; double lastMotion_ = 0;     // lastInterval_: frames between the last two keyframes
    Stats stats_;
    std::vector<vo_keypoint> kpBuf_; std::vector<uint8_t> descBuf_; std::vector<vo_match> matchBuf_;
    // The per-frame containers above (keypointsCurr_, flann*/pnp* lists) are only read on keyframes and by the viewer:
    // they are materialised on demand from the raw device results of the current frame.
    int nKeypointsCurr_ = 0; bool keypointsBuilt_ = false;
    int curLane_ = -1, nCurMatches_ = 0; bool matchesFetched_ = false, matchListsBuilt_ = false;    // lane of the last batch holding this frame's matches
    void EnsureKeypoints();
    void EnsureMatchLists();
    std::vector<int32_t> upIdx_; std::vector<double> upXyz_, upNrm_; std::vector<uint8_t> upDesc_, upFlags_;

    void Init(int device, int width, int height, int max_frames);
    void InitializationHandler();
    bool TrackingHandler();
    void LostHandler();
    void ExtractKeyPointsAndComputeDescriptors();
    void MatchAndEstimatePose();            // MatchKeyPointsInTrackingMap + EstimatePosePnP, coarse and fine
    void RefreshTrackingMap();
    void FlushDirtyMappoints();
    bool IsGoodEstimation();
    bool IsKeyframe();
    void AddCurrentKeyframeObservations();
    void CreateNewMappoints();
    void TriangulateMappointsInTrackingMap();
    void TriangulateAllBatched();                    // triangulate_all: 1 -- every eligible point, one vo_triangulate_batch call
    void RegisterKeyframeOnDevice();                 // device-resident bookkeeping: the new keyframe's pose and observations
    void AddNewMappointsObservationsForOldKeyframes();   // reobserve_new_mappoints: 1 (reference src/frontend.cpp:408-463, disabled there at :130)
    // device_keyframes: 1 -- the keyframe bookkeeping above (observations, covisibility, new map points, triangulation, local map) runs on the device
    // tables (vo_keyframe_commit, vo_map_set_active_covisible); the host keeps the keyframes' poses and covisibility ledgers, and no Mappoint objects
    // until somebody asks for them (MaterializeMap).  SURVEY 8f-2.
    bool deviceKeyframes_ = false, kfModeDecided_ = false, kfOnDevice_ = false;
    int nActive_ = 0;                                // size of the device's active list in that mode
    std::vector<int32_t> covisKf_, covisW_;
    bool UseDeviceKeyframes();
    void CommitKeyframeOnDevice();
    int ActiveCount() const { return kfOnDevice_ ? nActive_ : (int)activeList_.size(); }
public:
    // Mappoint objects, observation lists and Frame observation sets as of now, rebuilt from the device tables (device_keyframes; a no-op otherwise)
    void MaterializeMap();
    bool KeyframesOnDevice() const { return kfOnDevice_; }
    // device_keyframes given up for the rest of the run (a graph the device cut cannot take): host objects rebuilt from the tables, bookkeeping and
    // graph cut back on the host from the next keyframe on
    void FallBackToHostObjects();
private:
};
}  // namespace myslam
#endif
