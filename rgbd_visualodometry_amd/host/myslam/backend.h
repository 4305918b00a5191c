// backend.h -- local bundle adjustment over the covisibility graph of a new keyframe
// (reference include/myslam/backend.h:21-37, src/backend.cpp:19-195).  The graph is flattened on the
// caller's thread and solved by vo_local_ba on the GPU.
//
// Scheduling.  The reference solves on a worker thread and writes results back whenever it finishes
// (racing with the tracker).  Here the solve may also overlap tracking (own worker
// thread, own vo_ctx / HIP stream), but the result is merged at a DETERMINISTIC point: on the
// tracker thread, `lag` frames after the keyframe (or at the next keyframe, whichever comes first).
// lag = 0 (default) solves and merges synchronously inside OptimizeCovisibleGraphOfKeyframe.
#ifndef MYSLAM_BACKEND_H
#define MYSLAM_BACKEND_H
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>

#include "myslam/camera.h"
#include "myslam/common_include.h"
#include "myslam/frame.h"
#include "myslam/mapmanager.h"

namespace myslam {
class Backend {
public:
    typedef std::shared_ptr<Backend> Ptr;
    Backend(const Camera::Ptr camera);
    ~Backend();
    void SetContext(vo_ctx* ctx, int device);        // tracker's context (used when lag == 0)
    void SetLag(int frames) { lag_ = frames < 0 ? 0 : frames; if (ctx_ && lag_ > 0) EnsureWorker(); }
    void Stop();                                     // finish the pending job, join the worker
    void Flush() { if (job_) Finish(); FinishTail(); }
    // Device-resident graph: the back-end reads the tracker's device tables while it cuts the graph; the tracker calls this
    // before a keyframe's bookkeeping changes them (the cut is the first ~0.2 ms of a BA, a keyframe is >= 1 ms away).
    void WaitGraphCut();
    void SetDeviceGraph(bool on) { deviceGraph_ = on; }
    // The front-end keeps keyframes and map points on the device (FrontEnd with device_keyframes): a merge then also reports the covisibility
    // ledger's decrements and flags the graph's points on the device (vo_local_ba_resident_merge_ledger); no host map object is touched.
    void SetDeviceKeyframes(bool on) { deviceKeyframes_ = on; }
    // What to do when a device graph cut cannot take a keyframe's graph (VO_E_UNSUPPORTED / _OVERFLOW / _NOMEM) while the map lives on the device:
    // the front-end rebuilds its host objects from the tables and both go back to host bookkeeping and the host graph cut (FrontEnd::FallBackToHostObjects)
    void SetFallbackHook(std::function<void()> f) { onDeviceFallback_ = std::move(f); }
    bool DeviceGraph() const { return deviceGraph_; }             // wait for the pending job and merge it now (end of a sequence / of a timed region)
    void OptimizeCovisibleGraphOfKeyframe(const Frame::Ptr keyframeCurr, bool deferTail = false);
    // The three parts a caller may interleave with its own work (FrontEnd::TrackingHandler at a keyframe): the pending local BA is waited for and merged
    // (host ledger + device tables); the next one is started (OptimizeCovisibleGraphOfKeyframe with deferTail); the merged result reaches the host objects.
    void MergePending() { if (job_) Finish(true); }
    void FinishTailNow() { FinishTail(); }
    // tracker thread, once per frame before tracking: merge a finished/overdue job; true if the map changed
    bool Poll(size_t frameIndex);
    // frame index at which the pending job will be merged (SIZE_MAX: none pending)
    int Lag() const { return lag_; }
    size_t NextMergeFrame() const { return job_ ? job_->frameIndex + (size_t)lag_ : (size_t)-1; }
    struct Stats { int runs = 0, poses = 0, fixed = 0, points = 0, edges = 0, outliers = 0, failed = 0, capped = 0; double ms = 0, ms_build = 0, ms_solve = 0, ms_wait = 0, ms_wake = 0, ms_to_merge = 0; int waited = 0;
                   double sum_d3 = 0, sum_d2 = 0; long long sum_edges = 0, sum_points = 0, sum_pairs = 0; };
    const Stats& GetStats() const { return stats_; }
    // e-3: the local BA of every keyframe sharded over the ranks of a job by point, an RCCL all-reduce of the reduced system per LM step (vo_set_ba_shard_stream with
    // the native exchange of myslam/rccl_exchange.h).  The graph is then cut on the host (the sharded solve takes explicit problems).  Call before the first keyframe.
    void ShardOverRanks(int rank, int world, const std::string& idFile);
    // the flattened graph of a keyframe, for inspection (parity tests): what Solve would be handed
    struct GraphView { std::vector<size_t> poseIds; int nFree = 0; std::vector<size_t> pointIds; std::vector<int32_t> edgePose, edgePoint; std::vector<float> edgeUv; };
    GraphView DescribeGraph(const Frame::Ptr& keyframe);
private:
    struct Job {
        std::vector<Frame*> poseFrames; int nFree = 0;          // keyframes and map points are never removed from the map:
        std::vector<Mappoint*> points;                          // plain pointers stay valid for the job's lifetime
        std::vector<int32_t> edgePose, edgePoint; std::vector<float> edgeUv;
        std::vector<double> poses, pts, posesOut, ptsOut; std::vector<uint8_t> flags;
        size_t frameIndex = 0; int rc = 0; double solveMs = 0; bool done = false; std::chrono::steady_clock::time_point tDone;
        // device-resident graph: only the free keyframes' numbers go down, the cut happens on the device
        bool resident = false, cutDone = false; std::vector<int32_t> freeKf, pointSlots; std::vector<int64_t> culled; int nPoints = 0, nFixed = 0, nEdges = 0, nCulled = 0, nPairs = 0;
    };
    Camera::Ptr camera_;
    float chi2Threshold_;
    vo_ctx* ctx_ = nullptr;         // tracker context
    vo_ctx* ctxOwn_ = nullptr;      // worker's own context (lag > 0)
    void* rcclComm_ = nullptr; int shardRank_ = 0, shardWorld_ = 1;      // ShardOverRanks
    int device_ = 0, lag_ = 0;
    size_t frameIndex_ = 0;
    uint64_t buildStamp_ = 0;
    size_t lastEdges_ = 0, lastPoints_ = 0;          // reserve hints for the next graph
    std::vector<int32_t> applySlots_; std::vector<double> applyXyz_;      // merge scratch
    std::unique_ptr<Job> job_;
    std::thread worker_; std::mutex mu_; std::condition_variable cv_; bool quit_ = false, hasWork_ = false;
    // The two hand-offs of an overlapped BA (tracker -> worker: a job; worker -> tracker: done / graph cut done) are on the latency chain that bounds a
    // single stream: both sides poll these counters for a few hundred microseconds before they sleep on the condition variable.
    std::atomic<int> workSeq_{0}, doneSeq_{0}, cutSeq_{0};
    template <typename Pred> void SpinThenWait(std::unique_lock<std::mutex>& lk, std::atomic<int>& seq, int seen, Pred pred);
    Stats stats_;
    void Build(Job& j, const Frame::Ptr& kf);
    void Solve(Job& j, vo_ctx* ctx);
    void Apply(Job& j);
    void Finish(bool deferTail = false);   // wait for the pending job and merge it (deferTail: the host-side copy of a device-merged result is left for FinishTail)
    void FinishTail();              // positions and poses of a device-merged local BA reach the host objects (after the next graph cut has been started)
    std::unique_ptr<Job> tail_; vo_ctx* tailCtx_ = nullptr;
    void WorkerLoop();
    void EnsureWorker();            // the worker's context, stream and thread exist before the first keyframe (no one-time setup inside a timed run)
    bool deviceGraph_ = false, deviceKeyframes_ = false;
    std::function<void()> onDeviceFallback_;
    int testFailAt_ = 0, nCuts_ = 0;                // VO_TEST_FAIL_CUT_AT=k: the k-th device graph cut reports VO_E_UNSUPPORTED (tests of the fall-back paths)
    std::vector<int32_t> pairA_, pairB_;            // ledger decrements of a merge (device keyframes)
    std::vector<int64_t> culledSpare_;              // the culled-observation list's buffer, handed from job to job
    void FinishOnDevice(Job& j, vo_ctx* solver);
    bool fixOldest_ = false;        // ba_fix_oldest_free_keyframe: gauge-anchor experiment
    void SolveResident(Job& j, vo_ctx* ctx);
    int maxFree_ = 160;             // free-pose cap of one solve: the Cholesky of the reduced system is LDS resident (vo_local_ba: D = 6 n_free <= ~1050)
};
}  // namespace myslam
#endif
