/*
 * vo_hip.h -- C-ABI of the MI355X-native RGB-D visual-odometry hot path.
 *
 * This is the drop-in boundary.  The reference (BowenBZ/RGBD_VisualOdometry) has no
 * FFI of its own: its hot path is private methods of myslam::FrontEnd / Backend that
 * call OpenCV 3.1 / g2o directly.  The entry points below are cut exactly at those
 * third-party seams; every one names the reference call site it replaces.
 *
 *   vo_orb_detect_describe  <- cv::ORB::detectAndCompute          src/frontend.cpp:35-37,:153
 *   vo_match_active_map     <- Frame::IsCouldObserveMappoint loop  src/frontend.cpp:171-184, src/frame.cpp:70-91
 *                              + cv::FlannBasedMatcher::match      src/frontend.cpp:33,:187-211
 *   vo_pnp_ransac           <- cv::solvePnPRansac                  src/frontend.cpp:238-254
 *   vo_pose_refine_lm       <- g2o pose-only LM (2 x 10 iters)     src/frontend.cpp:257-329, include/myslam/g2o_types.h:47-108
 *   vo_track_frame          <- FrontEnd::TrackingHandler :102-108  (coarse + fine pass fused, one host sync)
 *   vo_track_batch          <- the same for several frames that share prior + map (one launch chain)
 *   vo_local_ba             <- Backend::Optimize                   src/backend.cpp:19-195, include/myslam/g2o_types.h:111-179
 *   vo_frame_upload         <- Frame::CreateFrame deep copy        src/frame.cpp:18-31
 *   vo_map_upsert/_set_active <- MapManager + trackingMap_         src/mapmanager.cpp:14-38, src/frontend.cpp:159-166
 *
 * Conventions: plain C, plain pointers and sizes.  Every function returns 0 (VO_OK) or a
 * negative vo_status; nothing throws across the boundary.  The caller owns all host
 * buffers; the context owns all device memory and one HIP stream.  One context per GPU
 * stream of frames; a context is not thread-safe (single caller thread, like AddFrame).
 * Poses are T_c_w (world -> camera) as 12 doubles: R row-major (9) then t (3).
 *
 * Two libraries export exactly these symbols: the product (libvo_hip.so, HIP/gfx950) and the
 * CPU oracle used only by tests / the CPU baseline (oracle/_build/liboracle_vo.so).
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

typedef enum vo_status {
    VO_OK = 0,
    VO_E_INVALID = -1,      /* bad argument / out-of-range slot or index          */
    VO_E_NOMEM = -2,        /* host or device allocation failed                    */
    VO_E_DEVICE = -3,       /* HIP runtime error (no device, launch failure, ...)  */
    VO_E_OVERFLOW = -4,     /* an internal fixed-capacity list overflowed          */
    VO_E_STATE = -5,        /* call sequence error (e.g. match before ORB)         */
    VO_E_UNSUPPORTED = -6
} vo_status;

/* Construction parameters.  Defaults mirror config/default.yaml:10-29 and the OpenCV-3.1
 * cv::ORB defaults the reference relies on (edgeThreshold 31, patchSize 31, FAST 20). */
typedef struct vo_params {
    int32_t width, height;          /* level-0 image size                                  */
    float fx, fy, cx, cy;           /* pinhole intrinsics, stored as float (camera.cpp:29-32) */
    float depth_scale;              /* raw depth units per metre (5000)                    */
    int32_t n_features;             /* number_of_features                                  */
    float scale_factor;             /* 1.2                                                 */
    int32_t n_levels;               /* 8                                                   */
    int32_t fast_threshold;         /* 20                                                  */
    int32_t edge_threshold;         /* 31 (>= 20: the steered BRIEF pattern reaches 19 pixels) */
    int32_t max_frames;             /* frame slots (frames in flight for batched ORB), >=1 */
    int32_t map_capacity;           /* device map capacity in points                        */
    int32_t max_hypotheses;         /* RANSAC hypothesis capacity (>= n_hyp ever passed)    */
    int32_t max_track_batch;        /* frames tracked concurrently by vo_track_batch (0/1 = one) */
    int32_t stream_priority;        /* class of the context's HIP stream: 0 = default, > 0 = highest, < 0 = lowest.  The runtime keeps a pool of
                                       hardware queues per class and streams of one class share its queues: the library's own pace-setting
                                       streams (BA engines, group chains) are in the highest class; a back-end's private context, which only
                                       prepares problems, takes the lowest so that its queues are not the trackers'  */
    int32_t reserved[6];
} vo_params;

/* cv::KeyPoint's seven fields + the raw depth sample of Frame::GetDepth. */
typedef struct vo_keypoint {
    float x, y;             /* level-0 pixel coordinates (level coords * 1.2^octave)   */
    float size;             /* 31 * 1.2^octave                                         */
    float angle;            /* degrees [0,360)                                         */
    float response;         /* Harris response                                         */
    int32_t octave;         /* pyramid level                                           */
    int32_t class_id;       /* -1                                                      */
    int32_t depth_raw;      /* src/frame.cpp:43-67: raw u16 at (round x, round y) or the
                               first non-zero of its 4 neighbours; 0 = no depth         */
} vo_keypoint;

typedef struct vo_match {
    int32_t map_index;      /* device-map slot of the query map point                  */
    int32_t kp_index;       /* index of the matched keypoint in the frame slot         */
    int32_t distance;       /* Hamming distance 0..256                                 */
    int32_t flags;          /* VO_MATCH_* bits, filled by vo_track_frame               */
} vo_match;

#define VO_MATCH_RANSAC_INLIER 1   /* in cv::solvePnPRansac's inlier list (frontend.cpp:242)   */
#define VO_MATCH_LM_INLIER     2   /* chi2 <= 1 after the second LM round (frontend.cpp:317-329) */

#define VO_MAP_FLAG_OUTLIER 1      /* Mappoint::outlier_ */

typedef struct vo_track_params {
    float match_ratio;          /* match_ratio 2.0 (default.yaml:21)            */
    float match_floor;          /* 30.0 (frontend.cpp:196)                      */
    int32_t n_hyp;              /* RANSAC iterations cap, 100 (frontend.cpp:240) */
    float reproj_px;            /* 4.0                                          */
    float confidence;           /* 0.99                                         */
    uint64_t seed;              /* hypothesis sampler seed                       */
    double huber_delta;         /* sqrt(7.815) (frontend.cpp:282)               */
    double chi2_cut;            /* 1.0 (frontend.cpp:300,:322)                  */
    int32_t it_robust, it_plain;/* 10, 10 (frontend.cpp:291,:310)               */
    int32_t passes;             /* 2 = coarse + fine (frontend.cpp:102-108)     */
    int32_t reserved[5];
} vo_track_params;

typedef struct vo_track_result {
    double T_cw[12];            /* refined pose                                               */
    int32_t n_candidates;       /* map points that passed the frustum/view-angle filter       */
    int32_t n_matches;          /* pairs that passed the distance gate                        */
    int32_t n_ransac_inliers;   /* numInliers_ (frontend.cpp:242)                              */
    int32_t n_lm_inliers;       /* |pnpMatchedMptSet_|                                         */
    int32_t min_distance;       /* min_dis (frontend.cpp:190-195), -1 if no match              */
    int32_t ransac_iters;       /* hypotheses actually consumed by the adaptive stop           */
    int32_t best_hypothesis;    /* index of the winning hypothesis, -1 if none                  */
    int32_t lm_iters;           /* LM outer iterations executed (both rounds)                   */
    int32_t status;             /* VO_OK or VO_E_OVERFLOW raised inside the device pipeline     */
    int32_t reserved[7];
} vo_track_result;

/* Local bundle adjustment problem (flattened covisibility graph, built by the host). */
typedef struct vo_ba_problem {
    int32_t n_poses;            /* free poses first, then fixed ones */
    int32_t n_free;
    int32_t n_points;
    int32_t n_edges;
    const double* poses;        /* n_poses x 12 */
    const double* points;       /* n_points x 3 */
    const int32_t* edge_pose;   /* n_edges */
    const int32_t* edge_point;  /* n_edges */
    const float* edge_uv;       /* n_edges x 2, observed pixel (Point2f) */
    double huber_delta;         /* sqrt(7.815) backend.cpp:83 */
    double chi2_th;             /* chi2_th backend.h:24 */
    int32_t it_robust, it_plain;/* 10, 10 backend.cpp:141,:159 */
} vo_ba_problem;

typedef struct vo_ba_result {
    double* poses;              /* n_free x 12, caller allocated */
    double* points;             /* n_points x 3, caller allocated */
    uint8_t* edge_flags;        /* n_edges: bit0 culled after round 1 (backend.cpp:147-152),
                                   bit1 chi2 > th after round 2 at level 0 (:165-170)        */
    double chi2_initial, chi2_final;
    int32_t lm_iters;
    int32_t reserved[3];
} vo_ba_result;

/* ---- lifetime ---------------------------------------------------------------------- */
int vo_ctx_create(const vo_params* p, int device, vo_ctx** out);
void vo_ctx_destroy(vo_ctx* ctx);
const char* vo_strerror(int status);
const char* vo_backend_name(void);          /* "hip-gfx950" or "cpu-oracle" */
int vo_trace_level(void);                   /* the environment's VO_TRACE (0: unset), read once: the library and the host layer print their [vo_trace] lines to stderr when it is > 0 */
int vo_default_params(vo_params* p);        /* fills default.yaml + TUM fr1 values */
int vo_default_track_params(vo_track_params* tp);

/* ---- frames ------------------------------------------------------------------------ */
/* Copy a BGR8 + depth16 frame from host memory into frame slot `slot` (strides in bytes).  Pageable sources: the call returns when
 * the copies are done.  Page-locked sources (hipHostMalloc / hipHostRegister / torch pin_memory): the copies are enqueued and the call
 * returns at once -- the buffers must stay untouched until the next call that waits for the context (vo_orb_detect_describe, vo_sync). */
int vo_frame_upload(vo_ctx* ctx, int slot, const uint8_t* bgr, int bgr_stride,
                    const uint16_t* depth, int depth_stride);
/* The uploads of the NEXT look-ahead batch ahead of time (page-locked sources only; a no-op otherwise): frames slot0 .. slot0 + n - 1
 * are enqueued on a copy stream of the context's own into one of two device slabs, beside whatever the context is doing with the slots'
 * current frames; evenly spaced host frames (one array) travel as one 2-D copy per image kind.  A later vo_frame_upload of a slot from
 * the same buffers and strides takes the frame from there (no copy, no host wait).  The host buffers must stay untouched from this
 * call until the frames' ORB results have been fetched; a slot that took a preloaded frame must be rebound before the second next
 * preload (its slab is reused then; such a slot is unbound rather than left showing new bytes).  Mirrors what a reader thread ahead of
 * FrontEnd::AddFrame does for run_vo (app/run_vo.cpp:91-109). */
int vo_frames_preload(vo_ctx* ctx, int slot0, int n, const uint8_t* const* bgr, int bgr_stride,
                      const uint16_t* const* depth, int depth_stride);
/* Use frames already resident in device memory (no copy; pointers must stay valid until the
 * slot is rebound).  For the CPU oracle these are ordinary host pointers. */
int vo_frame_bind_device(vo_ctx* ctx, int slot, const void* d_bgr, int bgr_stride,
                         const void* d_depth, int depth_stride);

/* ---- ORB --------------------------------------------------------------------------- */
/* Detect + describe on slots [slot0, slot0+nslots) in one batched launch chain (asynchronous
 * on the context's stream; results stay on the device). */
int vo_orb_detect_describe(vo_ctx* ctx, int slot0, int nslots);
/* Download slot results: up to `cap` keypoints and cap x 32 descriptor bytes (either pointer may be NULL). */
int vo_orb_fetch(vo_ctx* ctx, int slot, vo_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* Debug/parity taps: pyramid level image (gray u8, tightly packed w*h) and its size. */
int vo_orb_level_size(vo_ctx* ctx, int level, int* w, int* h, int* quota);
int vo_orb_fetch_level(vo_ctx* ctx, int slot, int level, uint8_t* gray_out);
/* The 7x7 sigma-2 blurred copy of that level (the image the rBRIEF tests read), same packing. */
int vo_orb_fetch_blur_level(vo_ctx* ctx, int slot, int level, uint8_t* blur_out);

/* ---- map ---------------------------------------------------------------------------- */
/* Insert or overwrite map points at device-map slots idx[i] (0 <= idx < map_capacity). */
int vo_map_upsert(vo_ctx* ctx, const int32_t* idx, const double* xyz, const double* normal,
                  const uint8_t* desc, const uint8_t* flags, int n);
/* The same for points created from keypoints of the frame in `frame_slot` (reference src/frontend.cpp:372-406, where the new
 * point copies its keypoint's descriptor row): the descriptors are copied on the device from that slot's ORB results, so they
 * never travel to the host and back.  kp_index[i] = keypoint of point i; the slot must still hold the frame's ORB results.
 * vo_orb_fetch with desc == NULL then leaves the descriptors of a batch on the device altogether. */
int vo_map_upsert_from_frame(vo_ctx* ctx, int frame_slot, const int32_t* kp_index, const int32_t* idx, const double* xyz,
                             const double* normal, const uint8_t* flags, int n);
/* Define the tracking map: the ordered list of device-map slots matched against. */
int vo_map_set_active(vo_ctx* ctx, const int32_t* idx, int n);

/* ---- per-stage entry points (parity tests, and callers that want the seams) ---------- */
int vo_match_active_map(vo_ctx* ctx, int slot, const double T_cw[12], float ratio, float floor_dist,
                        vo_match* out, int cap, int* n_out, int* n_candidates, int* min_distance);
/* Replace the context's current correspondence set by an explicit one (3-D points as the
 * float32 the reference casts to at frontend.cpp:228, pixels as Point2f). */
int vo_matches_set(vo_ctx* ctx, const float* xyz, const float* uv, int n);
int vo_pnp_ransac(vo_ctx* ctx, int n_hyp, float reproj_px, float confidence, uint64_t seed,
                  double T_cw_inout[12], int32_t* inliers, int cap, int* n_inliers,
                  int32_t* hyp_counts /* optional, n_hyp entries */, int* iters_used, int* best_hyp);
int vo_pose_refine_lm(vo_ctx* ctx, double T_cw_inout[12], double huber_delta, double chi2_cut,
                      int it_robust, int it_plain, uint8_t* inlier_mask /* one per RANSAC inlier */,
                      int cap, int* n_edges, int* lm_iters);

/* Matches of one lane of the LAST vo_track_batch / vo_track_frame call (lane 0 for vo_track_frame), for callers that
 * passed matches = NULL there and only need the records of some frames (the host layer: keyframes only -- the
 * reference reads flannMatchedMptKptMap_ / pnpMatchedMptSet_ only when a keyframe is inserted, frontend.cpp:366-406).
 * Valid until the next tracking call on the context. */
int vo_track_fetch_matches(vo_ctx* ctx, int lane, vo_match* matches, int cap, int* n_out);

/* ---- fused per-frame tracking ------------------------------------------------------- */
int vo_track_frame(vo_ctx* ctx, int slot, const double T_cw_prior[12], const vo_track_params* tp,
                   vo_track_result* res, vo_match* matches, int cap);

/* Track n frames (slots[i]) that share the prior pose and the map in ONE batched launch chain.
 * Frames between two keyframes are independent of each other: the reference's prior is the last
 * keyframe's pose (src/frontend.cpp:96), not the previous frame's.  res[i] / matches[i*cap ..] per
 * frame; seeds[i] replaces tp->seed for frame i.  n <= max_track_batch. */
int vo_track_batch(vo_ctx* ctx, int n, const int* slots, const double T_cw_prior[12], const vo_track_params* tp,
                   const uint64_t* seeds, vo_track_result* res, vo_match* matches, int cap);
/* vo_track_batch in two halves, for a caller with host work to do while the launch chain runs: _begin enqueues the chain and returns
 * (its arrays are copied), _end waits and fills results[n]; match records stay on the device (vo_track_fetch_matches).  Between the two
 * the context must not be given other tracking or map-changing calls.  A member of a stream group gets VO_E_UNSUPPORTED (its chain is
 * shared); _end without a chain in flight returns VO_E_STATE.  Results equal vo_track_batch's. */
int vo_track_batch_begin(vo_ctx* ctx, int n, const int* slots, const double T_cw_prior[12], const vo_track_params* tp,
                         const uint64_t* seeds, int match_cap);
int vo_track_batch_end(vo_ctx* ctx, vo_track_result* results);

/* ---- stream groups --------------------------------------------------------------------- */
/* Several contexts = several independent RGB-D streams on ONE GPU.  Frames of different streams never depend on each
 * other, so their tracking chains can share launches: after vo_group_join, a member's vo_track_batch (called from the
 * member's own host thread, as before) is coalesced with the calls the other members have pending into ONE launch chain
 * whose lanes carry their own map / frame / prior pointers.  Results are identical to un-grouped calls.  The reference
 * has no counterpart (one process tracks one stream, app/run_vo.cpp:89-117); this is how BASELINE config 4's streams
 * share a GPU when there are more streams than GPUs. */
typedef struct vo_group vo_group;
int vo_group_create(int device, int max_lanes, vo_group** out);        /* max_lanes: lanes of one fused chain (<= 128) */
void vo_group_destroy(vo_group* g);                                    /* after every member has left or been destroyed */
int vo_group_join(vo_group* g, vo_ctx* ctx);
int vo_group_leave(vo_group* g, vo_ctx* ctx);
/* A leader may wait up to timeout_us for min_requests pending requests before it launches (default: launch at once). */
int vo_group_set_gather(vo_group* g, int min_requests, int timeout_us);
int vo_group_stats(vo_group* g, int64_t* chains, int64_t* lanes, int64_t* requests);

/* ---- RANSAC hypotheses sharded over ranks ------------------------------ */
/* Several processes (one per GPU) track the SAME stream; each scores only its share of the PnP-RANSAC hypotheses
 * (cv::solvePnPRansac, reference src/frontend.cpp:238-241): hypothesis h belongs to rank h % world.  Every rank generates
 * all hypotheses (cheap, counter-based sampler: identical everywhere), scores its own, then `exchange` sums the per-
 * hypothesis inlier counts element-wise over the ranks IN PLACE (counts of foreign hypotheses are sent as 0), after which
 * every rank runs the same sequential adaptive-stop scan on the full count vector: results are identical to the un-sharded
 * call on every rank.  One exchange per RANSAC pass and launch chain (lanes x n_hyp int32).  The caller supplies the
 * collective (RCCL / gloo all-reduce through torch.distributed in this repo); world <= 1 switches sharding off. */
typedef void (*vo_exchange_fn)(void* user, int32_t* counts, int n);
int vo_set_hypothesis_shard(vo_ctx* ctx, int rank, int world, vo_exchange_fn exchange, void* user);
/* On-stream form of the same exchange (north_star: "an RCCL all-reduce over xGMI ... where a frame batch is in flight"): `allreduce`
 * is called on the caller's thread but must only ENQUEUE an in-place int32 SUM of `n` device-resident counts on `hip_stream` -- for
 * RCCL: ncclAllReduce(buf, buf, n, ncclInt32, ncclSum, (ncclComm_t)comm, (hipStream_t)hip_stream) -- so the launch chain never
 * returns to the host between scoring and the adaptive-stop scan.  The buffer is the context's count table (lanes x max_hypotheses:
 * entries beyond n_hyp of a lane are summed too and never read).  Returns nonzero on failure (the chain fails with VO_E_DEVICE).
 * The CPU restatement has no streams: VO_E_UNSUPPORTED there.  Setting one form clears the other. */
typedef int (*vo_stream_allreduce_fn)(void* comm, int32_t* device_counts, size_t n, void* hip_stream);
int vo_set_hypothesis_shard_stream(vo_ctx* ctx, int rank, int world, vo_stream_allreduce_fn allreduce, void* comm);

/* ---- local BA sharded over ranks (last sentence; BASELINE config 5) -------- */
/* Several processes (one per GPU) hold the SAME local-BA problem (Backend::Optimize, reference src/backend.cpp:19-195) and call vo_local_ba
 * with identical arguments; each linearises only the edges of ITS map points (point k belongs to rank k % world: a point's 3x3 block, its
 * edges and its pairs of the Schur complement are one rank's), and per LM step the ranks exchange
 *   1. the pose blocks H_pp, b_p, the robust chi2 of the linearisation (+ per-rank maxima for lambda_0)      36 n_free + D + 2 + world doubles
 *   2. the reduced system: S (D x D, row-major) and b_s -- the all-reduce SURVEY 8e names (115 KB at D = 120)   D^2 + D doubles
 *   3. the trial state's chi2, the gain ratio's denominator and the per-rank largest step                      2 + world doubles
 * as element-wise SUMs in place; every rank then factors the same S, takes the same LM decision and moves the free poses identically, so the
 * ranks stay in lockstep without a broadcast.  Points and edge flags are one rank's each; a last exchange (3 n_points + n_edges + 2 doubles) hands
 * every rank the whole result: poses, points, edge_flags and the chi2 values of the un-sharded call up to the summation order (tests: 1e-6 on
 * poses, identical flags).  The exchanges are latency-bound (20 LM steps x 3): the mode is opt-in, for problems whose linearisation dominates
 * (config 5: 160 k edges).  Every system takes the launch-per-phase step with S in global memory here (k_ba_chol16g), whatever its size.
 * `exchange` works on a HOST buffer (gloo / MPI); the _stream form must only ENQUEUE an in-place f64 SUM of n device-resident doubles on hip_stream --
 * for RCCL: ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, (ncclComm_t)comm, (hipStream_t)hip_stream).  world <= 1 switches sharding off; setting one
 * form clears the other.  vo_local_ba_resident (the device graph cut) is not sharded: VO_E_UNSUPPORTED while a BA shard is set. */
typedef void (*vo_exchange_f64_fn)(void* user, double* data, int n);
int vo_set_ba_shard(vo_ctx* ctx, int rank, int world, vo_exchange_f64_fn exchange, void* user);
typedef int (*vo_stream_allreduce_f64_fn)(void* comm, double* device_data, size_t n, void* hip_stream);
int vo_set_ba_shard_stream(vo_ctx* ctx, int rank, int world, vo_stream_allreduce_f64_fn allreduce, void* comm);

/* ---- batched triangulation -------------------------------------------------------------- */
/* Linear N-view triangulation (reference include/myslam/util.h:16-34) of MANY map points in one launch, as applied by
 * FrontEnd::TriangulateMappointsInTrackingMap (src/frontend.cpp:465-506) to the LM inliers of a keyframe: point i owns the
 * views view_start[i] .. view_start[i+1]-1; per view the keyframe pose T_cw (12 doubles) and the observation on the
 * normalised image plane (x, y; z = 1).  ok[i] = 1 iff sigma4 / sigma3 < 1e-2 (and the point has >= 2 views); xyz is written
 * for every point with >= 2 views. */
int vo_triangulate_batch(vo_ctx* ctx, int n_points, const int32_t* view_start, const double* T_cw, const double* xy,
                         double* xyz_out, uint8_t* ok_out);

/* ---- local bundle adjustment --------------------------------------------------------- */
int vo_local_ba(vo_ctx* ctx, const vo_ba_problem* in, vo_ba_result* out);

/* ---- device-resident observation table and graph cut ------------------ */
/* The keyframe bookkeeping the local BA's graph is cut from, kept on the device: one record per observation (keyframe k sees
 * map point m at pixel uv: Frame::AddObservedMappoint, reference src/frame.cpp:93-120) and the current pose of every keyframe.
 * The caller numbers its keyframes 0, 1, 2, ... in insertion order; observation ids are the append order (0, 1, 2, ...).
 * vo_local_ba_resident cuts the covisibility graph of reference src/backend.cpp:36-135 on the device -- points = the non-outlier
 * map points some free keyframe observes (ascending map slot), edges = every live observation of those points (per point in
 * ascending keyframe order), fixed poses = their observers outside the free set (ascending keyframe number, after the free
 * ones) -- solves it (same LM as vo_local_ba) and returns what the caller needs for its write-back (src/backend.cpp:144-194).
 * `tables` is the context that owns the map and the observation table (the tracker's); `ctx` provides scratch and the stream,
 * so a back-end thread can solve beside the tracker.  The device tables are NOT modified: the caller merges (vo_map_upsert,
 * vo_kf_set_pose, vo_obs_kill) at the moment it chooses.
 * Limits and threading: the observation table grows on demand up to 256 Mi entries (it starts at 4 Mi: 68 MB), the keyframe table holds 64 Ki keyframes, the map may use slots below 16 Mi (VO_E_OVERFLOW / VO_E_INVALID beyond: the caller goes
 * back to vo_local_ba with a graph of its own); a cut takes at most VO_BA_RESIDENT_MAX_FREE free keyframes (the host back-end's own cap), each listed once.  The tables belong to the
 * tracker's thread; a back-end thread may run _cut on them from its own context as long as the tracker appends nothing, changes no
 * flag or position of the map and kills no observation between the call and its return (host/src/backend.cpp: WaitGraphCut). */
int vo_kf_set_pose(vo_ctx* ctx, const int32_t* kf, const double* T_cw, int n);
int vo_obs_append(vo_ctx* ctx, const int32_t* kf, const int32_t* map_idx, const float* uv, int n, int64_t* first_id);
int vo_obs_kill(vo_ctx* ctx, const int64_t* obs_ids, int n);
typedef struct vo_ba_resident_result {
    double* poses;              /* n_free x 12, optimised free poses (caller allocated)                                   */
    int32_t* point_slots;       /* cap_points: device-map slots of the graph's points, graph order                        */
    double* points;             /* cap_points x 3: optimised positions                                                     */
    int64_t* culled_obs;        /* cap_culled: observations culled by the two chi2 tests (src/backend.cpp:144-172)         */
    int32_t cap_points, cap_culled;
    int32_t n_points, n_fixed, n_edges, n_culled;
    double chi2_initial, chi2_final;
    int32_t lm_iters;
    int32_t n_pairs;            /* pairs (e1, e2) of the reduced system's pair plan: what one Schur launch contracts (byte / flop accounting)  */
} vo_ba_resident_result;
#define VO_BA_RESIDENT_MAX_FREE 160
int vo_local_ba_resident(vo_ctx* ctx, vo_ctx* tables, const int32_t* free_kf, int n_free, double huber_delta, double chi2_th,
                         int it_robust, int it_plain, vo_ba_resident_result* out);
/* The same in two steps, for a back-end thread that solves beside the tracker: _cut returns as soon as the problem's arrays
 * are complete (from then on `tables` may change: new keyframes, merges), _solve runs the optimisation on what _cut left in `ctx`. */
int vo_local_ba_resident_cut(vo_ctx* ctx, vo_ctx* tables, const int32_t* free_kf, int n_free, double huber_delta, double chi2_th,
                             int32_t* n_points, int32_t* n_fixed, int32_t* n_edges);
int vo_local_ba_resident_solve(vo_ctx* ctx, int it_robust, int it_plain, vo_ba_resident_result* out);
/* Merge on the device, for a caller whose next graph cut should not wait for the result's trip to the host and back
 * (host/src/backend.cpp): vo_local_ba_resident_solve called with poses = point_slots = points = NULL leaves the optimised state in
 * `ctx` and returns only the counts and the culled observations (the caller's covisibility ledger needs them before it picks the next
 * free keyframes); _merge then writes that state into `tables` -- the positions of the graph's points whose map flag is not
 * VO_MAP_FLAG_OUTLIER (vo_map_upsert), the free keyframes' poses (vo_kf_set_pose), the culled observations (vo_obs_kill) -- on
 * the tables' stream, and the next _cut of `ctx` is ordered behind it; _fetch brings poses / point_slots / points to the host
 * afterwards (any time before the next _merge of `ctx`).  Replaces the write-back of reference src/backend.cpp:144-194 on the device side. */
int vo_local_ba_resident_merge(vo_ctx* ctx, vo_ctx* tables);
int vo_local_ba_resident_fetch(vo_ctx* ctx, vo_ba_resident_result* out);
/* Parity tap: the graph vo_local_ba_resident would cut, in vo_ba_problem layout (edge_obs: observation id per edge). */
int vo_ba_resident_graph(vo_ctx* ctx, vo_ctx* tables, const int32_t* free_kf, int n_free, int32_t* n_poses, int32_t* pose_kf, int cap_poses,
                         int32_t* n_points, int32_t* point_slots, int cap_points, int32_t* n_edges, int32_t* edge_pose, int32_t* edge_point,
                         float* edge_uv, int64_t* edge_obs, int cap_edges);

/* Diagnostic: how much of the tables the last vo_local_ba_resident_cut / vo_ba_resident_graph of `ctx` visited -- observations from
 * the first one of the oldest point a free keyframe observes, map slots from that point's slot.  Both depend on the cut's window, not on
 * how long the run has been (the reference's Backend walks only the covisible keyframes' observation maps: src/backend.cpp:36-120). */
int vo_ba_resident_window(vo_ctx* ctx, int64_t* observations_visited, int64_t* map_slots_visited);
/* The resident cut carves its scratch slab for UPPER BOUNDS of the graph's sizes (every observation of the window an edge, ...) so that the host
 * need not wait for the sizes in the middle of the launch chain.  When those bounds ask for more than `bytes` (default 1 GiB: a revisit whose window
 * spans most of the tables with ~100 free keyframes) the cut waits for the sizes and carves exactly.  0 < bytes; the results do not depend on it. */
int vo_ba_resident_set_slab_budget(vo_ctx* ctx, int64_t bytes);

/* ---- keyframe bookkeeping on the device (the rest of the row) ------------- */
/* What FrontEnd::TrackingHandler does to host objects when a frame becomes a keyframe (reference src/frontend.cpp:119-131), done on the
 * tables above without a trip through the host: the tracking chain's match records, the frame's keypoints, depth samples and descriptors
 * and the map are all resident already.  One call =
 *   AddCurrentKeyframeObservations   src/frontend.cpp:366-370   one observation per LM-inlier match (match order); Frame::AddObservedMappoint
 *                                    src/frame.cpp:93-120: the map point's mean viewing direction (src/mappoint.cpp:30-38) and the
 *                                    covisibility weight of every keyframe that already sees the point (+1 per shared point);
 *   CreateNewMappoints               src/frontend.cpp:372-406   every keypoint that is no LM inlier and has depth becomes a map point at
 *                                    Camera::Pixel2World (src/camera.cpp:41-86) with its descriptor row; slots first_new_slot, +1, ... in
 *                                    keypoint order; one observation each, behind the matched ones;
 *   TriangulateMappointsInTrackingMap src/frontend.cpp:465-506  the LM-inlier points that are neither outliers nor triangulated nor optimised
 *                                    are triangulated from all their live observations (include/myslam/util.h:16-34) in match order; the FIRST
 *                                    that succeeds with z > 0 is moved and flagged (the reference's loop breaks there, :501);
 *   the keyframe's pose and observations in the tables (what vo_kf_set_pose / vo_obs_append do from host arrays).
 * lane: lane of the context's last tracking call that holds this frame's match records (-1: no matches -- the first keyframe).
 * covis_kf / covis_weight receive the keyframes (ascending number) that share >= 1 live observation's point with the new one and the
 * counts: allCovisibleKeyframeIdToWeight_ of the new keyframe (include/myslam/frame.h:94); the caller keeps the ledger.  More than
 * cap_covis (or 4096) partners: the list is cut, out->n_covisible_total says how many there are and vo_kf_covisibility reads them all
 * (nothing is lost on the device).  VO_E_OVERFLOW means what it says elsewhere: the observation / keyframe tables are full, nothing was written.
 * The map flags carry Mappoint::triangulated_ / optimized_ beside outlier_ (VO_MAP_FLAG_*).
 * The map has no fixed size on this path (the reference's is a host container): when first_new_slot + n_features exceeds the context's map capacity the map
 * arrays and the tracking chain's per-lane buffers are reallocated at twice the size, contents kept, before anything is written (no tracking chain of this
 * context may be in flight, which holds at this point of AddFrame anyway; VO_E_NOMEM / VO_E_OVERFLOW if that fails or passes 2^28 points). */
#define VO_MAP_FLAG_TRIANGULATED 2   /* Mappoint::triangulated_ (src/frontend.cpp:498) */
#define VO_MAP_FLAG_OPTIMIZED    4   /* Mappoint::optimized_    (src/backend.cpp:190)  */
typedef struct vo_kf_commit_result {
    int32_t n_matched;          /* observations added for LM-inlier matches                                        */
    int32_t n_new;              /* map points created: slots first_new_slot .. first_new_slot + n_new - 1          */
    int64_t first_obs;          /* observation id of the first record appended (matched ones first, then the new)  */
    int32_t n_covisible;        /* entries written to covis_kf / covis_weight                                      */
    int32_t n_tri_candidates;   /* points the triangulation loop looked at                                         */
    int32_t triangulated_slot;  /* map slot moved by the first successful triangulation, -1: none                  */
    int32_t n_covisible_total;  /* partners the new keyframe has; > n_covisible: the list was cut -- vo_kf_covisibility has all */
} vo_kf_commit_result;
int vo_keyframe_commit(vo_ctx* ctx, int lane, int frame_slot, int32_t kf, const double T_cw[12], int32_t first_new_slot,
                       int32_t* covis_kf, int32_t* covis_weight, int cap_covis, vo_kf_commit_result* out);
/* Covisibility weights of keyframe `kf` recounted from the tables (live observations only): the parity tap of the ledger. */
int vo_kf_covisibility(vo_ctx* ctx, int32_t kf, int32_t* covis_kf, int32_t* covis_weight, int cap, int32_t* n);
/* MapManager::GetMappointsAroundKeyframe (reference src/mapmanager.cpp:14-38) + the tracking-map rule of src/frontend.cpp:159-166 on the
 * device: the active list becomes the non-outlier map points the listed keyframes observe (live observations), ordered by the first
 * listed keyframe that sees them and by observation order inside it; fewer than min_points of them: every slot below n_map_points
 * (the reference falls back to GetAllMappoints).  Replaces vo_map_set_active for a caller that keeps no host map. */
int vo_map_set_active_covisible(vo_ctx* ctx, const int32_t* kf, int n, int min_points, int32_t n_map_points, int32_t* n_active);
/* vo_local_ba_resident_merge for a caller that keeps no host map objects: besides positions, free poses and culled observations, every
 * point of the graph is flagged VO_MAP_FLAG_OPTIMIZED (src/backend.cpp:190), a point that lost its last live observation becomes an
 * outlier (Mappoint::RemoveObservedByKeyframe, src/mappoint.cpp:40-45), and the covisibility ledger's decrements come back as keyframe
 * pairs: culling the observation (K, P) costs K and every other keyframe that still sees P one shared point (Frame::RemoveObservedMappoint,
 * src/frame.cpp:122-152).  poses receives the n_free optimised poses (what _fetch would bring).  Waits for the tables' stream.
 * The number of pairs has no bound the caller could know (culled observations x the other observers of their points: the reference's chi2
 * threshold of 1 culls a large share of a real sequence's edges).  More than cap_pairs of them: VO_E_OVERFLOW, *n_pairs = the number the call
 * needs, and the merge is INCOMPLETE (the culled observations are still marked in the tables) until the call has been repeated with arrays of
 * that size: same pairs and same tables then as from one call with large arrays.  Nothing else may use `tables` between the two calls. */
int vo_local_ba_resident_merge_ledger(vo_ctx* ctx, vo_ctx* tables, int32_t* pair_a, int32_t* pair_b, int cap_pairs, int32_t* n_pairs,
                                      double* poses, int cap_poses);
/* Parity / inspection tap: the tables as they stand.  Any pointer may be NULL; n_obs / n_map receive the totals. */
int vo_tables_fetch(vo_ctx* ctx, int64_t obs0, int64_t obs_cap, int32_t* obs_kf, int32_t* obs_mp, float* obs_uv, uint8_t* obs_alive, int64_t* n_obs,
                    int32_t map0, int32_t map_cap, double* map_xyz, double* map_normal, uint8_t* map_desc, uint8_t* map_flags,
                    int32_t* active, int active_cap, int32_t* n_active);

/* Test tap: the process-wide call number of the one-launch prefix sums (the graph cut's two and the local-map query's one per keyframe; a scan's
 * workgroups publish (call number << 32 | tile total) words).  set_to >= 0 sets it first -- a test then walks the number through the range of values
 * the scratch arrays hold (tests/test_device_keyframes.py: the words once lived at a place that follows the window).  Returns the number (the CPU
 * restatement has no such scan: 0). */
long long vo_scan_call_number(long long set_to);

/* ---- plumbing ------------------------------------------------------------------------- */
int vo_sync(vo_ctx* ctx);
/* Per-kernel accumulated device time measured with HIP events on the context's stream
 * (enabled by vo_profile_enable).  names/ms/calls are caller arrays of `cap` entries. */
int vo_profile_enable(vo_ctx* ctx, int on);
int vo_profile_read(vo_ctx* ctx, char (*names)[48], double* ms, int64_t* calls, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
