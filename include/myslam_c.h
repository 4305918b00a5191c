/*
 * myslam_c.h -- C wrapper of the libmyslam-compatible host layer (rgbd_visualodometry_amd/host),
 * so that Python (bench.py, tests) can drive the same FrontEnd::AddFrame loop that the
 * reference's app/run_vo.cpp:89-117 drives.  One myslam_system = Camera + FrontEnd + Backend +
 * its own MapManager, i.e. one independent RGB-D stream.
 */
#ifndef MYSLAM_C_H
#define MYSLAM_C_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct myslam_system myslam_system;

typedef struct myslam_options {
    int32_t width, height;
    float fx, fy, cx, cy, depth_scale;      /* camera.* keys of config/default.yaml */
    int32_t number_of_features;             /* 500  */
    float scale_factor;                     /* 1.2  */
    int32_t level_pyramid;                  /* 8    */
    float match_ratio;                      /* 2.0  */
    int32_t max_num_lost;                   /* 10   */
    int32_t min_inliers;                    /* 10   */
    double keyframe_rotation, keyframe_translation;   /* 0.05, 0.05 */
    int32_t enable_local_optimization;      /* 1    */
    float chi2_th;                          /* 1    */
    int32_t ransac_iterations;              /* 100 (frontend.cpp:240) */
    int32_t backend_lag_frames;             /* 0: BA solved + merged inside AddFrame; L>0: solved on a worker thread,
                                               merged deterministically L frames later (or at the next keyframe) */
    int32_t max_frames_in_flight;           /* look-ahead ORB batch (1 = none) */
    int32_t track_batch;                    /* frames tracked speculatively in one launch chain (1 = none, <= 16) */
    int32_t map_capacity;                   /* device map slots to start with (4 Mi); with device_keyframes the map doubles when a keyframe's points may not fit */
    int32_t device;
    int32_t verbose;
    int32_t triangulate_all;                /* 0: the reference's loop (stops after the first success, frontend.cpp:501); 1: every eligible point, batched */
    int32_t ba_device_graph;                /* 1: the local BA's graph is cut on the device from the resident observation table */
    int32_t reobserve_new_mappoints;        /* 1: run the reference's disabled re-observation pass (frontend.cpp:408-463) at every keyframe */
    int32_t map_descriptors_on_device;      /* 1: new map points take their descriptor from the frame's ORB results on the device (vo_map_upsert_from_frame);
                                               descriptors are never fetched to the host */
    int32_t device_keyframes;               /* 1 (needs ba_device_graph and map_descriptors_on_device): the keyframe bookkeeping -- observations, covisibility weights,
                                               new map points, the triangulation loop, the local-map query, the BA write-back -- runs on the device tables
                                               (vo_keyframe_commit, vo_map_set_active_covisible, vo_local_ba_resident_merge_ledger); the host keeps keyframe poses and
                                               covisibility ledgers and builds Mappoint objects only on request (myslam_materialize).  SURVEY 8f-2 */
} myslam_options;

typedef struct myslam_stats {
    int32_t frames, keyframes, lost, state;         /* state: 0 INITIALIZING, 1 TRACKING, 2 LOST */
    int32_t last_keypoints, last_candidates, last_matches, last_ransac_inliers, last_lm_inliers;
    int32_t map_points;
    int32_t ba_runs, ba_poses, ba_fixed, ba_points, ba_edges, ba_outliers;
    double ba_ms;
    double ms_extract, ms_track, ms_keyframe, ms_backend;   /* accumulated host wall time per stage */
    /* sums over the frames that went through the tracking chain (the averages size the algorithmic bytes of a launch) */
    int64_t tracked_frames, sum_active, sum_candidates, sum_matches, sum_ransac_inliers, sum_lm_inliers, sum_lm_iters;
    int64_t track_launches;                                 /* vo_track_batch calls (launch chains) */
    int32_t ba_failed, ba_capped;                           /* local BA runs skipped after a failed solve / solved with a capped free set */
    int64_t triangulated, reobserved_matches;               /* points refined by triangulation; gated matches counted by the re-observation pass */
    double ba_sum_d3, ba_sum_d2; int64_t ba_sum_edges;      /* over the local BA runs: (6 n_free)^3, (6 n_free)^2, edges (flop / byte accounting) */
    int64_t ba_sum_points, ba_sum_pairs;                    /* ... points, pairs of the Schur pair plan (device graph cut only; 0 otherwise) */
} myslam_stats;

int myslam_default_options(myslam_options* o);
/* yaml_path may be NULL; keys present in the file override `o` defaults, explicit `o` wins when yaml is NULL */
int myslam_system_create(const myslam_options* o, const char* yaml_path, myslam_system** out);
void myslam_system_destroy(myslam_system* s);
/* Register the next n frames (host or device memory) and run batched ORB on them (look-ahead).  The image buffers are NOT copied
 * on the host: they must stay valid and unchanged until the frame has been consumed (myslam_add_prefetched returned for it) -- from
 * page-locked host memory the upload is asynchronous.  (The C++ surface, Frame::CreateFrame, deep-copies like the reference.)
 * Exception: with reobserve_new_mappoints = 1 keyframes are detected again at later keyframes; host frames are then deep-copied
 * by this call, and frames in DEVICE memory must stay valid for as long as the frame can be a covisible keyframe (the whole run). */
int myslam_prefetch(myslam_system* s, int n, const double* stamps, const void* const* bgr, const void* const* depth,
                    int bgr_stride, int depth_stride, int on_device);
/* Start the uploads of the frames of the NEXT myslam_prefetch call (the same buffers, in the same order) now, on a copy stream, beside
 * the tracking of the frames already queued: what a reader thread ahead of FrontEnd::AddFrame does in run_vo.  Page-locked host memory
 * only (a no-op otherwise); the buffers must stay untouched until those frames have been consumed. */
int myslam_preload(myslam_system* s, int n, const void* const* bgr, const void* const* depth, int bgr_stride, int depth_stride);
/* AddFrame on the next prefetched frame, or on an explicit host/device frame if none is queued.
 * tracked = return value of FrontEnd::AddFrame; T_wc = GetPose().inverse() as written by run_vo.cpp:116. */
int myslam_add_frame(myslam_system* s, double stamp, const void* bgr, const void* depth, int bgr_stride, int depth_stride,
                     int on_device, int* tracked, double T_wc[12]);
int myslam_add_prefetched(myslam_system* s, int* tracked, double T_wc[12]);
int myslam_get_stats(myslam_system* s, myslam_stats* st);
/* Stream group: systems (independent streams) on the same GPU whose per-frame tracking shares launch chains (vo_group, include/vo_hip.h).
 * Each member is still driven from its own host thread; trajectories are those of un-grouped systems. */
typedef struct myslam_group myslam_group;
int myslam_group_create(int device, int max_lanes, myslam_group** out);
void myslam_group_destroy(myslam_group* g);                  /* after its members have been destroyed */
int myslam_group_join(myslam_group* g, myslam_system* s);
int myslam_group_stats(myslam_group* g, int64_t* chains, int64_t* lanes, int64_t* requests);
/* Wait for a pending (overlapped) local BA and merge it now: end of a sequence, or of a timed region (Backend::Flush). */
int myslam_flush(myslam_system* s);
/* The vo_ctx (include/vo_hip.h) the system's FrontEnd owns, for profiling taps (vo_profile_*). */
void* myslam_get_context(myslam_system* s);
const char* myslam_last_error(void);

/* ---- taps for parity tests: the host-layer logic checked against an independent model (tests/ref_model.py) ----------
 * None of these is on the product path.  A "scenario" is a map built by hand on a myslam_system that tracks no frames:
 * keyframes with given poses, map points with given positions, observations added and removed one by one through the
 * same Frame / Mappoint / MapManager / Backend methods that FrontEnd::AddFrame uses. */
/* Triangulation (reference include/myslam/util.h:16-34): n views, T_cw n x 12, pts n x 3 on the normalised plane. */
int myslam_triangulate(int n, const double* T_cw, const double* pts, double out_xyz[3], int* ok);
/* Sophus conventions the keyframe policy relies on (reference src/frontend.cpp:343-358): tangent = [translation, rotation]. */
int myslam_se3_log(const double T[12], double out6[6]);
int myslam_se3_exp(const double d6[6], double T_out[12]);
/* FrontEnd::IsGoodEstimation / IsKeyframe on explicit poses (reference src/frontend.cpp:334-364): bit0 good, bit1 keyframe. */
int myslam_keyframe_policy(myslam_system* s, const double T_ref_cw[12], const double T_cur_cw[12], int num_inliers, int* flags);
int myslam_scn_add_keyframe(myslam_system* s, const double T_cw[12], int64_t* id_out);
int myslam_scn_add_mappoint(myslam_system* s, const double xyz[3], int64_t* id_out);
int myslam_scn_observe(myslam_system* s, int64_t keyframe_id, int64_t mappoint_id, float u, float v);       /* Frame::AddObservedMappoint   src/frame.cpp:93-120  */
int myslam_scn_unobserve(myslam_system* s, int64_t keyframe_id, int64_t mappoint_id);                       /* Frame::RemoveObservedMappoint src/frame.cpp:122-152 */
/* allCovisibleKeyframeIdToWeight_ / activeCovisibleKeyframes_ of a keyframe (reference include/myslam/frame.h:94-95), id order */
int myslam_scn_covisibility(myslam_system* s, int64_t keyframe_id, int64_t* ids, int32_t* weights, uint8_t* active, int cap, int* n);
/* MapManager::GetMappointsAroundKeyframe (reference src/mapmanager.cpp:14-38), in the host layer's matching order */
int myslam_scn_local_map(myslam_system* s, int64_t keyframe_id, int64_t* mappoint_ids, int cap, int* n);
/* The graph Backend::Optimize builds for a keyframe (reference src/backend.cpp:36-135): pose ids (free first), points, edges */
int myslam_scn_ba_graph(myslam_system* s, int64_t keyframe_id, int64_t* pose_ids, int cap_poses, int* n_poses, int* n_free,
                        int64_t* point_ids, int cap_points, int* n_points, int32_t* edge_pose, int32_t* edge_point, float* edge_uv, int cap_edges, int* n_edges);
/* map point state: outlier flag, number of observing keyframes, position, mean viewing direction */
int myslam_scn_mappoint(myslam_system* s, int64_t mappoint_id, int* outlier, int* n_obs, double xyz[3], double normal[3]);
/* Run the local BA of a keyframe synchronously and merge it (Backend::Optimize incl. write-back, src/backend.cpp:19-195). */
int myslam_scn_run_ba(myslam_system* s, int64_t keyframe_id);
int myslam_scn_keyframe_pose(myslam_system* s, int64_t keyframe_id, double T_cw[12]);
/* device_keyframes: rebuild the host map objects (Mappoint, observation lists, Frame observation sets) from the device tables as of now, so
 * that the taps above and MapManager's containers show the run's map; keyframe ids in insertion order -> ids[cap], *n = their number.
 * 1 in *on_device iff the system keeps its keyframes on the device. */
int myslam_materialize(myslam_system* s, int64_t* keyframe_ids, int cap, int* n, int* on_device);
/* map point ids in device-slot order (after myslam_materialize for a device_keyframes system) */
int myslam_mappoint_ids(myslam_system* s, int64_t* ids, int cap, int* n);
const char* myslam_backend_name(void);
#ifdef __cplusplus
}
#endif
#endif
