// vo_internal.h -- internal declarations of the HIP implementation of include/vo_hip.h.
// Target: gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, 160 KiB LDS per CU, 256 CUs.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "vo_hip.h"

#define VO_MAX_LEVELS 16
#define VO_WAVE 64

// Everything a kernel needs to know about the pyramid / detector geometry; passed by value.
#ifndef VO_FAST_TH
#define VO_FAST_TH 32                // rows per k_fast_nms tile (64 columns)
#endif
struct DevPlan {
    int W, H, L, nfeat, fast_thr, edge;
    int lw[VO_MAX_LEVELS], lh[VO_MAX_LEVELS], pitch[VO_MAX_LEVELS];
    int quota[VO_MAX_LEVELS], qprefix[VO_MAX_LEVELS + 1];       // selected-keypoint quota per level and its prefix sum
    int ccap[VO_MAX_LEVELS], cprefix[VO_MAX_LEVELS + 1];        // FAST candidate capacity per level and prefix
    unsigned loff[VO_MAX_LEVELS];                               // byte offset of each level inside a slot's pyramid slab
    float scale[VO_MAX_LEVELS];
    int tiles_x[VO_MAX_LEVELS], tile_prefix[VO_MAX_LEVELS + 1]; // FAST tiling (64 x VO_FAST_TH tiles), blockIdx.x -> (level, tile)
    int xcd_map;                                                // 1: XCD-aware tile order in the tiled ORB kernels (always on; kept in the plan for the kernels that read it)
    int btiles_x[VO_MAX_LEVELS], btile_prefix[VO_MAX_LEVELS + 1]; // blur tiling (128x16 tiles over the whole level)
    int tabx[VO_MAX_LEVELS], taby[VO_MAX_LEVELS];               // offsets into the resize tables
    int umax[16];
    unsigned long long umax_pk;     // umax[] as sixteen 4-bit fields (the radius-15 disc is symmetric: |u| <= umax[|v|]  <=>  |v| <= umax[|u|])
    int gk[7];
    int sel_cap;                                                // power-of-two sort capacity of the select kernel
    unsigned pyr_stride;                                        // bytes per slot in the pyramid slab
    float fx, fy, cx, cy;
};

struct SlotDesc {               // where a slot's input frame lives (device memory)
    const uint8_t* bgr; const uint8_t* depth; int bgr_stride, depth_stride;
};

struct ProfRec { const char* name; hipEvent_t a, b; };

// Device-resident result header of the tracking chain (one per context).
struct TrackDev {
    int n_cand, n_match, min_dist, pad0;
    int n_inl, best_hyp, iters_used, best_cnt;
    int lm_iters, n_lm_inl, status, pad1;
    double T[12];               // current pose estimate (in/out through the chain)
    double T_ransac[12];
    long long dbg[8];           // diagnostic stamps (only written by -DVO_LM_STAMPS builds)
};

#define VO_MAX_LANES 16              // lanes one context contributes to a launch chain (vo_track_batch)
#define VO_GROUP_MAX_LANES 128       // lanes of one fused chain of a stream group
struct CamD { double fx, fy, cx, cy; int W, H; };
// Per-lane working set of the tracking chain.  A "lane" is one frame tracked by a launch chain (blockIdx.z).  Frames
// between two keyframes of a stream share the prior pose and the map (reference src/frontend.cpp:96 -- the prior is the
// last KEYFRAME's pose), so they are independent lanes; frames of different streams (contexts of a stream group) are
// independent anyway.  Every lane carries its own pointers: the chain's kernels never assume that two lanes share a map.
struct LaneDesc {
    TrackDev* tr; uint32_t* best; int32_t* cand; vo_match* matches; float* cxyz; float* cuv;
    double* hyp_pose; int* hyp_cnt; int32_t* inliers; uint8_t* mask;
    double* lm_x;                                   // hand-off area of a pose LM that runs in several workgroups (vo_track.hip, LM_X_DOUBLES)
    const uint32_t* fdesc; const int* nkp; const vo_keypoint* kps;                  // the lane's frame slot (ORB results)
    const double* map_pos; const double* map_nrm; const uint8_t* map_flags; const uint32_t* map_desc; const int32_t* active;
    int n_active, cap, max_hyp, gate_lds;
    unsigned long long seed;
    CamD cam;
};
// Buffers of one launch chain: lane descriptors + result headers, device copies and pinned mirrors.  A context owns
// one (its own lanes); a stream group owns one for the fused chain of its members.
struct LaunchSet {
    int cap = 0;
    LaneDesc* d_lanes = nullptr; LaneDesc* h_lanes = nullptr;      // h_lanes pinned
    TrackDev* d_track = nullptr; TrackDev* h_track = nullptr;      // h_track pinned
};

struct vo_ctx {
    vo_params p;
    DevPlan plan;
    int device;
    hipStream_t stream;
    // frames
    std::vector<uint8_t*> own_bgr, own_depth;       // per slot (allocated lazily on upload)
    std::vector<size_t> own_bgr_bytes, own_depth_bytes;
    // vo_frames_preload: two slabs of max_frames frames each, filled in turn on a copy stream of the context's own while the slots' current
    // frames are still in use; a slot that takes a preloaded frame points into the slab until it is rebound
    struct PreSlab { uint8_t* bgr = nullptr; uint8_t* depth = nullptr; size_t nb = 0, nd = 0; hipEvent_t ev = nullptr; bool waited = true; };
    PreSlab pre[2]; int pre_next = 0;
    struct PreSlot { const void* src_bgr = nullptr; const void* src_depth = nullptr; int bs = 0, ds = 0, gen = -1; bool valid = false; };
    std::vector<PreSlot> pre_slot; std::vector<signed char> slot_gen;      // slot_gen[i]: the slab slot i's frame lives in (-1: not a preloaded one)
    hipStream_t copy_stream = nullptr; hipEvent_t orb_ev = nullptr; bool orb_ev_set = false;
    SlotDesc* d_slots; std::vector<SlotDesc> h_slots;
    SlotDesc* h_slots_pinned; hipEvent_t slots_ev; bool slots_dirty, slots_pending;
    std::vector<char> slot_bound, slot_orb;
    // pyramid + ORB work buffers
    uint8_t* d_pyr; uint8_t* d_blur;                // gray pyramid and its 7x7 sigma-2 blurred copy, same layout
    int* d_tab; short* d_tabs;                      // resize tables: int offsets, short weights
    void* d_pyr_rng = nullptr; int pyr_gx = 0, pyr_gy = 0;      // k_pyramid: tiles per level and what each tile needs of each level (0 x 0: level-by-level k_resize)
    uint32_t* d_cand; int* d_cand_cnt;              // [slot][cprefix[L]] packed candidates, [slot][L] counts
    uint32_t* d_sel; long long* d_sel_key; int* d_sel_cnt;   // [slot][nfeat] selected (x|y<<12), keys, [slot][L] counts
    vo_keypoint* d_kps; uint8_t* d_desc; int* d_nkp;         // [slot][nfeat], [slot][nfeat*32], [slot]
    int* d_status;                                  // sticky overflow flag
    // map
    double* d_map_pos; double* d_map_nrm; uint32_t* d_map_desc; uint8_t* d_map_flags;
    int32_t* d_active; int n_active; int active_cap;
    // tracking chain
    int last_track_lanes = 0;                       // lanes of the last vo_track_batch (vo_track_fetch_matches)
    int match_hint = 0;                             // largest recent match count (sizes the result copy of vo_track_batch)
    uint32_t* d_best;                               // per active query: (dist << 22) | kp, 0xFFFFFFFF = not a candidate
    int32_t* d_mcand;                               // visible candidates (indices into the active list), unordered
    vo_match* d_matches; float* d_corr_xyz; float* d_corr_uv; int corr_cap;
    double* d_hyp_pose; int* d_hyp_cnt;             // [max_hyp][12], [max_hyp]
    int32_t* d_inliers; uint8_t* d_lm_mask; double* d_lm_x = nullptr;
    LaunchSet ls;                                   // this context's own lanes (d_track / h_track alias ls.d_track / ls.h_track)
    TrackDev* d_track; TrackDev* h_track;           // [lanes]; h_track pinned
    size_t lane_stride; int lanes;                  // element stride of the per-lane chain buffers
    struct vo_group* group = nullptr; hipEvent_t group_ev = nullptr;     // stream group membership (vo_group_join)
    int shard_rank = 0, shard_world = 1; vo_exchange_fn shard_fn = nullptr; void* shard_user = nullptr;     // RANSAC hypotheses sharded over ranks
    vo_stream_allreduce_fn shard_stream_fn = nullptr;     // on-stream form of the exchange (shard_user = communicator)
    int ba_shard_rank = 0, ba_shard_world = 1; vo_exchange_f64_fn ba_shard_fn = nullptr; vo_stream_allreduce_f64_fn ba_shard_stream_fn = nullptr; void* ba_shard_user = nullptr;      // local BA sharded over ranks by point (vo_set_ba_shard*)
    void* d_ba_shard = nullptr; size_t ba_shard_bytes = 0; void* h_ba_shard = nullptr;      // its exchange buffers + descriptor / control block (device), status record (pinned)
    vo_match* h_matches;                            // pinned staging
    int h_matches_cap;
    int h_matches_lanes = 0, h_matches_first = 0;   // lanes whose first `h_matches_first` records the last chain left in h_matches (group mode)
    void* h_stage; size_t h_stage_bytes;            // pinned general staging
    uint8_t* h_orb_cache; bool orb_cache_valid; bool orb_cache_desc = false; int orb_batch0, orb_batchn;   // pinned copy of the last ORB batch's results
    bool corr_external;
    // BA scratch; the device's BA engine (shared by the contexts of that device, vo_ba.hip)
    void* d_ba; size_t d_ba_bytes; struct BaEngine* ba_engine = nullptr; struct BaEngine* ba_engine_sel = nullptr;
    void* h_ba_up = nullptr; size_t h_ba_up_bytes = 0;      // pinned mirror of a BA problem's upload region
    std::vector<int32_t> ba_pt_start, ba_ps_start, ba_cursor;      // host scratch of vo_ba_run (kept between problems)
    // device-resident keyframe bookkeeping: observation table and keyframe poses, fixed capacity (a back-end
    // thread may be reading them while the tracker appends: no reallocation, appends only write beyond what a reader was given)
    int32_t* d_obs_kf = nullptr; int32_t* d_obs_mp = nullptr; float* d_obs_uv = nullptr; uint8_t* d_obs_alive = nullptr;
    // per-point chains through the observation table (vo_kf.hip): obs_link[o] = int2(the previous, older observation of o's map point or -1, o's keyframe),
    // pt_last[slot] / pt_first[slot] = a point's newest / oldest observation (-1: none); d_kf_reach[kf] = (obs_lo, slot_lo) as int2
    int* d_obs_link = nullptr; int32_t* d_pt_last = nullptr; int32_t* d_pt_first = nullptr; int* d_kf_reach = nullptr;
    struct KfState* kf = nullptr;                           // buffers of vo_keyframe_commit / vo_map_set_active_covisible (vo_kf.hip)
    std::vector<long long> kf_first_obs;                    // host: where a keyframe's own observations begin in the table (-1: none yet)
    long long n_pre_calls = 0, n_pre_unpinned = 0, n_pre_frames = 0, n_up_hit = 0, n_up_copy = 0;      // VO_TRACE: how the frames reached the device (vo_frames_preload / vo_frame_upload)
    long long n_obs = 0, obs_cap = 0, obs_cap_max = 0;     // entries used / allocated / the bound the table may grow to
    // what a keyframe's points reach back to (the resident graph cut enters the tables there): kf_reach[kf] = minimum, over the points the keyframe
    // observes, of the table position of the point's first observation, and of the slot (host mirror of d_kf_reach, read back per keyframe)
    struct KfReach { long long obs_lo = -1; int slot_lo = 0; };
    std::vector<KfReach> kf_reach;
    double* d_kf_pose = nullptr; int n_kf = 0, kf_cap = 0;
    int map_hi = 0;                                         // highest map slot ever upserted + 1
    void* d_cut = nullptr; size_t d_cut_bytes = 0;          // scratch of the resident graph cut
    int cut_seq = 0;                                        // sequence number of the cut's pinned report words
    void* d_cut_sync = nullptr;                             // k_scan_one's published tile totals (call number << 32 | total): a block of its own, 4 KiB, zeroed once -- d_cut's carves move with the keyframe count and the window, and a word that held another array's data could be taken for a published total
    long long cut_slab_budget = 1ll << 30;                  // vo_ba_resident_set_slab_budget: a cut whose bound-sized slab would exceed it waits for the graph's sizes and carves exactly
    struct BaResident* resident = nullptr;                  // state between vo_local_ba_resident_cut and _solve (vo_ba.hip)
    // vo_track_batch_begin / _end: the request of the chain in flight (copies: the caller's arrays need not outlive _begin)
    bool async_pending = false; int async_n = 0, async_cap = 0; std::vector<int> async_slots; std::vector<uint64_t> async_seeds; double async_T0[12]; vo_track_params async_tp;
    // profiling
    std::atomic<bool> prof_on; std::mutex prof_mu; std::vector<ProfRec> prof; std::vector<hipEvent_t> ev_pool;      // prof / ev_pool: under prof_mu
    ProfRec prof_open; hipStream_t prof_open_stream = nullptr; uint64_t prof_ticket = 0, prof_closed = 0;
};

// profiling helpers (vo_capi.hip)
int vo_prof_begin(vo_ctx* c, const char* name, hipStream_t st);     // returns a ticket for vo_prof_end (-1: not recorded)
void vo_prof_end(vo_ctx* c, int ticket);
// the flag is latched at construction: a toggle from another thread between begin and end cannot unbalance the records
struct ProfScope { vo_ctx* c; int idx;
    ProfScope(vo_ctx* c_, const char* n, hipStream_t st = nullptr) : c(c_), idx(c_->prof_on.load(std::memory_order_relaxed) ? vo_prof_begin(c_, n, st) : -1) {}      // st: the stream the kernel runs on (default: the context's)
    ~ProfScope() { if (idx >= 0) vo_prof_end(c, idx); } };
// A non-blocking stream of one of the runtime's three priority classes (cls < 0: lowest, 0: default, > 0: highest).  The runtime keeps a pool of
// hardware queues per class (GPU_MAX_HW_QUEUES each; streams beyond that share queues of their class), so the class also says WHOSE queues a
// stream may share: the chains that set a stream's pace (a group's tracking chain, the BA engines' step launches) live in the highest class, where
// nothing with cross-stream waits or long batches is created beside them.
inline hipError_t vo_stream_create(hipStream_t* st, int cls) {
    if (cls == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // numerically lower = higher priority: `hi` is the highest, `lo` the lowest
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, cls > 0 ? hi : lo);
}
void* vo_stage(vo_ctx* c, size_t bytes);           // pinned host staging buffer of at least `bytes`
int vo_scratch(vo_ctx* c, size_t bytes);           // grow the device scratch slab c->d_ba to at least `bytes`
// nothing the context's BA engine has enqueued may still run when a buffer its step kernels read is freed: since round 6 a step kernel reads words of the slab
// BEFORE it knows that its problem has finished (one-trip heads), and a chunk's last launches may still be queued when the host has long moved on (vo_ba.hip)
void vo_ba_engine_drain(vo_ctx* c);
int vo_map_scatter_launch(vo_ctx* c, int n, const int32_t* d_idx, const double* d_xyz, const double* d_nrm, const uint32_t* d_desc, const int32_t* d_kp, const uint8_t* d_flags);

// stage launchers
int vo_orb_launch(vo_ctx* c, int slot0, int nslots);                                        // vo_orb.hip
int vo_orb_pyramid_plan(vo_ctx* c, const std::vector<int>& tab);
// vo_track.hip: the chain's stages over nl lanes described by d_lanes (device), on stream st; `prof` receives the timing records
struct ChainDims { int max_active, max_feat; };
int vo_track_match_launch(vo_ctx* prof, hipStream_t st, const LaneDesc* d_lanes, int nl, ChainDims dims, float ratio, float floor_dist);
// stage: 1 = hypotheses + scoring, 2 = adaptive-stop scan + inlier list, 3 = both; (rank, world): this process scores hypotheses h % world == rank
// corr_hint: upper bound of the lanes' match counts (sizes the scoring grid: the counts themselves are only known on the device)
int vo_track_ransac_launch(vo_ctx* prof, hipStream_t st, const LaneDesc* d_lanes, int nl, int n_hyp, float reproj_px, float conf, int pass, int stage = 3, int rank = 0, int world = 1, int corr_hint = 1 << 30, bool all_counts = false);
int vo_track_lm_launch(vo_ctx* prof, hipStream_t st, const LaneDesc* d_lanes, int nl, double delta, double cut, int it_r, int it_p, bool write_flags, int inlier_hint = 0);
#define VO_LM_X_DOUBLES (32 + 2 * 8 * 32)               // = LM_X_DOUBLES of vo_track.hip
void vo_lane_fill(vo_ctx* c, int lane, int slot, uint64_t seed, TrackDev* d_tr, LaneDesc* out);   // descriptor of lane `lane` of context c
int vo_corr_from_host(vo_ctx* c, const float* xyz, const float* uv, int n);
int vo_ba_run(vo_ctx* c, const vo_ba_problem* in, vo_ba_result* out);                       // vo_ba.hip
// One array that follows the map's capacity (vo_map_grow): `nl` lanes of `old_b` bytes become `nl` lanes of `new_b`; the first `keep_b` bytes of every lane are
// copied, the rest is filled with `fill` (< 0: left as allocated).  All fresh arrays are allocated before anything is copied or freed: a failed allocation leaves
// the context as it was.
struct MapRegrow { void** slot; size_t old_b, new_b, keep_b, nl; int fill; void* fresh; };
int vo_map_grow(vo_ctx* c, long long need_slots);          // vo_capi.hip: the map arrays and the per-lane chain buffers for at least `need_slots` map points (doubling; VO_E_OVERFLOW beyond VO_MAP_CAP_MAX); the caller has no chain in flight
void vo_kf_map_regrow_records(vo_ctx* c, size_t m_old, size_t m_new, size_t stride_new, std::vector<MapRegrow>& v);   // vo_kf.hip: chain heads, leader keys, candidate buffers
void vo_kf_map_regrown(vo_ctx* c, size_t m_new, size_t stride_new);      // ... and their bookkeeping once the new arrays are in place
int vo_scan_i32(hipStream_t st, const int* in, int n, int* bsum /* >= 1024 ints, 256-byte aligned, zeroed once when allocated (k_scan_one's published totals) */, int* out, int* total);      // vo_ba.hip: exclusive scan, n <= 16 Mi; *total = -1 if a tile of the one-launch form never showed up
// vo_kf.hip
int vo_obs_tables_ensure(vo_ctx* c);
void vo_kf_free(vo_ctx* c);
int vo_kf_host_pairs(vo_ctx* c, int** pair_a, int** pair_b, int* cap, int** n_total, double** poses);      // pinned block a BA merge reports into (vo_local_ba_resident_merge_ledger)
void vo_ba_resident_free(vo_ctx* c);
struct BaEngine* vo_ba_engine_acquire(int device);
void vo_ba_engine_release(struct BaEngine* e);

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "[vo_hip] %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return VO_E_DEVICE; } } while (0)
