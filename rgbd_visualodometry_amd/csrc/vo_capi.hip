// vo_capi.hip -- context management and the extern "C" entry points of include/vo_hip.h for the
// HIP (gfx950) implementation.  No CPU fallback exists in this library: without a usable HIP
// device vo_ctx_create returns VO_E_DEVICE.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <new>

#include "vo_internal.h"

int vo_orb_upload_constants();
int vo_track_set_attrs();
int vo_ba_set_attrs();

static inline short sat_short(long v) { return (short)std::min(32767L, std::max(-32768L, v)); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// Host-side plan: level sizes, quotas and the fixed-point tables (cv::ORB / cv::resize semantics,
// see SURVEY.md 8a-1).  Computed once per context; the kernels only read it.
static int build_plan(const vo_params& p, DevPlan& P, std::vector<int>& tab, std::vector<short>& tabs) {
    memset(&P, 0, sizeof(P));
    P.xcd_map = 1;
    P.W = p.width; P.H = p.height; P.L = p.n_levels; P.nfeat = p.n_features; P.fast_thr = p.fast_threshold; P.edge = p.edge_threshold;
    P.fx = p.fx; P.fy = p.fy; P.cx = p.cx; P.cy = p.cy;
    const double sf = (double)p.scale_factor;
    if (P.edge < 20) return VO_E_INVALID;       // k_describe copies rows y-19..y+19, columns x-19..x+20 of the blurred level around every keypoint
    unsigned off = 0;
    for (int l = 0; l < P.L; ++l) {
        P.scale[l] = (float)std::pow(sf, (double)l);
        P.lw[l] = (int)lrintf((float)P.W / P.scale[l]);
        P.lh[l] = (int)lrintf((float)P.H / P.scale[l]);
        if (P.lw[l] < 2 * P.edge + 1 || P.lh[l] < 2 * P.edge + 1 || P.lw[l] > 4095 || P.lh[l] > 4095) return VO_E_INVALID;
        P.pitch[l] = align_up(P.lw[l], 64);
        P.loff[l] = off;
        off += (unsigned)P.pitch[l] * (unsigned)P.lh[l];
        off = (off + 255u) & ~255u;
    }
    P.pyr_stride = off;
    float factor = (float)(1.0 / sf);
    float per = P.nfeat * (1 - factor) / (1 - (float)std::pow((double)factor, (double)P.L));
    int sum = 0;
    for (int l = 0; l < P.L - 1; ++l) { P.quota[l] = (int)lrintf(per); sum += P.quota[l]; per *= factor; }
    P.quota[P.L - 1] = std::max(P.nfeat - sum, 0);
    int maxq = 1;
    for (int l = 0; l < P.L; ++l) {
        P.qprefix[l + 1] = P.qprefix[l] + P.quota[l];
        // strict 3x3 maxima never touch (8-neighbourhood): at most every other pixel in x and in y survives the NMS, so
        // this capacity cannot overflow whatever the image (no arrival-order truncation, no error path)
        P.ccap[l] = std::max(4096, ((P.lw[l] + 1) / 2) * ((P.lh[l] + 1) / 2));
        P.cprefix[l + 1] = P.cprefix[l] + P.ccap[l];
        P.tiles_x[l] = (P.lw[l] - 2 * P.edge + 63) / 64;
        const int tiles_y = (P.lh[l] - 2 * P.edge + VO_FAST_TH - 1) / VO_FAST_TH;
        P.tile_prefix[l + 1] = P.tile_prefix[l] + P.tiles_x[l] * tiles_y;
        P.btiles_x[l] = (P.lw[l] + 127) / 128;
        P.btile_prefix[l + 1] = P.btile_prefix[l] + P.btiles_x[l] * ((P.lh[l] + 15) / 16);
        maxq = std::max(maxq, P.quota[l]);
    }
    P.sel_cap = 64;                                         // (k_select sorts whole blocks of 64)
    while (P.sel_cap < 4 * maxq) P.sel_cap <<= 1;
    if ((size_t)P.sel_cap * 12 + 4 * (256 + 8 + 4096) > 160 * 1024) return VO_E_INVALID;
    // bilinear tables
    tab.clear(); tabs.clear();
    for (int l = 1; l < P.L; ++l) {
        const int sw = P.lw[l - 1], sh = P.lh[l - 1], dw = P.lw[l], dh = P.lh[l];
        const double sx_ = (double)sw / dw, sy_ = (double)sh / dh;
        P.tabx[l] = (int)tab.size();
        for (int dx = 0; dx < dw; ++dx) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5);
            int sx = (int)std::floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            tab.push_back(sx);
            tabs.push_back(sat_short(lrintf((1.f - fx) * 2048.f))); tabs.push_back(sat_short(lrintf(fx * 2048.f)));
        }
        P.taby[l] = (int)tab.size();
        for (int dy = 0; dy < dh; ++dy) {
            float fy = (float)((dy + 0.5) * sy_ - 0.5);
            int sy = (int)std::floor(fy);
            fy -= sy;
            tab.push_back(sy);
            tabs.push_back(sat_short(lrintf((1.f - fy) * 2048.f))); tabs.push_back(sat_short(lrintf(fy * 2048.f)));
        }
    }
    const int hp = 15;
    const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) P.umax[v] = (int)lrint(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (P.umax[v0] == P.umax[v0 + 1]) ++v0;
        P.umax[v] = v0;
        ++v0;
    }
    P.umax_pk = 0;
    for (int v = 0; v <= hp; ++v) P.umax_pk |= (unsigned long long)P.umax[v] << (4 * v);
    double g[7], gs = 0;
    for (int i = 0; i < 7; ++i) { double x = i - 3; g[i] = (double)(float)std::exp(-0.5 / 4.0 * x * x); gs += g[i]; }
    for (int i = 0; i < 7; ++i) P.gk[i] = (int)lrintf((float)(g[i] / gs) * 256.f);
    return VO_OK;
}

void* vo_stage(vo_ctx* c, size_t bytes) {
    if (bytes <= c->h_stage_bytes) return c->h_stage;
    if (c->h_stage) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->h_stage); }     // an async copy may still read it
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 20);     // grows with the map: leave headroom, pinned allocations are slow
    if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) != hipSuccess) { c->h_stage = nullptr; return nullptr; }
    c->h_stage_bytes = want;
    return c->h_stage;
}

int vo_scratch(vo_ctx* c, size_t bytes) {
    if (bytes <= c->d_ba_bytes) return VO_OK;
    if (c->d_ba) { (void)hipStreamSynchronize(c->stream); vo_ba_engine_drain(c); (void)hipFree(c->d_ba); }
    c->d_ba = nullptr; c->d_ba_bytes = 0;
    const size_t want = bytes + bytes / 2 + (1 << 20);
    if (hipMalloc(&c->d_ba, want) != hipSuccess) return VO_E_NOMEM;
    c->d_ba_bytes = want;
    return VO_OK;
}

// ---- the map grows.  The reference's map is a host container without a size; here the map arrays (and the per-lane chain buffers, whose worst case is the
// whole map as the local map: src/frontend.cpp:163-166) are device allocations for `map_capacity` points.  vo_keyframe_commit -- the one place where map points
// are created on the device path -- calls this when a keyframe's new points may not fit: every array is reallocated at twice the size and its contents copied
// (device to device: 2 GB in well under a millisecond of copy time; the hipMalloc / hipFree pairs dominate, ~1 ms per array, once per doubling).  Every new array
// is allocated before anything is copied or freed, so a failed allocation (VO_E_NOMEM) leaves the context as it was.  The caller guarantees that no chain
// that reads these buffers is in flight (the host layer commits a keyframe behind the frame's own chain and behind the graph cut).
int vo_map_grow(vo_ctx* c, long long need) {
    const long long cap_max = 1ll << 28;                   // VO_MAP_CAP_MAX of the docs: 2^28 map points
    long long cap = c->p.map_capacity;
    if (need <= cap) return VO_OK;
    if (need > cap_max || need > (1ll << 30)) return VO_E_OVERFLOW;
    while (cap < need) cap = std::min(2 * cap, std::max(cap_max, need));
    const size_t m_old = (size_t)c->p.map_capacity, m0 = c->lane_stride, m1 = ((size_t)cap + 3) & ~(size_t)3, nl = (size_t)c->lanes;
    hipStream_t st = c->stream;
    std::vector<MapRegrow> v;
    auto rec = [&](auto*& p, size_t per, size_t lanes, int fill) {
        typedef typename std::remove_reference<decltype(*p)>::type T;
        v.push_back(MapRegrow{(void**)&p, sizeof(T) * per * m0, sizeof(T) * per * m1, sizeof(T) * per * m0, lanes, fill, nullptr});
    };
    rec(c->d_map_pos, 3, 1, -1); rec(c->d_map_nrm, 3, 1, -1); rec(c->d_map_desc, 8, 1, -1); rec(c->d_map_flags, 1, 1, 0); rec(c->d_active, 1, 1, -1);
    rec(c->d_best, 1, nl, -1); rec(c->d_mcand, 1, nl, -1); rec(c->d_matches, 1, nl, -1); rec(c->d_corr_xyz, 3, nl, -1); rec(c->d_corr_uv, 2, nl, -1);
    rec(c->d_inliers, 1, nl, -1); rec(c->d_lm_mask, 1, nl, -1);
    vo_kf_map_regrow_records(c, m_old, (size_t)cap, m1, v);
    for (MapRegrow& r : v)
        if (hipMalloc(&r.fresh, r.new_b * r.nl) != hipSuccess) {      // nothing has been touched yet: the context stays as it was
            (void)hipGetLastError();
            for (MapRegrow& q : v) if (q.fresh) (void)hipFree(q.fresh);
            return VO_E_NOMEM;
        }
    HIP_TRY(hipStreamSynchronize(st));
    for (MapRegrow& r : v)
        for (size_t l = 0; l < r.nl; ++l) {
            uint8_t* dst = (uint8_t*)r.fresh + l * r.new_b;
            if (r.keep_b && *r.slot) HIP_TRY(hipMemcpyAsync(dst, (const uint8_t*)*r.slot + l * r.old_b, r.keep_b, hipMemcpyDeviceToDevice, st));
            if (r.fill >= 0 && r.new_b > r.keep_b) HIP_TRY(hipMemsetAsync(dst + r.keep_b, r.fill, r.new_b - r.keep_b, st));
        }
    HIP_TRY(hipStreamSynchronize(st));
    for (MapRegrow& r : v) { if (*r.slot) (void)hipFree(*r.slot); *r.slot = r.fresh; }
    c->lane_stride = m1; c->p.map_capacity = (int32_t)cap; c->active_cap = (int)cap; c->corr_cap = (int)cap;
    vo_kf_map_regrown(c, (size_t)cap, m1);
    if (vo_trace_level()) fprintf(stderr, "[vo_trace] the device map grew to %lld points (%zu lanes)\n", cap, nl);
    return VO_OK;
}

// Profiling is process-wide: a VO system owns several contexts (tracker + overlapped back-end), and the
// per-kernel table has to cover all of them.  Contexts register themselves; reads merge every context.
#include <mutex>
#include <type_traits>
static std::mutex g_prof_mu;
static std::vector<vo_ctx*> g_ctxs;
static bool g_prof_on = false;
static std::vector<std::string> g_prof_names; static std::vector<double> g_prof_ms; static std::vector<int64_t> g_prof_calls;

// Records are appended by the context's owning thread (under the context's prof_mu) and drained by whichever thread
// reads the table; a record whose end event has not been recorded yet stays in the list.
int vo_prof_begin(vo_ctx* c, const char* name, hipStream_t st) {
    std::unique_lock<std::mutex> lk(c->prof_mu);
    if (!st) st = c->stream;
    ProfRec r; r.name = name;
    auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    r.a = get(); r.b = get();
    (void)hipEventRecord(r.a, st);
    c->prof.push_back(r);
    c->prof_open = r; c->prof_open_stream = st;
    return (int)(++c->prof_ticket & 0x3FFFFFFF);
}
void vo_prof_end(vo_ctx* c, int ticket) {
    std::unique_lock<std::mutex> lk(c->prof_mu);
    if (ticket != (int)(c->prof_ticket & 0x3FFFFFFF)) return;      // not the innermost open record any more (cannot happen: scopes do not nest)
    (void)hipEventRecord(c->prof_open.b, c->prof_open_stream);
    c->prof_closed = c->prof_ticket;
}

static void prof_collect(vo_ctx* c) {      // caller holds g_prof_mu
    std::unique_lock<std::mutex> lk(c->prof_mu);
    if (c->prof.empty()) return;
    (void)hipSetDevice(c->device);
    const bool open_tail = c->prof_closed != c->prof_ticket;       // the newest record still waits for its end event
    const size_t n = c->prof.size() - (open_tail ? 1 : 0);
    if (n == 0) return;
    (void)hipEventSynchronize(c->prof[n - 1].b);
    for (size_t k = 0; k < n; ++k) {
        const ProfRec& r = c->prof[k];
        float ms = 0;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        size_t i = 0;
        for (; i < g_prof_names.size(); ++i) if (g_prof_names[i] == r.name) break;
        if (i == g_prof_names.size()) { g_prof_names.push_back(r.name); g_prof_ms.push_back(0); g_prof_calls.push_back(0); }
        g_prof_ms[i] += ms; g_prof_calls[i] += 1;
        c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b);
    }
    c->prof.erase(c->prof.begin(), c->prof.begin() + n);
}

template <typename T>
static int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return VO_E_NOMEM;
    return VO_OK;
}
#define ALLOC(ptr, count) do { int rc_ = dev_alloc(&(ptr), (count)); if (rc_) { vo_ctx_destroy(c); return rc_; } } while (0)

static void launchset_free(LaunchSet& ls) {
    if (ls.d_lanes) (void)hipFree(ls.d_lanes);
    if (ls.d_track) (void)hipFree(ls.d_track);
    if (ls.h_lanes) (void)hipHostFree(ls.h_lanes);
    if (ls.h_track) (void)hipHostFree(ls.h_track);
    ls = LaunchSet();
}
static int launchset_alloc(LaunchSet& ls, int cap) {
    ls = LaunchSet(); ls.cap = cap;
    if (hipMalloc((void**)&ls.d_lanes, sizeof(LaneDesc) * cap) != hipSuccess || hipMalloc((void**)&ls.d_track, sizeof(TrackDev) * cap) != hipSuccess ||
        hipHostMalloc((void**)&ls.h_lanes, sizeof(LaneDesc) * cap, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&ls.h_track, sizeof(TrackDev) * cap, hipHostMallocDefault) != hipSuccess) { launchset_free(ls); return VO_E_NOMEM; }
    memset(ls.h_lanes, 0, sizeof(LaneDesc) * cap); memset(ls.h_track, 0, sizeof(TrackDev) * cap);
    return VO_OK;
}

// ---- stream group: the tracking requests of several contexts (independent streams on one GPU) fused into one launch
// chain.  Every member keeps its own host thread, map and buffers; a thread that calls vo_track_batch on a member context
// queues its lanes, and whichever thread finds the group idle becomes the leader: it takes everything queued so far,
// runs ONE chain for all of it on the group's stream, hands the results out and wakes the others.  Requests that arrive
// while a chain is in flight pile up for the next one, so the batch size follows the load and nobody waits on a timer.
#include <condition_variable>
struct GroupReq {
    vo_ctx* c; int n; const int* slots; const double* T0; const vo_track_params* tp; const uint64_t* seeds;
    vo_track_result* res; vo_match* matches; int cap; int rc; bool done;
    bool taken = false;                                     // some leader has put it into its chain
};
#define VO_GROUP_MAX_CHAINS 4
struct vo_group {
    int device = 0, max_lanes = 0;
    // chain slots: a launch chain in flight owns one (stream + launch set).  ONE slot: requests that arrive while a chain is running pile up
    // for the next one.  (Two chains side by side were measured in rounds 2 and 3 and lost: 8 / 16 streams 3202 / 3689 frames/s with one slot,
    // 2725 / 2864 with two -- more, smaller chains on a GPU that is already the bottleneck.)
    struct Slot { hipStream_t stream = nullptr; LaunchSet ls; bool busy = false; };
    Slot slot[VO_GROUP_MAX_CHAINS]; int n_slots = 1;
    std::mutex mu; std::condition_variable cv;
    std::vector<GroupReq*> pending;
    int members = 0, gather_min = 1; long gather_timeout_us = 0;
    int64_t n_chains = 0, n_lanes = 0, n_requests = 0;
};
static int chain_run(vo_ctx* prof, hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch);
static int shard_exchange(vo_ctx* c, hipStream_t st, int nl, int n_hyp);

extern "C" {

const char* vo_backend_name(void) { return "hip-gfx950"; }

int vo_trace_level(void) { static const int v = [] { const char* e = getenv("VO_TRACE"); return e ? std::max(1, atoi(e)) : 0; }(); return v; }

const char* vo_strerror(int s) {
    switch (s) {
        case VO_OK: return "ok"; case VO_E_INVALID: return "invalid argument"; case VO_E_NOMEM: return "out of memory";
        case VO_E_DEVICE: return "device error"; case VO_E_OVERFLOW: return "capacity overflow"; case VO_E_STATE: return "bad call sequence";
        case VO_E_UNSUPPORTED: return "unsupported"; default: return "unknown";
    }
}

int vo_default_params(vo_params* p) {
    if (!p) return VO_E_INVALID;
    memset(p, 0, sizeof(*p));
    p->width = 640; p->height = 480; p->fx = 517.3f; p->fy = 516.5f; p->cx = 318.6f; p->cy = 255.3f; p->depth_scale = 5000.f;
    p->n_features = 500; p->scale_factor = 1.2f; p->n_levels = 8; p->fast_threshold = 20; p->edge_threshold = 31;
    p->max_frames = 1; p->map_capacity = 1 << 18; p->max_hypotheses = 2048; p->max_track_batch = 1;
    return VO_OK;
}

int vo_default_track_params(vo_track_params* t) {
    if (!t) return VO_E_INVALID;
    memset(t, 0, sizeof(*t));
    t->match_ratio = 2.0f; t->match_floor = 30.0f; t->n_hyp = 100; t->reproj_px = 4.0f; t->confidence = 0.99f; t->seed = 0x5eed5eedull;
    t->huber_delta = std::sqrt(7.815); t->chi2_cut = 1.0; t->it_robust = 10; t->it_plain = 10; t->passes = 2;
    return VO_OK;
}

void vo_ctx_destroy(vo_ctx* c) {
    if (!c) return;
    if (vo_trace_level() && c->n_pre_calls) fprintf(stderr, "[vo_trace] uploads: %lld preload calls (%lld refused: not page-locked), %lld frames preloaded, %lld of them taken by vo_frame_upload, %lld frames copied by vo_frame_upload itself\n", c->n_pre_calls, c->n_pre_unpinned, c->n_pre_frames, c->n_up_hit, c->n_up_copy);
    { std::unique_lock<std::mutex> lk(g_prof_mu); prof_collect(c); for (size_t i = 0; i < g_ctxs.size(); ++i) if (g_ctxs[i] == c) { g_ctxs.erase(g_ctxs.begin() + i); break; } }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ba_engine) { vo_ba_engine_release(c->ba_engine); c->ba_engine = nullptr; }
    for (auto p : c->own_bgr) if (p) (void)hipFree(p);
    for (auto p : c->own_depth) if (p) (void)hipFree(p);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& sl : c->pre) { if (sl.bgr) (void)hipFree(sl.bgr); if (sl.depth) (void)hipFree(sl.depth); if (sl.ev) (void)hipEventDestroy(sl.ev); }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->orb_ev) (void)hipEventDestroy(c->orb_ev);
    void* ptrs[] = {c->d_slots, c->d_pyr, c->d_blur, c->d_tab, c->d_tabs, c->d_cand, c->d_cand_cnt, c->d_sel, c->d_sel_key, c->d_sel_cnt, c->d_kps,
                    c->d_desc, c->d_nkp, c->d_status, c->d_map_pos, c->d_map_nrm, c->d_map_desc, c->d_map_flags, c->d_active, c->d_best, c->d_mcand,
                    c->d_matches, c->d_corr_xyz, c->d_corr_uv, c->d_hyp_pose, c->d_hyp_cnt, c->d_inliers, c->d_lm_mask, c->d_lm_x, c->d_ba, c->d_pyr_rng};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    launchset_free(c->ls);
    if (c->group_ev) (void)hipEventDestroy(c->group_ev);
    if (c->h_matches) (void)hipHostFree(c->h_matches);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_slots_pinned) (void)hipHostFree(c->h_slots_pinned);
    if (c->h_orb_cache) (void)hipHostFree(c->h_orb_cache);
    if (c->h_ba_up) (void)hipHostFree(c->h_ba_up);
    if (c->d_ba_shard) (void)hipFree(c->d_ba_shard);
    if (c->h_ba_shard) (void)hipHostFree(c->h_ba_shard);
    vo_ba_resident_free(c);
    { void* tp[] = {c->d_obs_kf, c->d_obs_mp, c->d_obs_uv, c->d_obs_alive, c->d_obs_link, c->d_kf_pose, c->d_cut, c->d_cut_sync}; for (void* q : tp) if (q) (void)hipFree(q); }
    vo_kf_free(c);
    if (c->slots_ev) (void)hipEventDestroy(c->slots_ev);
    for (auto& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int vo_ctx_create(const vo_params* p, int device, vo_ctx** out) {
    if (!p || !out || p->width < 64 || p->height < 64 || p->n_levels < 1 || p->n_levels > VO_MAX_LEVELS || p->max_frames < 1 ||
        p->n_features < 1 || !(p->scale_factor > 1.0f) || p->map_capacity < 1 || p->max_hypotheses < 1 || p->max_track_batch > VO_MAX_LANES) return VO_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "[vo_hip] no usable HIP device (requested %d of %d): the HIP path has no CPU fallback\n", device, ndev);
        return VO_E_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    vo_ctx* c = new (std::nothrow) vo_ctx();
    if (!c) return VO_E_NOMEM;
    c->p = *p; c->device = device; c->stream = nullptr; c->prof_on.store(false); c->corr_external = false;
    c->d_slots = nullptr; c->d_pyr = nullptr; c->d_blur = nullptr; c->d_tab = nullptr; c->d_tabs = nullptr; c->d_cand = nullptr; c->d_cand_cnt = nullptr;
    c->d_sel = nullptr; c->d_sel_key = nullptr; c->d_sel_cnt = nullptr; c->d_kps = nullptr; c->d_desc = nullptr; c->d_nkp = nullptr; c->d_status = nullptr;
    c->d_map_pos = nullptr; c->d_map_nrm = nullptr; c->d_map_desc = nullptr; c->d_map_flags = nullptr; c->d_active = nullptr; c->n_active = 0; c->active_cap = 0;
    c->lanes = std::max(1, p->max_track_batch);
    c->d_best = nullptr; c->d_mcand = nullptr; c->d_matches = nullptr; c->d_corr_xyz = nullptr; c->d_corr_uv = nullptr; c->corr_cap = 0; c->d_hyp_pose = nullptr; c->d_hyp_cnt = nullptr;
    c->d_inliers = nullptr; c->d_lm_mask = nullptr; c->d_track = nullptr; c->h_track = nullptr; c->h_matches = nullptr; c->h_matches_cap = 0; c->lane_stride = 0;
    c->h_stage = nullptr; c->h_stage_bytes = 0; c->d_ba = nullptr; c->d_ba_bytes = 0;
    c->h_orb_cache = nullptr; c->orb_cache_valid = false; c->orb_batch0 = 0; c->orb_batchn = 0;
    c->h_slots_pinned = nullptr; c->slots_ev = nullptr; c->slots_dirty = false; c->slots_pending = false;
    std::vector<int> tab; std::vector<short> tabs;
    int rc = build_plan(*p, c->plan, tab, tabs);
    if (rc) { delete c; return rc; }
    if (vo_stream_create(&c->stream, p->stream_priority) != hipSuccess) { delete c; return VO_E_DEVICE; }
    rc = vo_orb_upload_constants();
    if (rc) { vo_ctx_destroy(c); return rc; }
    rc = vo_orb_pyramid_plan(c, tab);
    if (rc) { vo_ctx_destroy(c); return rc; }
    rc = vo_track_set_attrs();
    if (rc) { vo_ctx_destroy(c); return rc; }
    rc = vo_ba_set_attrs();
    if (rc) { vo_ctx_destroy(c); return rc; }
    const DevPlan& P = c->plan;
    const int F = p->max_frames;
    c->own_bgr.assign(F, nullptr); c->own_depth.assign(F, nullptr);
    c->h_slots.assign(F, SlotDesc{nullptr, nullptr, 0, 0}); c->slot_bound.assign(F, 0); c->slot_orb.assign(F, 0);
    ALLOC(c->d_slots, (size_t)F);
    if (hipHostMalloc((void**)&c->h_slots_pinned, sizeof(SlotDesc) * F, hipHostMallocDefault) != hipSuccess) { vo_ctx_destroy(c); return VO_E_NOMEM; }
    memset(c->h_slots_pinned, 0, sizeof(SlotDesc) * F);
    if (hipEventCreateWithFlags(&c->slots_ev, hipEventDisableTiming) != hipSuccess) { vo_ctx_destroy(c); return VO_E_DEVICE; }
    ALLOC(c->d_pyr, (size_t)F * P.pyr_stride);
    ALLOC(c->d_blur, (size_t)F * P.pyr_stride);
    ALLOC(c->d_tab, tab.size()); ALLOC(c->d_tabs, tabs.size());
    ALLOC(c->d_cand, (size_t)F * P.cprefix[P.L]); ALLOC(c->d_cand_cnt, (size_t)F * VO_MAX_LEVELS);
    ALLOC(c->d_sel, (size_t)F * P.nfeat); ALLOC(c->d_sel_key, (size_t)F * P.nfeat); ALLOC(c->d_sel_cnt, (size_t)F * VO_MAX_LEVELS);
    ALLOC(c->d_kps, (size_t)F * P.nfeat); ALLOC(c->d_desc, (size_t)F * P.nfeat * 32); ALLOC(c->d_nkp, (size_t)F);
    ALLOC(c->d_status, 1);
    const size_t M = ((size_t)p->map_capacity + 3) & ~(size_t)3;      // per-lane buffers stay 16-byte aligned (k_match_gate loads uint4)
    ALLOC(c->d_map_pos, 3 * M); ALLOC(c->d_map_nrm, 3 * M); ALLOC(c->d_map_desc, 8 * M); ALLOC(c->d_map_flags, M);
    c->active_cap = p->map_capacity; c->corr_cap = p->map_capacity;
    const size_t NL = (size_t)c->lanes;
    ALLOC(c->d_active, M); ALLOC(c->d_best, NL * M); ALLOC(c->d_mcand, NL * M); ALLOC(c->d_matches, NL * M); ALLOC(c->d_corr_xyz, NL * 3 * M); ALLOC(c->d_corr_uv, NL * 2 * M);
    ALLOC(c->d_hyp_pose, NL * 12 * p->max_hypotheses); ALLOC(c->d_hyp_cnt, NL * p->max_hypotheses);
    ALLOC(c->d_inliers, NL * M); ALLOC(c->d_lm_mask, NL * M); ALLOC(c->d_lm_x, NL * VO_LM_X_DOUBLES);
    if (launchset_alloc(c->ls, (int)NL) != VO_OK) { vo_ctx_destroy(c); return VO_E_NOMEM; }
    c->d_track = c->ls.d_track; c->h_track = c->ls.h_track; c->lane_stride = M;
    if (hipEventCreateWithFlags(&c->group_ev, hipEventDisableTiming) != hipSuccess) { vo_ctx_destroy(c); return VO_E_DEVICE; }
    if (!(c->ba_engine = vo_ba_engine_acquire(device))) { vo_ctx_destroy(c); return VO_E_NOMEM; }
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_tabs, tabs.data(), tabs.size() * sizeof(short), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(int), st));
    HIP_TRY(hipMemsetAsync(c->d_lm_x, 0, sizeof(double) * NL * VO_LM_X_DOUBLES, st));      // hand-off counters start at 0; the last workgroup of a launch clears them again
    HIP_TRY(hipMemsetAsync(c->d_nkp, 0, sizeof(int) * F, st));
    HIP_TRY(hipMemsetAsync(c->d_sel_cnt, 0, sizeof(int) * F * VO_MAX_LEVELS, st));
    HIP_TRY(hipMemsetAsync(c->d_map_flags, 0, M, st));
    HIP_TRY(hipMemsetAsync(c->d_track, 0, sizeof(TrackDev) * NL, st));
    HIP_TRY(hipStreamSynchronize(st));
    { std::unique_lock<std::mutex> lk(g_prof_mu); g_ctxs.push_back(c); c->prof_on.store(g_prof_on); }
    *out = c;
    return VO_OK;
}

// Slot descriptors are mirrored in pinned memory and pushed lazily, once, in front of the next ORB launch.
static int mark_slot(vo_ctx* c, int slot) {
    if (c->slots_pending) { HIP_TRY(hipEventSynchronize(c->slots_ev)); c->slots_pending = false; }
    c->h_slots_pinned[slot] = c->h_slots[slot];
    c->slots_dirty = true;
    return VO_OK;
}
static int push_slots(vo_ctx* c) {
    if (!c->slots_dirty) return VO_OK;
    HIP_TRY(hipMemcpyAsync(c->d_slots, c->h_slots_pinned, sizeof(SlotDesc) * c->p.max_frames, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->slots_ev, c->stream));
    c->slots_dirty = false; c->slots_pending = true;
    return VO_OK;
}

// The uploads of a whole look-ahead batch ahead of time, on the context's copy stream, into one of two slabs: the slots' current frames
// (and everything enqueued on the context's stream) go on undisturbed.  Evenly spaced host frames (one array of frames) travel as ONE
// two-dimensional copy per image kind -- per-frame API calls, not bytes, were what the per-slot upload cost the caller's thread (64 copies
// and 64 pointer queries per batch of 32: ~2 ms of host time per 21 ms of frames).  A later vo_frame_upload of a slot from the same host
// buffers finds the frame on the device: it points the slot into the slab and makes the context's stream wait for the copy's event once.
int vo_frames_preload(vo_ctx* c, int slot0, int n, const uint8_t* const* bgr, int bs, const uint16_t* const* depth, int ds) {
    if (!c || slot0 < 0 || n < 1 || slot0 + n > c->p.max_frames || !bgr || !depth || bs < 3 * c->p.width || ds < 2 * c->p.width) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) if (!bgr[i] || !depth[i]) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, bgr[0]) == hipSuccess && at.type == hipMemoryTypeHost && hipPointerGetAttributes(&at, depth[0]) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    ++c->n_pre_calls;
    if (!pinned) { ++c->n_pre_unpinned; return VO_OK; }      // pageable memory copies synchronously: nothing to gain, vo_frame_upload does it
    c->n_pre_frames += n;
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->orb_ev) HIP_TRY(hipEventCreateWithFlags(&c->orb_ev, hipEventDisableTiming));
    const int F = c->p.max_frames, H = c->p.height;
    if ((int)c->pre_slot.size() != F) { c->pre_slot.assign(F, vo_ctx::PreSlot{}); c->slot_gen.assign(F, -1); }
    const int g = c->pre_next;
    vo_ctx::PreSlab& sl = c->pre[g];
    const size_t nb = (size_t)bs * H, nd = (size_t)ds * H;
    if (sl.nb != nb || sl.nd != nd || !sl.bgr) {            // (first use, or the strides changed: nobody may still be reading the slab)
        HIP_TRY(hipStreamSynchronize(c->copy_stream)); HIP_TRY(hipStreamSynchronize(c->stream));
        if (sl.bgr) (void)hipFree(sl.bgr); if (sl.depth) (void)hipFree(sl.depth);
        sl.bgr = sl.depth = nullptr; sl.nb = sl.nd = 0;
        if (hipMalloc((void**)&sl.bgr, nb * F) != hipSuccess || hipMalloc((void**)&sl.depth, nd * F) != hipSuccess) return VO_E_NOMEM;
        sl.nb = nb; sl.nd = nd;
    }
    if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    // a slot that still shows a frame of this slab (not rebound for two batches) loses it: its next ORB launch fails instead of reading the new bytes
    for (int i = 0; i < F; ++i) if (c->slot_gen[i] == g) { c->slot_gen[i] = -1; c->slot_bound[i] = 0; }
    // the slab's previous frames were read by ORB chains enqueued before the last recorded orb_ev
    if (c->orb_ev_set) HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->orb_ev, 0));
    const size_t cb = (size_t)bs * (H - 1) + 3 * (size_t)c->p.width, cd = (size_t)ds * (H - 1) + 2 * (size_t)c->p.width;
    bool even = n >= 2;
    const ptrdiff_t db = n >= 2 ? (const uint8_t*)bgr[1] - (const uint8_t*)bgr[0] : 0, dd = n >= 2 ? (const uint8_t*)depth[1] - (const uint8_t*)depth[0] : 0;
    for (int i = 1; i + 1 < n && even; ++i) even = ((const uint8_t*)bgr[i + 1] - (const uint8_t*)bgr[i]) == db && ((const uint8_t*)depth[i + 1] - (const uint8_t*)depth[i]) == dd;
    even = even && db >= (ptrdiff_t)cb && dd >= (ptrdiff_t)cd;
    if (even) {
        HIP_TRY(hipMemcpy2DAsync(sl.bgr + nb * slot0, nb, bgr[0], (size_t)db, cb, n, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipMemcpy2DAsync(sl.depth + nd * slot0, nd, depth[0], (size_t)dd, cd, n, hipMemcpyHostToDevice, c->copy_stream));
    } else {
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipMemcpyAsync(sl.bgr + nb * (slot0 + i), bgr[i], cb, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipMemcpyAsync(sl.depth + nd * (slot0 + i), depth[i], cd, hipMemcpyHostToDevice, c->copy_stream));
        }
    }
    HIP_TRY(hipEventRecord(sl.ev, c->copy_stream));
    sl.waited = false;
    for (int i = 0; i < n; ++i) c->pre_slot[slot0 + i] = vo_ctx::PreSlot{bgr[i], depth[i], bs, ds, g, true};
    c->pre_next = g ^ 1;
    return VO_OK;
}

int vo_frame_upload(vo_ctx* c, int slot, const uint8_t* bgr, int bs, const uint16_t* depth, int ds) {
    if (!c || slot < 0 || slot >= c->p.max_frames || !bgr || !depth || bs < 3 * c->p.width || ds < 2 * c->p.width) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    const int H = c->p.height;
    if (slot < (int)c->pre_slot.size()) {
        vo_ctx::PreSlot& ps = c->pre_slot[slot];
        c->slot_gen[slot] = -1;
        if (ps.valid) {
            ps.valid = false;
            if (ps.src_bgr == bgr && ps.src_depth == depth && ps.bs == bs && ps.ds == ds) {      // preloaded: the slot shows the slab's frame; the copy is waited for on the device
                vo_ctx::PreSlab& sl = c->pre[ps.gen];
                if (!sl.waited) { HIP_TRY(hipStreamWaitEvent(c->stream, sl.ev, 0)); sl.waited = true; }
                ++c->n_up_hit;
                c->h_slots[slot] = SlotDesc{sl.bgr + sl.nb * slot, sl.depth + sl.nd * slot, bs, ds};
                c->slot_bound[slot] = 1; c->slot_orb[slot] = 0; c->slot_gen[slot] = (signed char)ps.gen;
                return mark_slot(c, slot);
            }
        }
    }
    // The slot keeps the caller's strides (the kernels take any pitch, as for vo_frame_bind_device): each image travels as ONE
    // contiguous copy -- a pitch-converting 2-D copy of 480 rows cost 0.14 ms per image.  From page-locked memory the copies are
    // asynchronous and the call does not wait (include/vo_hip.h); pageable sources are staged by the runtime, the wait is then free.
    const size_t nb = (size_t)bs * H, nd = (size_t)ds * H;
    if (c->own_bgr_bytes.size() != c->own_bgr.size()) { c->own_bgr_bytes.assign(c->own_bgr.size(), 0); c->own_depth_bytes.assign(c->own_depth.size(), 0); }
    if (c->own_bgr_bytes[slot] < nb) {
        if (c->own_bgr[slot]) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipFree(c->own_bgr[slot]); c->own_bgr[slot] = nullptr; c->own_bgr_bytes[slot] = 0; }
        if (hipMalloc((void**)&c->own_bgr[slot], nb) != hipSuccess) return VO_E_NOMEM;
        c->own_bgr_bytes[slot] = nb;
    }
    if (c->own_depth_bytes[slot] < nd) {
        if (c->own_depth[slot]) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipFree(c->own_depth[slot]); c->own_depth[slot] = nullptr; c->own_depth_bytes[slot] = 0; }
        if (hipMalloc((void**)&c->own_depth[slot], nd) != hipSuccess) return VO_E_NOMEM;
        c->own_depth_bytes[slot] = nd;
    }
    // only (H - 1) strides + one row of pixels are the caller's to read: the padding behind the LAST row need not exist (a column
    // slice of a wider image, a cv::Mat ROI); the slot itself is stride * H so that every row has its pitch
    const size_t cb = (size_t)bs * (H - 1) + 3 * (size_t)c->p.width, cd = (size_t)ds * (H - 1) + 2 * (size_t)c->p.width;
    ++c->n_up_copy;
    HIP_TRY(hipMemcpyAsync(c->own_bgr[slot], bgr, cb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->own_depth[slot], depth, cd, hipMemcpyHostToDevice, c->stream));
    c->h_slots[slot] = SlotDesc{c->own_bgr[slot], c->own_depth[slot], bs, ds};
    c->slot_bound[slot] = 1; c->slot_orb[slot] = 0;
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, bgr) == hipSuccess && at.type == hipMemoryTypeHost && hipPointerGetAttributes(&at, depth) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();                                // (an unregistered pointer makes the query fail: that is the pageable case)
    if (!pinned) HIP_TRY(hipStreamSynchronize(c->stream));  // pageable host copies done: the caller may reuse its buffers
    return mark_slot(c, slot);
}

int vo_frame_bind_device(vo_ctx* c, int slot, const void* b, int bs, const void* d, int ds) {
    if (!c || slot < 0 || slot >= c->p.max_frames || !b || !d || bs < 3 * c->p.width || ds < 2 * c->p.width) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    c->h_slots[slot] = SlotDesc{(const uint8_t*)b, (const uint8_t*)d, bs, ds};
    if (slot < (int)c->slot_gen.size()) c->slot_gen[slot] = -1;
    c->slot_bound[slot] = 1; c->slot_orb[slot] = 0;
    return mark_slot(c, slot);
}

int vo_orb_detect_describe(vo_ctx* c, int slot0, int n) {
    if (!c || slot0 < 0 || n < 1 || slot0 + n > c->p.max_frames) return VO_E_INVALID;
    for (int i = slot0; i < slot0 + n; ++i) if (!c->slot_bound[i]) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    int rc = push_slots(c);
    if (rc) return rc;
    rc = vo_orb_launch(c, slot0, n);
    if (rc) return rc;
    if (c->orb_ev) { HIP_TRY(hipEventRecord(c->orb_ev, c->stream)); c->orb_ev_set = true; }      // (vo_frame_preload: the frames of these slots have been read once this has passed)
    for (int i = slot0; i < slot0 + n; ++i) c->slot_orb[i] = 1;
    c->orb_cache_valid = false; c->orb_batch0 = slot0; c->orb_batchn = n;
    return VO_OK;
}

static int read_status(vo_ctx* c) {
    int* st = (int*)vo_stage(c, 64);
    if (!st) return VO_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(st, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return *st;
}

// Results of the whole last ORB batch come down in ONE set of D2H copies into a pinned cache; the
// per-slot fetches of a look-ahead batch are then plain host memcpys.
int vo_orb_fetch(vo_ctx* c, int slot, vo_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    if (!c || slot < 0 || slot >= c->p.max_frames || !n_out || cap < 0) return VO_E_INVALID;
    if (!c->slot_orb[slot]) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    const int N = c->p.n_features, F = c->p.max_frames;
    const size_t o_n = 0, o_st = 4 * (size_t)F, o_k = (o_st + 64 + 255) & ~(size_t)255, o_d = o_k + sizeof(vo_keypoint) * (size_t)F * N;
    if (!c->h_orb_cache) {
        if (hipHostMalloc((void**)&c->h_orb_cache, o_d + (size_t)32 * F * N, hipHostMallocDefault) != hipSuccess) return VO_E_NOMEM;
    }
    // The results of the whole ORB batch come down together on the first fetch of any of its slots (pinned cache): the
    // copy rides behind the ORB launch chain, off the per-keyframe path; later fetches are host memcpys.
    const bool in_batch = slot >= c->orb_batch0 && slot < c->orb_batch0 + c->orb_batchn;
    if (!c->orb_cache_valid || !in_batch) {
        const int s0 = in_batch ? c->orb_batch0 : slot, sn = in_batch ? c->orb_batchn : 1;
        uint8_t* h = c->h_orb_cache;
        HIP_TRY(hipMemcpyAsync(h + o_n + 4 * (size_t)s0, c->d_nkp + s0, sizeof(int) * sn, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h + o_st, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h + o_k + sizeof(vo_keypoint) * (size_t)s0 * N, c->d_kps + (size_t)s0 * N, sizeof(vo_keypoint) * (size_t)sn * N, hipMemcpyDeviceToHost, c->stream));
        if (desc) HIP_TRY(hipMemcpyAsync(h + o_d + (size_t)32 * s0 * N, c->d_desc + (size_t)s0 * N * 32, (size_t)32 * sn * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (in_batch) { c->orb_cache_valid = true; c->orb_cache_desc = desc != nullptr; }
    } else if (desc && !c->orb_cache_desc) {                // the batch came down without its descriptors (callers that keep them on the device)
        HIP_TRY(hipMemcpyAsync(c->h_orb_cache + o_d + (size_t)32 * c->orb_batch0 * N, c->d_desc + (size_t)c->orb_batch0 * N * 32, (size_t)32 * c->orb_batchn * N, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->orb_cache_desc = true;
    }
    const uint8_t* h = c->h_orb_cache;
    const int st = *(const int*)(h + o_st);
    if (st != VO_OK) return st;
    const int n = ((const int*)(h + o_n))[slot], k = std::min(n, std::min(cap, N));
    if (kps) memcpy(kps, h + o_k + sizeof(vo_keypoint) * (size_t)slot * N, sizeof(vo_keypoint) * k);
    if (desc) memcpy(desc, h + o_d + (size_t)32 * slot * N, (size_t)32 * k);
    *n_out = n;
    return VO_OK;
}

int vo_orb_level_size(vo_ctx* c, int l, int* w, int* h, int* quota) {
    if (!c || l < 0 || l >= c->plan.L) return VO_E_INVALID;
    if (w) *w = c->plan.lw[l];
    if (h) *h = c->plan.lh[l];
    if (quota) *quota = c->plan.quota[l];
    return VO_OK;
}

int vo_orb_fetch_level(vo_ctx* c, int slot, int l, uint8_t* out) {
    if (!c || slot < 0 || slot >= c->p.max_frames || l < 0 || l >= c->plan.L || !out) return VO_E_INVALID;
    if (!c->slot_orb[slot]) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    const DevPlan& P = c->plan;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy2D(out, P.lw[l], c->d_pyr + (size_t)slot * P.pyr_stride + P.loff[l], P.pitch[l], P.lw[l], P.lh[l], hipMemcpyDeviceToHost));
    return VO_OK;
}

int vo_orb_fetch_blur_level(vo_ctx* c, int slot, int l, uint8_t* out) {
    if (!c || slot < 0 || slot >= c->p.max_frames || l < 0 || l >= c->plan.L || !out) return VO_E_INVALID;
    if (!c->slot_orb[slot]) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    const DevPlan& P = c->plan;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy2D(out, P.lw[l], c->d_blur + (size_t)slot * P.pyr_stride + P.loff[l], P.pitch[l], P.lw[l], P.lh[l], hipMemcpyDeviceToHost));
    return VO_OK;
}

int vo_map_upsert(vo_ctx* c, const int32_t* idx, const double* xyz, const double* nrm, const uint8_t* desc, const uint8_t* flags, int n) {
    if (!c || n < 0 || (n && !idx)) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= c->p.map_capacity) return VO_E_INVALID;
    if (n == 0) return VO_OK;
    for (int i = 0; i < n; ++i) c->map_hi = std::max(c->map_hi, idx[i] + 1);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // pack -> one pinned staging buffer -> one H2D copy -> scatter kernel
    const size_t N = (size_t)n;
    const size_t o_idx = 0, o_xyz = (4 * N + 255) & ~(size_t)255, o_nrm = o_xyz + ((24 * N + 255) & ~(size_t)255),
                 o_desc = o_nrm + ((24 * N + 255) & ~(size_t)255), o_flags = o_desc + ((32 * N + 255) & ~(size_t)255),
                 total = o_flags + ((N + 255) & ~(size_t)255);
    uint8_t* h = (uint8_t*)vo_stage(c, total);
    if (!h) return VO_E_NOMEM;
    int rc = vo_scratch(c, total);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));                      // staging buffer may still feed an earlier copy
    memcpy(h + o_idx, idx, 4 * N);
    if (xyz) memcpy(h + o_xyz, xyz, 24 * N);
    if (nrm) memcpy(h + o_nrm, nrm, 24 * N);
    if (desc) memcpy(h + o_desc, desc, 32 * N);
    if (flags) memcpy(h + o_flags, flags, N);
    uint8_t* d = (uint8_t*)c->d_ba;
    HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st));
    rc = vo_map_scatter_launch(c, n, (const int32_t*)(d + o_idx), xyz ? (const double*)(d + o_xyz) : nullptr, nrm ? (const double*)(d + o_nrm) : nullptr,
                               desc ? (const uint32_t*)(d + o_desc) : nullptr, nullptr, flags ? d + o_flags : nullptr);
    if (rc) return rc;
    // no wait here: the copy and the scatter are ordered before any later work on this stream, and the next user of the
    // staging buffer (this function, vo_ba_run, vo_corr_from_host) drains the stream before touching it
    return VO_OK;
}

int vo_map_upsert_from_frame(vo_ctx* c, int slot, const int32_t* kp, const int32_t* idx, const double* xyz, const double* nrm, const uint8_t* flags, int n) {
    if (!c || n < 0 || (n && (!idx || !kp)) || slot < 0 || slot >= c->p.max_frames) return VO_E_INVALID;
    if (!c->slot_orb[slot]) return VO_E_STATE;
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= c->p.map_capacity || kp[i] < 0 || kp[i] >= c->p.n_features) return VO_E_INVALID;
    if (n == 0) return VO_OK;
    for (int i = 0; i < n; ++i) c->map_hi = std::max(c->map_hi, idx[i] + 1);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t N = (size_t)n;
    const size_t o_idx = 0, o_kp = (4 * N + 255) & ~(size_t)255, o_xyz = o_kp + ((4 * N + 255) & ~(size_t)255), o_nrm = o_xyz + ((24 * N + 255) & ~(size_t)255),
                 o_flags = o_nrm + ((24 * N + 255) & ~(size_t)255), total = o_flags + ((N + 255) & ~(size_t)255);
    uint8_t* h = (uint8_t*)vo_stage(c, total);
    if (!h) return VO_E_NOMEM;
    int rc = vo_scratch(c, total);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));                      // staging buffer may still feed an earlier copy
    memcpy(h + o_idx, idx, 4 * N); memcpy(h + o_kp, kp, 4 * N);
    if (xyz) memcpy(h + o_xyz, xyz, 24 * N);
    if (nrm) memcpy(h + o_nrm, nrm, 24 * N);
    if (flags) memcpy(h + o_flags, flags, N);
    uint8_t* d = (uint8_t*)c->d_ba;
    HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st));
    // the slot's descriptor rows (keypoint order) are the source: a keypoint index beyond the frame's count reads a stale row of
    // the same table, never outside it
    return vo_map_scatter_launch(c, n, (const int32_t*)(d + o_idx), xyz ? (const double*)(d + o_xyz) : nullptr, nrm ? (const double*)(d + o_nrm) : nullptr,
                                 (const uint32_t*)(c->d_desc + (size_t)slot * c->p.n_features * 32), (const int32_t*)(d + o_kp), flags ? d + o_flags : nullptr);
}

int vo_map_set_active(vo_ctx* c, const int32_t* idx, int n) {
    if (!c || n < 0 || (n && !idx) || n > c->active_cap) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= c->p.map_capacity) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    if (n) HIP_TRY(hipMemcpyAsync(c->d_active, idx, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_active = n;
    return VO_OK;
}

// ---- the context's own launch set (lane 0 .. lanes-1) ---------------------------------------------------------------
static int upload_pose(vo_ctx* c, int nl, const double T[12], bool reset) {
    // whole headers are rewritten: counters start from zero for a new frame
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nl; ++i) {
        TrackDev* h = c->h_track + i;
        if (reset) memset(h, 0, sizeof(*h));
        memcpy(h->T, T, sizeof(double) * 12);
        memcpy(h->T_ransac, T, sizeof(double) * 12);
    }
    HIP_TRY(hipMemcpyAsync(c->d_track, c->h_track, sizeof(TrackDev) * nl, hipMemcpyHostToDevice, c->stream));
    return VO_OK;
}

// lane descriptors of the context's own lanes (stream must be idle: the pinned mirror is rewritten)
static int upload_lanes(vo_ctx* c, int nl, const int* slots, const uint64_t* seeds) {
    for (int i = 0; i < nl; ++i) vo_lane_fill(c, i, slots ? slots[i] : 0, seeds ? seeds[i] : 0, c->d_track + i, c->ls.h_lanes + i);
    HIP_TRY(hipMemcpyAsync(c->ls.d_lanes, c->ls.h_lanes, sizeof(LaneDesc) * nl, hipMemcpyHostToDevice, c->stream));
    return VO_OK;
}

static int download_track(vo_ctx* c, int nl = 1) {
    HIP_TRY(hipMemcpyAsync(c->h_track, c->d_track, sizeof(TrackDev) * nl, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VO_OK;
}

static int ensure_match_stage(vo_ctx* c, int n) {
    if (n <= c->h_matches_cap) return VO_OK;
    if (c->h_matches) (void)hipHostFree(c->h_matches);
    c->h_matches = nullptr; c->h_matches_cap = 0;
    const int want = std::max(n, 4096);
    if (hipHostMalloc((void**)&c->h_matches, sizeof(vo_match) * (size_t)want, hipHostMallocDefault) != hipSuccess) return VO_E_NOMEM;
    c->h_matches_cap = want;
    return VO_OK;
}

int vo_match_active_map(vo_ctx* c, int slot, const double T[12], float ratio, float floor_dist, vo_match* out, int cap,
                        int* n_out, int* n_cand, int* min_distance) {
    if (!c || slot < 0 || slot >= c->p.max_frames || !T || cap < 0) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;                  // a chain started by vo_track_batch_begin owns the lane buffers and the pinned mirrors until vo_track_batch_end
    if (!c->slot_orb[slot]) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    int rc = upload_pose(c, 1, T, true);
    if (rc) return rc;
    if ((rc = upload_lanes(c, 1, &slot, nullptr))) return rc;
    c->corr_external = false;
    rc = vo_track_match_launch(c, c->stream, c->ls.d_lanes, 1, ChainDims{c->n_active, c->p.n_features}, ratio, floor_dist);
    if (rc) return rc;
    const int take = std::min(cap, c->n_active);
    rc = ensure_match_stage(c, take);
    if (rc) return rc;
    if (take > 0 && out) HIP_TRY(hipMemcpyAsync(c->h_matches, c->d_matches, sizeof(vo_match) * take, hipMemcpyDeviceToHost, c->stream));
    rc = download_track(c);
    if (rc) return rc;
    const TrackDev& t = *c->h_track;
    if (out) memcpy(out, c->h_matches, sizeof(vo_match) * std::min(take, t.n_match));
    if (n_out) *n_out = t.n_match;
    if (n_cand) *n_cand = t.n_cand;
    if (min_distance) *min_distance = t.min_dist;
    return t.status;
}

int vo_matches_set(vo_ctx* c, const float* xyz, const float* uv, int n) {
    if (!c || n < 0 || (n && (!xyz || !uv))) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;                  // a chain started by vo_track_batch_begin owns the lane buffers and the pinned mirrors until vo_track_batch_end
    HIP_TRY(hipSetDevice(c->device));
    return vo_corr_from_host(c, xyz, uv, n);
}

int vo_pnp_ransac(vo_ctx* c, int n_hyp, float reproj_px, float conf, uint64_t seed, double T[12], int32_t* inl, int cap,
                  int* n_inl, int32_t* hyp_counts, int* iters_used, int* best_hyp) {
    if (!c || !T || n_hyp < 1 || n_hyp > c->p.max_hypotheses || cap < 0) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;                  // a chain started by vo_track_batch_begin owns the lane buffers and the pinned mirrors until vo_track_batch_end
    HIP_TRY(hipSetDevice(c->device));
    // keep n_match of the current correspondence set; only the pose is replaced
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = download_track(c);
    if (rc) return rc;
    memcpy(c->h_track->T, T, sizeof(double) * 12);
    HIP_TRY(hipMemcpyAsync(c->d_track, c->h_track, sizeof(TrackDev), hipMemcpyHostToDevice, c->stream));
    if ((rc = upload_lanes(c, 1, nullptr, &seed))) return rc;
    if (c->shard_world > 1) {
        if ((rc = vo_track_ransac_launch(c, c->stream, c->ls.d_lanes, 1, n_hyp, reproj_px, conf, 0, 1, c->shard_rank, c->shard_world, c->h_track->n_match))) return rc;
        if ((rc = shard_exchange(c, c->stream, 1, n_hyp))) return rc;
        rc = vo_track_ransac_launch(c, c->stream, c->ls.d_lanes, 1, n_hyp, reproj_px, conf, 0, 2);
    } else rc = vo_track_ransac_launch(c, c->stream, c->ls.d_lanes, 1, n_hyp, reproj_px, conf, 0, 3, 0, 1, c->h_track->n_match, hyp_counts != nullptr);
    if (rc) return rc;
    rc = download_track(c);
    if (rc) return rc;
    const TrackDev t = *c->h_track;
    memcpy(T, t.T, sizeof(double) * 12);
    const int take = std::min(cap, t.n_inl);
    if (take > 0 && inl) HIP_TRY(hipMemcpy(inl, c->d_inliers, sizeof(int32_t) * take, hipMemcpyDeviceToHost));
    if (hyp_counts) HIP_TRY(hipMemcpy(hyp_counts, c->d_hyp_cnt, sizeof(int32_t) * n_hyp, hipMemcpyDeviceToHost));
    if (n_inl) *n_inl = t.n_inl;
    if (iters_used) *iters_used = t.iters_used;
    if (best_hyp) *best_hyp = t.best_hyp;
    return VO_OK;
}

int vo_pose_refine_lm(vo_ctx* c, double T[12], double delta, double cut, int it_r, int it_p, uint8_t* mask, int cap,
                      int* n_edges, int* lm_iters) {
    if (!c || !T || cap < 0) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;                  // a chain started by vo_track_batch_begin owns the lane buffers and the pinned mirrors until vo_track_batch_end
    HIP_TRY(hipSetDevice(c->device));
    int rc = download_track(c);
    if (rc) return rc;
    memcpy(c->h_track->T, T, sizeof(double) * 12);
    c->h_track->lm_iters = 0;
    HIP_TRY(hipMemcpyAsync(c->d_track, c->h_track, sizeof(TrackDev), hipMemcpyHostToDevice, c->stream));
    if ((rc = upload_lanes(c, 1, nullptr, nullptr))) return rc;
    rc = vo_track_lm_launch(c, c->stream, c->ls.d_lanes, 1, delta, cut, it_r, it_p, false, c->h_track->n_inl);
    if (rc) return rc;
    rc = download_track(c);
    if (rc) return rc;
    const TrackDev t = *c->h_track;
    memcpy(T, t.T, sizeof(double) * 12);
    const int take = std::min(cap, t.n_inl);
    if (take > 0 && mask) {
        std::vector<uint8_t> tmp(take);
        HIP_TRY(hipMemcpy(tmp.data(), c->d_lm_mask, take, hipMemcpyDeviceToHost));
        for (int i = 0; i < take; ++i) mask[i] = tmp[i] & 1;
    }
    if (n_edges) *n_edges = t.n_inl;
    if (lm_iters) *lm_iters = t.lm_iters;
    return VO_OK;
}

#define MATCH_COPY_FIRST 4096       // floor of the per-lane match copy that travels with the headers; a frame with more gets a second copy

// ---- one launch chain over the lanes of a set of requests (one request = the lanes one context contributes) ----------
// Used for a context's own vo_track_batch (one request, its own stream and launch set) and for the fused chain of a
// stream group (its stream and launch set).  Returns after the chain has finished and every request's results are filled in.
static int chain_enqueue(vo_ctx* prof, hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch);
static int chain_collect(hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch);
static int chain_run_impl(vo_ctx* prof, hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch) {
    const int rc = chain_enqueue(prof, st, ls, batch);
    return rc ? rc : chain_collect(st, ls, batch);
}
// A failing call inside the chain returns early with kernels / copies of the fused chain possibly still in flight on the shared stream:
// the stream is drained before the members are told (they may free or reuse their lane buffers at once).
static int chain_run(vo_ctx* prof, hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch) {
    const int rc = chain_run_impl(prof, st, ls, batch);
    if (rc != VO_OK) (void)hipStreamSynchronize(st);
    return rc;
}
// everything up to the read-back copies is enqueued; nothing waits (vo_track_batch_begin returns here)
static int chain_enqueue(vo_ctx* prof, hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch) {
    const vo_track_params* tp = batch[0]->tp;
    int nl = 0; ChainDims dims{0, 0};
    HIP_TRY(hipStreamSynchronize(st));                      // the pinned mirrors are rewritten below
    for (GroupReq* r : batch) {
        vo_ctx* c = r->c;
        c->corr_external = false;
        for (int i = 0; i < r->n; ++i, ++nl) {
            TrackDev* h = ls.h_track + nl;
            memset(h, 0, sizeof(*h));
            memcpy(h->T, r->T0, sizeof(double) * 12); memcpy(h->T_ransac, r->T0, sizeof(double) * 12);
            vo_lane_fill(c, i, r->slots[i], 0, ls.d_track + nl, ls.h_lanes + nl);
        }
        dims.max_active = std::max(dims.max_active, c->n_active); dims.max_feat = std::max(dims.max_feat, c->p.n_features);
        if (c->stream != st) {                              // a member's own stream carries its ORB results and map updates
            HIP_TRY(hipEventRecord(c->group_ev, c->stream));
            HIP_TRY(hipStreamWaitEvent(st, c->group_ev, 0));
        }
    }
    HIP_TRY(hipMemcpyAsync(ls.d_track, ls.h_track, sizeof(TrackDev) * nl, hipMemcpyHostToDevice, st));
    int rc = VO_OK;
    { int k = 0; for (GroupReq* r : batch) for (int i = 0; i < r->n; ++i, ++k) ls.h_lanes[k].seed = r->seeds ? r->seeds[i] : r->tp->seed; }      // the request's own seed: same_track_params does not compare it
    HIP_TRY(hipMemcpyAsync(ls.d_lanes, ls.h_lanes, sizeof(LaneDesc) * nl, hipMemcpyHostToDevice, st));
    int lm_hint = 0;                                        // the inlier sets are not known on the host yet: the largest recent match count bounds them
    for (GroupReq* r : batch) lm_hint = std::max(lm_hint, r->c->match_hint);
    for (int pass = 0; pass < tp->passes; ++pass) {           // coarse, fine (frontend.cpp:100-108); the sampler's seed is lane seed + pass
        if ((rc = vo_track_match_launch(prof, st, ls.d_lanes, nl, dims, tp->match_ratio, tp->match_floor))) return rc;
        vo_ctx* sc = batch[0]->c;                           // hypothesis sharding applies to un-grouped contexts (one request per chain)
        if (sc->shard_world > 1 && batch.size() == 1) {
            if ((rc = vo_track_ransac_launch(prof, st, ls.d_lanes, nl, tp->n_hyp, tp->reproj_px, tp->confidence, pass, 1, sc->shard_rank, sc->shard_world, dims.max_active))) return rc;
            if ((rc = shard_exchange(sc, st, nl, tp->n_hyp))) return rc;
            if ((rc = vo_track_ransac_launch(prof, st, ls.d_lanes, nl, tp->n_hyp, tp->reproj_px, tp->confidence, pass, 2))) return rc;
        } else if ((rc = vo_track_ransac_launch(prof, st, ls.d_lanes, nl, tp->n_hyp, tp->reproj_px, tp->confidence, pass, 3, 0, 1, dims.max_active))) return rc;      // matches <= candidates <= active points
        if ((rc = vo_track_lm_launch(prof, st, ls.d_lanes, nl, tp->huber_delta, tp->chi2_cut, tp->it_robust, tp->it_plain, pass == tp->passes - 1, lm_hint))) return rc;
    }
    // match records that callers asked for travel with the headers; sized from the largest match count seen recently
    // (+25 %): consecutive frames see the same map, and a synchronous second copy per lane costs more than the extra bytes
    const bool grouped = batch[0]->c->stream != st;         // a stream group's chain
    for (GroupReq* r : batch) {
        vo_ctx* c = r->c;
        c->h_matches_lanes = 0;
        // Members of a group get the records of every lane with the chain's own read-back: a later vo_track_fetch_matches
        // is then a host memcpy.  (An on-demand copy on the member's stream queues behind the other streams' ORB / BA copies.)
        if (!r->matches && !grouped) continue;
        const int first = std::min(std::min(r->cap, c->n_active), std::max(MATCH_COPY_FIRST, c->match_hint + c->match_hint / 4));
        if ((rc = ensure_match_stage(c, std::max(first * r->n, std::min(r->cap, c->n_active))))) return rc;
        if (first > 0)
            HIP_TRY(hipMemcpy2DAsync(c->h_matches, sizeof(vo_match) * (size_t)first, c->d_matches, sizeof(vo_match) * c->lane_stride, sizeof(vo_match) * (size_t)first, r->n, hipMemcpyDeviceToHost, st));
        if (!r->matches) { c->h_matches_lanes = r->n; c->h_matches_first = first; }
    }
    HIP_TRY(hipMemcpyAsync(ls.h_track, ls.d_track, sizeof(TrackDev) * nl, hipMemcpyDeviceToHost, st));
    return VO_OK;
}
// wait for the chain and hand the results out
static int chain_collect(hipStream_t st, LaunchSet& ls, std::vector<GroupReq*>& batch) {
    HIP_TRY(hipStreamSynchronize(st));
    int k0 = 0;
    for (GroupReq* r : batch) {
        vo_ctx* c = r->c;
        int seen = 0;
        for (int i = 0; i < r->n; ++i) { c->h_track[i] = ls.h_track[k0 + i]; seen = std::max(seen, c->h_track[i].n_match); }
        c->last_track_lanes = r->n;
        const int first = std::min(std::min(r->cap, c->n_active), std::max(MATCH_COPY_FIRST, c->match_hint + c->match_hint / 4));
        c->match_hint = std::max(seen, c->match_hint - c->match_hint / 16);      // slow decay
        for (int i = 0; i < r->n; ++i) {
            const TrackDev& t = c->h_track[i];
            vo_track_result& o = r->res[i];
            memset(&o, 0, sizeof(o));
            memcpy(o.T_cw, t.T, sizeof(double) * 12);
            o.n_candidates = t.n_cand; o.n_matches = t.n_match; o.n_ransac_inliers = t.n_inl; o.n_lm_inliers = t.n_lm_inl;
            o.min_distance = t.min_dist; o.ransac_iters = t.iters_used; o.best_hypothesis = t.best_hyp; o.lm_iters = t.lm_iters;
            o.status = t.status;
#ifdef VO_LM_STAMPS
            for (int k = 0; k < 7; ++k) o.reserved[k] = (int32_t)(t.dbg[k == 6 ? 7 : k] >> ((k == 2 || k == 3) ? 0 : 4));
            o.n_lm_inliers = (int32_t)(t.dbg[6] >> 4);
#endif
            // a pose-LM hand-off that ran out (k_pose_lm, lm_xchg) leaves this lane's counters behind: cleared before the next chain reads them
            if (t.status == VO_E_DEVICE && c->d_lm_x) HIP_TRY(hipMemsetAsync(c->d_lm_x + (size_t)i * VO_LM_X_DOUBLES, 0, sizeof(double) * VO_LM_X_DOUBLES, st));
            if (t.n_match > r->cap && r->matches) o.status = VO_E_OVERFLOW;
            if (r->matches) memcpy(r->matches + (size_t)i * r->cap, c->h_matches + (size_t)i * first, sizeof(vo_match) * std::min(first, t.n_match));
        }
        if (r->matches)
            for (int i = 0; i < r->n; ++i) {                  // rare: more matches than the first copy carried
                const int want = std::min(r->cap, c->h_track[i].n_match);
                if (want > first) {
                    HIP_TRY(hipMemcpy(c->h_matches, c->d_matches + (size_t)i * c->lane_stride, sizeof(vo_match) * (size_t)want, hipMemcpyDeviceToHost));
                    memcpy(r->matches + (size_t)i * r->cap, c->h_matches, sizeof(vo_match) * (size_t)want);
                }
            }
        k0 += r->n;
    }
    return VO_OK;
}

// the one exchange step of a hypothesis-sharded RANSAC pass: per-hypothesis inlier counts of every lane, summed over the ranks
static int shard_exchange(vo_ctx* c, hipStream_t st, int nl, int n_hyp) {
    if (c->shard_stream_fn)                                 // enqueued on the chain's stream, in place on the count table: no host round trip
        return c->shard_stream_fn(c->shard_user, c->d_hyp_cnt, (size_t)nl * c->p.max_hypotheses, (void*)st) == 0 ? VO_OK : VO_E_DEVICE;
    const size_t n = (size_t)nl * n_hyp;
    int32_t* h = (int32_t*)vo_stage(c, sizeof(int32_t) * n);
    if (!h) return VO_E_NOMEM;
    HIP_TRY(hipMemcpy2DAsync(h, sizeof(int32_t) * n_hyp, c->d_hyp_cnt, sizeof(int32_t) * c->p.max_hypotheses, sizeof(int32_t) * n_hyp, nl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->shard_fn(c->shard_user, h, (int)n);
    HIP_TRY(hipMemcpy2DAsync(c->d_hyp_cnt, sizeof(int32_t) * c->p.max_hypotheses, h, sizeof(int32_t) * n_hyp, sizeof(int32_t) * n_hyp, nl, hipMemcpyHostToDevice, st));
    return VO_OK;
}

static bool same_track_params(const vo_track_params* a, const vo_track_params* b) {
    return a->match_ratio == b->match_ratio && a->match_floor == b->match_floor && a->n_hyp == b->n_hyp && a->reproj_px == b->reproj_px &&
           a->confidence == b->confidence && a->huber_delta == b->huber_delta && a->chi2_cut == b->chi2_cut && a->it_robust == b->it_robust &&
           a->it_plain == b->it_plain && a->passes == b->passes;
}

// a member's request: queue it; lead a chain when the group is idle, otherwise sleep until some leader has served it
static int group_submit(vo_group* g, GroupReq* req) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->pending.push_back(req);
    ++g->n_requests;
    g->cv.notify_all();                                     // a leader gathering requests counts this one
    while (!req->done) {
        int k = -1;
        if (!req->taken) for (int i = 0; i < g->n_slots; ++i) if (!g->slot[i].busy) { k = i; break; }
        if (k < 0) { g->cv.wait(lk); continue; }            // its chain is already running, or every chain slot is busy
        g->slot[k].busy = true;
        if (g->gather_min > 1 && g->gather_timeout_us > 0) {  // optional: wait a moment for the other members (tests, tuning)
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(g->gather_timeout_us);
            g->cv.wait_until(lk, until, [&] { return (int)g->pending.size() >= std::min(g->gather_min, g->members); });
        }
        std::vector<GroupReq*> batch;
        int lanes = 0;
        for (size_t i = 0; i < g->pending.size();) {          // oldest first; same solver parameters only
            GroupReq* r = g->pending[i];
            if ((batch.empty() || same_track_params(batch[0]->tp, r->tp)) && lanes + r->n <= g->max_lanes) { batch.push_back(r); lanes += r->n; r->taken = true; g->pending.erase(g->pending.begin() + i); }
            else ++i;
        }
        if (batch.empty()) { g->slot[k].busy = false; g->cv.notify_all(); continue; }    // another leader took everything while this one gathered
        lk.unlock();
        int rc = VO_OK;
        if (hipSetDevice(g->device) != hipSuccess) rc = VO_E_DEVICE;
        if (rc == VO_OK) rc = chain_run(req->c, g->slot[k].stream, g->slot[k].ls, batch);
        lk.lock();
        ++g->n_chains; g->n_lanes += lanes;
        for (GroupReq* r : batch) { r->rc = rc; r->done = true; }
        g->slot[k].busy = false;
        g->cv.notify_all();
    }
    return req->rc;
}

int vo_track_batch(vo_ctx* c, int n, const int* slots, const double T0[12], const vo_track_params* tp, const uint64_t* seeds,
                   vo_track_result* res, vo_match* matches, int cap) {
    if (!c || n < 1 || n > c->lanes || !slots || !T0 || !tp || !res || tp->passes < 1 || cap < 0 || tp->n_hyp < 1 || tp->n_hyp > c->p.max_hypotheses) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;                  // a chain started by vo_track_batch_begin owns the lane buffers until vo_track_batch_end
    for (int i = 0; i < n; ++i) { if (slots[i] < 0 || slots[i] >= c->p.max_frames) return VO_E_INVALID; if (!c->slot_orb[slots[i]]) return VO_E_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    GroupReq req{c, n, slots, T0, tp, seeds, res, matches, cap, VO_OK, false};
    if (c->group) return group_submit(c->group, &req);
    std::vector<GroupReq*> one{&req};
    return chain_run(c, c->stream, c->ls, one);
}

// The same call in two halves for a caller that has host work to do while the chain runs (host/src/frontend.cpp: at a keyframe the next
// frames' chain is started before the BA write-back).  Un-grouped contexts only: a group's chain is led by whichever member finds it idle.
int vo_track_batch_begin(vo_ctx* c, int n, const int* slots, const double T0[12], const vo_track_params* tp, const uint64_t* seeds, int cap) {
    if (!c || n < 1 || n > c->lanes || !slots || !T0 || !tp || tp->passes < 1 || cap < 0 || tp->n_hyp < 1 || tp->n_hyp > c->p.max_hypotheses) return VO_E_INVALID;
    if (c->group) return VO_E_UNSUPPORTED;
    if (c->async_pending) return VO_E_STATE;
    for (int i = 0; i < n; ++i) { if (slots[i] < 0 || slots[i] >= c->p.max_frames) return VO_E_INVALID; if (!c->slot_orb[slots[i]]) return VO_E_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    c->async_slots.assign(slots, slots + n); c->async_seeds.clear(); if (seeds) c->async_seeds.assign(seeds, seeds + n);
    memcpy(c->async_T0, T0, sizeof(double) * 12); c->async_tp = *tp; c->async_n = n; c->async_cap = cap;
    GroupReq req{c, n, c->async_slots.data(), c->async_T0, &c->async_tp, seeds ? c->async_seeds.data() : nullptr, nullptr, nullptr, cap, VO_OK, false};
    std::vector<GroupReq*> one{&req};
    const int rc = chain_enqueue(c, c->stream, c->ls, one);
    if (rc != VO_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    c->async_pending = true;
    return VO_OK;
}
int vo_track_batch_end(vo_ctx* c, vo_track_result* res) {
    if (!c || !res) return VO_E_INVALID;
    if (!c->async_pending) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    c->async_pending = false;
    GroupReq req{c, c->async_n, c->async_slots.data(), c->async_T0, &c->async_tp, c->async_seeds.empty() ? nullptr : c->async_seeds.data(), res, nullptr, c->async_cap, VO_OK, false};
    std::vector<GroupReq*> one{&req};
    return chain_collect(c->stream, c->ls, one);
}

int vo_track_fetch_matches(vo_ctx* c, int lane, vo_match* matches, int cap, int* n_out) {
    if (!c || lane < 0 || lane >= c->last_track_lanes || !matches || cap < 0 || !n_out) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    const int n = std::min(cap, c->h_track[lane].n_match);
    if (lane < c->h_matches_lanes && n <= c->h_matches_first) {      // the chain's read-back already holds this lane's records
        memcpy(matches, c->h_matches + (size_t)lane * c->h_matches_first, sizeof(vo_match) * (size_t)n);
        *n_out = n;
        return VO_OK;
    }
    c->h_matches_lanes = 0;                                 // the staging buffer is reused below
    int rc = ensure_match_stage(c, std::max(n, 1));
    if (rc) return rc;
    if (n > 0) {
        // (On the context's own stream the copy queues behind whatever has been enqueued since -- 77 us per keyframe in the bench's VO_TRACE
        // scopes.  A stream of its own was tried and lost: one more stream, even in the lowest priority class, changed which streams share
        // a hardware queue -- every step kernel of the local BA a third slower by HIP events, upload_inclusive 0.97 -> 0.75, -3 % frames/s.)
        HIP_TRY(hipMemcpyAsync(c->h_matches, c->d_matches + (size_t)lane * c->lane_stride, sizeof(vo_match) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(matches, c->h_matches, sizeof(vo_match) * (size_t)n);
    }
    *n_out = n;
    return VO_OK;
}

int vo_track_frame(vo_ctx* c, int slot, const double T0[12], const vo_track_params* tp, vo_track_result* res,
                   vo_match* matches, int cap) {
    if (!tp) return VO_E_INVALID;
    return vo_track_batch(c, 1, &slot, T0, tp, &tp->seed, res, matches, cap);
}

// ---- stream groups ----------------------------------------------------------------------------------------------------
int vo_group_create(int device, int max_lanes, vo_group** out) {
    if (!out || max_lanes < 1 || max_lanes > VO_GROUP_MAX_LANES) return VO_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VO_E_DEVICE;
    HIP_TRY(hipSetDevice(device));
    vo_group* g = new (std::nothrow) vo_group();
    if (!g) return VO_E_NOMEM;
    g->device = device; g->max_lanes = max_lanes;
    g->n_slots = 1;
    for (int i = 0; i < g->n_slots; ++i) {
        if (vo_stream_create(&g->slot[i].stream, 1) != hipSuccess) { vo_group_destroy(g); return VO_E_DEVICE; }      // the members' pace: a queue of the highest class
        if (launchset_alloc(g->slot[i].ls, max_lanes) != VO_OK) { vo_group_destroy(g); return VO_E_NOMEM; }
    }
    *out = g;
    return VO_OK;
}

void vo_group_destroy(vo_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (int i = 0; i < VO_GROUP_MAX_CHAINS; ++i) {
        if (g->slot[i].stream) { (void)hipStreamSynchronize(g->slot[i].stream); (void)hipStreamDestroy(g->slot[i].stream); }
        launchset_free(g->slot[i].ls);
    }
    delete g;
}

int vo_set_hypothesis_shard(vo_ctx* c, int rank, int world, vo_exchange_fn fn, void* user) {
    if (!c || world < 0 || (world > 1 && (rank < 0 || rank >= world || !fn))) return VO_E_INVALID;
    if (world > 1 && c->group) return VO_E_UNSUPPORTED;      // a context either shares launch chains with other streams or shares a stream with other ranks
    c->shard_rank = rank; c->shard_world = world > 1 ? world : 1; c->shard_fn = fn; c->shard_user = user; c->shard_stream_fn = nullptr;
    return VO_OK;
}

// e-3: the local BA sharded over ranks by point (vo_ba.hip: ba_shard_solve)
int vo_set_ba_shard(vo_ctx* c, int rank, int world, vo_exchange_f64_fn fn, void* user) {
    if (!c || world < 0 || world > 64 || (world > 1 && (rank < 0 || rank >= world || !fn))) return VO_E_INVALID;
    c->ba_shard_rank = world > 1 ? rank : 0; c->ba_shard_world = world > 1 ? world : 1; c->ba_shard_fn = world > 1 ? fn : nullptr; c->ba_shard_stream_fn = nullptr; c->ba_shard_user = user;
    return VO_OK;
}
int vo_set_ba_shard_stream(vo_ctx* c, int rank, int world, vo_stream_allreduce_f64_fn fn, void* comm) {
    if (!c || world < 0 || world > 64 || (world > 1 && (rank < 0 || rank >= world || !fn))) return VO_E_INVALID;
    c->ba_shard_rank = world > 1 ? rank : 0; c->ba_shard_world = world > 1 ? world : 1; c->ba_shard_fn = nullptr; c->ba_shard_stream_fn = world > 1 ? fn : nullptr; c->ba_shard_user = comm;
    return VO_OK;
}

int vo_set_hypothesis_shard_stream(vo_ctx* c, int rank, int world, vo_stream_allreduce_fn fn, void* comm) {
    if (!c || world < 0 || (world > 1 && (rank < 0 || rank >= world || !fn))) return VO_E_INVALID;
    if (world > 1 && c->group) return VO_E_UNSUPPORTED;
    c->shard_rank = rank; c->shard_world = world > 1 ? world : 1; c->shard_fn = nullptr; c->shard_user = comm; c->shard_stream_fn = world > 1 ? fn : nullptr;
    return VO_OK;
}

int vo_group_join(vo_group* g, vo_ctx* c) {
    if (!g || !c || c->group || c->shard_world > 1 || c->device != g->device || c->lanes > g->max_lanes) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;                  // its own chain is still in flight (vo_track_batch_begin)
    std::unique_lock<std::mutex> lk(g->mu);
    c->group = g; ++g->members;
    return VO_OK;
}

int vo_group_leave(vo_group* g, vo_ctx* c) {
    if (!g || !c || c->group != g) return VO_E_INVALID;
    std::unique_lock<std::mutex> lk(g->mu);
    c->group = nullptr; --g->members;
    g->cv.notify_all();
    return VO_OK;
}

int vo_group_set_gather(vo_group* g, int min_requests, int timeout_us) {
    if (!g || min_requests < 1 || timeout_us < 0) return VO_E_INVALID;
    std::unique_lock<std::mutex> lk(g->mu);
    g->gather_min = min_requests; g->gather_timeout_us = timeout_us;
    return VO_OK;
}

int vo_group_stats(vo_group* g, int64_t* chains, int64_t* lanes, int64_t* requests) {
    if (!g) return VO_E_INVALID;
    std::unique_lock<std::mutex> lk(g->mu);
    if (chains) *chains = g->n_chains;
    if (lanes) *lanes = g->n_lanes;
    if (requests) *requests = g->n_requests;
    return VO_OK;
}

int vo_local_ba(vo_ctx* c, const vo_ba_problem* in, vo_ba_result* out) {
    if (!c || !in || !out || !out->poses || !out->points || !out->edge_flags) return VO_E_INVALID;
    if (in->n_free < 0 || in->n_free > in->n_poses || in->n_points < 0 || in->n_edges < 0) return VO_E_INVALID;
    for (int e = 0; e < in->n_edges; ++e)
        if (in->edge_pose[e] < 0 || in->edge_pose[e] >= in->n_poses || in->edge_point[e] < 0 || in->edge_point[e] >= in->n_points) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    return vo_ba_run(c, in, out);
}

int vo_sync(vo_ctx* c) {
    if (!c) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return read_status(c);
}

int vo_profile_enable(vo_ctx* c, int on) {
    if (!c) return VO_E_INVALID;
    std::unique_lock<std::mutex> lk(g_prof_mu);
    for (vo_ctx* x : g_ctxs) { prof_collect(x); x->prof_on.store(on != 0); }
    g_prof_on = on != 0;
    if (on) { g_prof_names.clear(); g_prof_ms.clear(); g_prof_calls.clear(); }
    return VO_OK;
}

int vo_profile_read(vo_ctx* c, char (*names)[48], double* ms, int64_t* calls, int cap, int* n) {
    if (!c || !n) return VO_E_INVALID;
    std::unique_lock<std::mutex> lk(g_prof_mu);
    for (vo_ctx* x : g_ctxs) prof_collect(x);
    const int k = std::min<int>(cap, (int)g_prof_names.size());
    for (int i = 0; i < k; ++i) {
        if (names) { strncpy(names[i], g_prof_names[i].c_str(), 47); names[i][47] = 0; }
        if (ms) ms[i] = g_prof_ms[i];
        if (calls) calls[i] = g_prof_calls[i];
    }
    *n = k;
    return VO_OK;
}

}  // extern "C"
