// vo_kf.hip -- the keyframe bookkeeping of the reference as device-resident tables on gfx950.
//
// The reference keeps keyframes and map points as shared_ptr objects that point at each other (src/frame.cpp:93-171,
// src/mappoint.cpp:17-49, src/mapmanager.cpp:14-38) and walks them at every keyframe.  Here the same state is four flat tables in HBM:
//
//   observation table   (keyframe number, map slot, pixel, alive) per observation, append-only; an observation's index is its id
//   per-point chains    obs_link[o] = (the previous, older observation of the same map point; its keyframe), pt_last[slot] = its newest one,
//                       pt_first[slot] = its oldest one: "who else sees this point" is a walk of 2 .. ~20 links instead of a table scan
//   map SoA             position, mean viewing direction, descriptor, flags (outlier / triangulated / optimised) per slot
//   keyframe poses      T_cw per keyframe number
//
// and a keyframe is ONE launch sequence on the tracker's stream (vo_keyframe_commit):
//   k_kf_count, k_kf_place   a lane per record: LM-inlier matches -> observations + viewing directions (src/frontend.cpp:366-370, src/frame.cpp:93-120,
//                 src/mappoint.cpp:30-38); unmatched keypoints with depth -> new map points (src/frontend.cpp:372-406, src/camera.cpp:41-86);
//                 ordered by per-block counts + ballot ranks, so slots and observation ids are those of the host loop
//   k_kf_covis_tri  two independent jobs side by side: one lane per new observation walks the point's chain, +1 for every keyframe that already sees it
//                 (src/frame.cpp:104-119); one lane per candidate of the reference's triangulation loop (src/frontend.cpp:465-506), views gathered along the chain
//   k_kf_finish   one workgroup: the covisibility weights as an ascending (keyframe, weight) list and the FIRST successful triangulation
//                 (the reference's loop breaks there) go straight into pinned host memory with the counts
// The local-map query (src/mapmanager.cpp:14-38) is vo_map_set_active_covisible, the ledger side of a BA merge (src/frame.cpp:122-152)
// k_merge_ledger in vo_ba.hip.  Nothing here is bandwidth-bound (a keyframe touches ~0.5 MB): the point is that the tracker's thread
// spends ~0.1 ms waiting for five small launches instead of 0.85 ms chasing pointers.
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>

#include "vo_internal.h"
#include "vo_tri_dev.h"

// ---- table management -------------------------------------------------------------------------------------------------------------
#define VO_OBS_CAP (256ll << 20)    // observations the table may grow to (25 B each: 6.4 GB); ~5000 per keyframe at the bench workload: ~170 000 frames
#define VO_KF_CAP 65536             // keyframes (96 B each)
// The observation table starts at VO_OBS_CAP0 entries and doubles when a keyframe does not fit, up to VO_OBS_CAP (env VO_OBS_CAP lowers
// that bound: tests exercise the overflow path with it).  Growing copies the live prefix on the context's stream and waits for it: nothing
// may be reading the table then -- the front-end appends only behind Backend::WaitGraphCut, and merges are ordered on this stream.
#define VO_OBS_CAP0 (4ll << 20)     // 100 MB; ~800 keyframes of the bench workload
#define KF_COVIS_CAP 4096           // partners of one keyframe that fit the pinned result block
#define KF_PAIR_CAP 16384           // ledger decrements of one BA merge
#define KF_LIST_CAP 4096            // keyframes of one local-map query

struct KfDev {                      // kernel-to-kernel header of a commit (device memory)
    int n_matched, n_new, n_tri, reach_obs, reach_slot, pad[3];
};
struct KfHost {                     // pinned, written by the finishing kernels, read by the host behind a stream wait
    vo_kf_commit_result r;
    int reach_obs, reach_slot, n_covis_total, n_active, n_pairs_total, pad[3];
    int covis_kf[KF_COVIS_CAP], covis_w[KF_COVIS_CAP];
    int pair_a[KF_PAIR_CAP], pair_b[KF_PAIR_CAP];
    int kf_list[KF_LIST_CAP];
    double poses[12 * VO_BA_RESIDENT_MAX_FREE];
};
struct KfState {
    KfHost* h = nullptr;            // pinned
    KfDev* d_hdr = nullptr;
    int32_t* d_w = nullptr; int32_t* d_mark = nullptr;              // [kf_cap] covisibility counters (zero between calls), epoch marks
    unsigned long long* d_key = nullptr;                            // [map_capacity] leader keys of the local-map query
    int32_t* d_cand = nullptr; uint8_t* d_tri_ok = nullptr; double* d_tri_xyz = nullptr; int cand_cap = 0;
    int2* d_cnt = nullptr; unsigned* d_kp_bits = nullptr;           // per 256 matches: (LM inliers, triangulation candidates); keypoints explained by an inlier (zero between commits)
    void* d_act = nullptr; size_t act_bytes = 0;                    // flag / position scratch of the local-map query
    // vo_scan_i32's published tile totals (call number << 32 | total) and the scan's total: a block of its OWN, allocated once and zeroed.  Up to round 6 it was
    // carved out of d_act behind the position array, i.e. at an offset that follows the window: the words then held an earlier, larger window's prefix sums -- values
    // up to the local map's size, ~18 000 -- and once the process-wide call number had counted up to that range (three scans per keyframe: keyframe ~6000, frame
    // ~20 000) a tile's stale word could pass for this call's published total: wrong places, a wrong n_active, and the chain indexing the map with whatever
    // d_active held behind the written entries: the GPU memory fault of two 30 000-frame soaks in sixteen (profiles/r06_soak_30k_fault.txt).
    void* d_scan = nullptr;                                         // [4096 B tile totals | 256 B total]
    uint32_t epoch = 0;
};

static int obs_tables_alloc(long long cap, int32_t** kf, int32_t** mp, float** uv, uint8_t** alive, int** prev) {
    *kf = nullptr; *mp = nullptr; *uv = nullptr; *alive = nullptr; *prev = nullptr;
    if (hipMalloc((void**)kf, 4 * (size_t)cap) != hipSuccess || hipMalloc((void**)mp, 4 * (size_t)cap) != hipSuccess ||
        hipMalloc((void**)uv, 8 * (size_t)cap) != hipSuccess || hipMalloc((void**)alive, (size_t)cap) != hipSuccess || hipMalloc((void**)prev, 8 * (size_t)cap) != hipSuccess) {
        // all or nothing: a later call must not find one array set beside null siblings
        void* q[] = {*kf, *mp, *uv, *alive, *prev};
        for (void* x : q) if (x) (void)hipFree(x);
        *kf = nullptr; *mp = nullptr; *uv = nullptr; *alive = nullptr; *prev = nullptr;
        (void)hipGetLastError();
        return VO_E_NOMEM;
    }
    return VO_OK;
}
// the keyframe-commit state alone (kf_state's failure path: the chain heads and the keyframes' reach belong to the observation tables, which stay)
static void kf_state_free(vo_ctx* c) {
    KfState* k = c->kf;
    if (!k) return;
    if (k->h) (void)hipHostFree(k->h);
    void* q[] = {k->d_hdr, k->d_w, k->d_mark, k->d_key, k->d_cand, k->d_tri_ok, k->d_tri_xyz, k->d_act, k->d_cnt, k->d_kp_bits, k->d_scan};
    for (void* x : q) if (x) (void)hipFree(x);
    delete k; c->kf = nullptr;
}
void vo_kf_free(vo_ctx* c) {                 // vo_ctx_destroy: the tables' chain heads / reach, then the commit state
    if (c->d_pt_last) (void)hipFree(c->d_pt_last);
    if (c->d_pt_first) (void)hipFree(c->d_pt_first);
    if (c->d_kf_reach) (void)hipFree(c->d_kf_reach);
    c->d_pt_last = nullptr; c->d_pt_first = nullptr; c->d_kf_reach = nullptr;
    kf_state_free(c);
}
int vo_obs_tables_ensure(vo_ctx* c) {     // allocates the observation / keyframe tables on first use
    if (c->d_obs_kf) return VO_OK;
    const char* env = getenv("VO_OBS_CAP");
    c->obs_cap_max = env && atoll(env) > 0 ? std::min<long long>(atoll(env), VO_OBS_CAP) : VO_OBS_CAP; c->kf_cap = VO_KF_CAP;
    const char* env0 = getenv("VO_OBS_CAP0");             // tests start small to exercise the growth
    const long long cap = std::min<long long>(env0 && atoll(env0) > 0 ? atoll(env0) : VO_OBS_CAP0, c->obs_cap_max);
    const size_t M = (size_t)c->p.map_capacity;
    bool ok = hipMalloc((void**)&c->d_kf_pose, 96 * (size_t)c->kf_cap) == hipSuccess && hipMalloc((void**)&c->d_pt_last, 4 * M) == hipSuccess &&
              hipMalloc((void**)&c->d_pt_first, 4 * M) == hipSuccess && hipMalloc((void**)&c->d_kf_reach, 8 * (size_t)c->kf_cap) == hipSuccess;
    ok = ok && obs_tables_alloc(cap, &c->d_obs_kf, &c->d_obs_mp, &c->d_obs_uv, &c->d_obs_alive, &c->d_obs_link) == VO_OK;
    if (!ok) {
        void* q[] = {c->d_kf_pose, c->d_pt_last, c->d_pt_first, c->d_kf_reach};
        for (void* x : q) if (x) (void)hipFree(x);
        c->d_kf_pose = nullptr; c->d_pt_last = nullptr; c->d_pt_first = nullptr; c->d_kf_reach = nullptr; (void)hipGetLastError();
        return VO_E_NOMEM;
    }
    HIP_TRY(hipMemsetAsync(c->d_pt_last, 0xFF, 4 * M, c->stream)); HIP_TRY(hipMemsetAsync(c->d_pt_first, 0xFF, 4 * M, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_kf_reach, 0x7F, 8 * (size_t)c->kf_cap, c->stream));      // 0x7F7F7F7F: larger than any index
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->obs_cap = cap;
    return VO_OK;
}
static int obs_tables_grow(vo_ctx* c, long long need) {
    long long cap = c->obs_cap;
    while (cap < need) cap = std::min(2 * cap, c->obs_cap_max);
    int32_t* kf; int32_t* mp; float* uv; uint8_t* alive; int* prev;
    int rc = obs_tables_alloc(cap, &kf, &mp, &uv, &alive, &prev);
    if (rc) return rc;
    const size_t n = (size_t)c->n_obs;
    bool ok = true;
    if (n) {
        ok = hipMemcpyAsync(kf, c->d_obs_kf, 4 * n, hipMemcpyDeviceToDevice, c->stream) == hipSuccess && hipMemcpyAsync(mp, c->d_obs_mp, 4 * n, hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
             hipMemcpyAsync(uv, c->d_obs_uv, 8 * n, hipMemcpyDeviceToDevice, c->stream) == hipSuccess && hipMemcpyAsync(alive, c->d_obs_alive, n, hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
             hipMemcpyAsync(prev, c->d_obs_link, 8 * n, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
    }
    ok = hipStreamSynchronize(c->stream) == hipSuccess && ok;
    if (!ok) {                                              // the fresh arrays go back; the tables stay as they were
        void* q[] = {kf, mp, uv, alive, prev};
        for (void* x : q) (void)hipFree(x);
        (void)hipGetLastError();
        return VO_E_DEVICE;
    }
    (void)hipFree(c->d_obs_kf); (void)hipFree(c->d_obs_mp); (void)hipFree(c->d_obs_uv); (void)hipFree(c->d_obs_alive); (void)hipFree(c->d_obs_link);
    c->d_obs_kf = kf; c->d_obs_mp = mp; c->d_obs_uv = uv; c->d_obs_alive = alive; c->d_obs_link = prev; c->obs_cap = cap;
    return VO_OK;
}
static int obs_room(vo_ctx* c, long long n) {               // make room for n more observations
    if (c->n_obs + n > c->obs_cap_max) return VO_E_OVERFLOW;
    if (c->n_obs + n > c->obs_cap) return obs_tables_grow(c, c->n_obs + n);
    return VO_OK;
}
static int kf_state(vo_ctx* c) {
    if (c->kf) return VO_OK;
    int rc = vo_obs_tables_ensure(c);
    if (rc) return rc;
    KfState* k = new (std::nothrow) KfState();
    if (!k) return VO_E_NOMEM;
    c->kf = k;
    const size_t M = (size_t)c->p.map_capacity;
    k->cand_cap = (int)std::min<size_t>(c->lane_stride, M);
    bool ok = hipHostMalloc((void**)&k->h, sizeof(KfHost), hipHostMallocDefault) == hipSuccess && hipMalloc((void**)&k->d_hdr, sizeof(KfDev)) == hipSuccess &&
              hipMalloc((void**)&k->d_w, 4 * (size_t)c->kf_cap) == hipSuccess && hipMalloc((void**)&k->d_mark, 4 * (size_t)c->kf_cap) == hipSuccess &&
              hipMalloc((void**)&k->d_key, 8 * M) == hipSuccess && hipMalloc((void**)&k->d_cand, 4 * (size_t)k->cand_cap) == hipSuccess &&
              hipMalloc((void**)&k->d_tri_ok, (size_t)k->cand_cap) == hipSuccess && hipMalloc((void**)&k->d_tri_xyz, 24 * (size_t)k->cand_cap) == hipSuccess &&
              hipMalloc((void**)&k->d_cnt, 8 * ((size_t)k->cand_cap / 256 + 2)) == hipSuccess && hipMalloc((void**)&k->d_kp_bits, 4 * ((size_t)c->p.n_features / 32 + 2)) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); kf_state_free(c); return VO_E_NOMEM; }
    memset(k->h, 0, sizeof(KfHost));
    HIP_TRY(hipMemsetAsync(k->d_w, 0, 4 * (size_t)c->kf_cap, c->stream)); HIP_TRY(hipMemsetAsync(k->d_mark, 0, 4 * (size_t)c->kf_cap, c->stream));
    HIP_TRY(hipMemsetAsync(k->d_key, 0, 8 * M, c->stream)); HIP_TRY(hipMemsetAsync(k->d_hdr, 0, sizeof(KfDev), c->stream));
    HIP_TRY(hipMemsetAsync(k->d_kp_bits, 0, 4 * ((size_t)c->p.n_features / 32 + 2), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VO_OK;
}
// the map's arrays grow (vo_map_grow): chain heads keep their contents and get empty tails, the leader keys keep theirs (they carry their epoch) and get
// zeroes, the candidate buffers are scratch
void vo_kf_map_regrow_records(vo_ctx* c, size_t m_old, size_t m_new, size_t stride_new, std::vector<MapRegrow>& v) {
    if (c->d_pt_last) {
        v.push_back(MapRegrow{(void**)&c->d_pt_last, 4 * m_old, 4 * m_new, 4 * m_old, 1, 0xFF, nullptr});
        v.push_back(MapRegrow{(void**)&c->d_pt_first, 4 * m_old, 4 * m_new, 4 * m_old, 1, 0xFF, nullptr});
    }
    if (c->kf) {
        KfState& K = *c->kf;
        const size_t cand_old = (size_t)K.cand_cap, cand_new = std::min(stride_new, m_new);
        v.push_back(MapRegrow{(void**)&K.d_key, 8 * m_old, 8 * m_new, 8 * m_old, 1, 0, nullptr});
        v.push_back(MapRegrow{(void**)&K.d_cand, 4 * cand_old, 4 * cand_new, 0, 1, -1, nullptr});
        v.push_back(MapRegrow{(void**)&K.d_tri_ok, cand_old, cand_new, 0, 1, -1, nullptr});
        v.push_back(MapRegrow{(void**)&K.d_tri_xyz, 24 * cand_old, 24 * cand_new, 0, 1, -1, nullptr});
        v.push_back(MapRegrow{(void**)&K.d_cnt, 8 * (cand_old / 256 + 2), 8 * (cand_new / 256 + 2), 0, 1, -1, nullptr});
    }
}
void vo_kf_map_regrown(vo_ctx* c, size_t m_new, size_t stride_new) {
    if (c->kf) c->kf->cand_cap = (int)std::min(stride_new, m_new);
}
static void note_kf_first(vo_ctx* c, int kf, long long at) {
    if (c->kf_first_obs.size() <= (size_t)kf) c->kf_first_obs.resize((size_t)kf + 1, -1);
    if (c->kf_first_obs[kf] < 0) c->kf_first_obs[kf] = at;
}
int vo_kf_host_pairs(vo_ctx* c, int** pair_a, int** pair_b, int* cap, int** n_total, double** poses) {
    int rc = kf_state(c);
    if (rc) return rc;
    *pair_a = c->kf->h->pair_a; *pair_b = c->kf->h->pair_b; *cap = KF_PAIR_CAP; *n_total = &c->kf->h->n_pairs_total; *poses = c->kf->h->poses;
    return VO_OK;
}

// vo_obs_append: the packed upload -> the observation table's columns, the points' chains and the keyframes' reach.  One launch per run of
// equal keyframe numbers: inside a run every map slot occurs once (a keyframe observes a point once), so the chain updates do not collide.
__global__ void k_obs_append(int n, int at, const int32_t* __restrict__ kf, const int32_t* __restrict__ mp, const float2* __restrict__ uv,
                             int32_t* __restrict__ o_kf, int32_t* __restrict__ o_mp, float2* __restrict__ o_uv, uint8_t* __restrict__ o_alive, int2* __restrict__ o_link,
                             int32_t* __restrict__ pt_last, int32_t* __restrict__ pt_first, int2* __restrict__ kf_reach) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int o = at + i, slot = mp[i], k = kf[i];
    o_kf[o] = k; o_mp[o] = slot; o_uv[o] = uv[i]; o_alive[o] = 1;
    o_link[o] = make_int2(pt_last[slot], k); pt_last[slot] = o;
    int pf = pt_first[slot];
    if (pf < 0) { pf = o; pt_first[slot] = o; }
    atomicMin(&kf_reach[k].x, pf); atomicMin(&kf_reach[k].y, slot);
}

extern "C" int vo_kf_set_pose(vo_ctx* c, const int32_t* kf, const double* T, int n) {
    if (!c || n < 0 || (n && (!kf || !T))) return VO_E_INVALID;
    if (n == 0) return VO_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = vo_obs_tables_ensure(c);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) if (kf[i] < 0 || kf[i] >= c->kf_cap) return VO_E_INVALID;
    double* h = (double*)vo_stage(c, 96 * (size_t)n);
    if (!h) return VO_E_NOMEM;
    HIP_TRY(hipStreamSynchronize(c->stream));               // the staging buffer may still feed an earlier copy
    memcpy(h, T, 96 * (size_t)n);
    int run0 = 0;                                           // consecutive keyframe numbers travel as one copy
    for (int i = 1; i <= n; ++i)
        if (i == n || kf[i] != kf[i - 1] + 1) {
            HIP_TRY(hipMemcpyAsync(c->d_kf_pose + 12 * (size_t)kf[run0], h + 12 * (size_t)run0, 96 * (size_t)(i - run0), hipMemcpyHostToDevice, c->stream));
            run0 = i;
        }
    for (int i = 0; i < n; ++i) c->n_kf = std::max(c->n_kf, kf[i] + 1);
    HIP_TRY(hipStreamSynchronize(c->stream));               // a back-end thread may read the table from its own stream next
    return VO_OK;
}

extern "C" int vo_obs_append(vo_ctx* c, const int32_t* kf, const int32_t* mp, const float* uv, int n, int64_t* first) {
    if (!c || n < 0 || (n && (!kf || !mp || !uv))) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    int rc = vo_obs_tables_ensure(c);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) if (kf[i] < 0 || kf[i] >= c->kf_cap || mp[i] < 0 || mp[i] >= c->p.map_capacity) return VO_E_INVALID;
    if ((rc = obs_room(c, n))) return rc;
    if (first) *first = (int64_t)c->n_obs;
    if (n == 0) return VO_OK;
    // pack -> one pinned staging buffer -> one H2D copy into the scratch slab -> kernels write the table columns and the chains
    const size_t N = (size_t)n, o_mp = (4 * N + 255) & ~(size_t)255, o_uv = o_mp + ((4 * N + 255) & ~(size_t)255), total = o_uv + 8 * N;
    uint8_t* h = (uint8_t*)vo_stage(c, total);
    if (!h) return VO_E_NOMEM;
    rc = vo_scratch(c, total);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));               // the staging buffer / the scratch slab may still feed an earlier copy or scatter
    memcpy(h, kf, 4 * N); memcpy(h + o_mp, mp, 4 * N); memcpy(h + o_uv, uv, 8 * N);
    const int at = (int)c->n_obs;
    uint8_t* d = (uint8_t*)c->d_ba;
    HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, c->stream));
    std::vector<int> kfs;                                   // distinct keyframes of this call
    for (int r0 = 0; r0 < n;) {
        int r1 = r0 + 1;
        while (r1 < n && kf[r1] == kf[r0]) ++r1;
        hipLaunchKernelGGL(k_obs_append, dim3((r1 - r0 + 255) / 256), dim3(256), 0, c->stream, r1 - r0, at + r0, (const int32_t*)d + r0, (const int32_t*)(d + o_mp) + r0,
                           (const float2*)(d + o_uv) + r0, c->d_obs_kf, c->d_obs_mp, reinterpret_cast<float2*>(c->d_obs_uv), c->d_obs_alive, reinterpret_cast<int2*>(c->d_obs_link), c->d_pt_last, c->d_pt_first,
                           reinterpret_cast<int2*>(c->d_kf_reach));
        if (std::find(kfs.begin(), kfs.end(), kf[r0]) == kfs.end()) kfs.push_back(kf[r0]);
        note_kf_first(c, kf[r0], (long long)at + r0);
        r0 = r1;
    }
    HIP_TRY(hipGetLastError());
    // where this call's keyframes reach back to in the tables (the resident graph cut enters them there: vo_ba.hip, CutTabs)
    if (c->kf_reach.size() < (size_t)c->kf_cap) c->kf_reach.resize((size_t)c->kf_cap);
    int2* hr = (int2*)h;                                    // (the staging buffer has been consumed by the copy above once the stream is idle)
    HIP_TRY(hipStreamSynchronize(c->stream));               // a back-end thread may read the table from its own stream next
    for (size_t i = 0; i < kfs.size(); ++i) HIP_TRY(hipMemcpyAsync(hr + i, reinterpret_cast<int2*>(c->d_kf_reach) + kfs[i], 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < kfs.size(); ++i) { c->kf_reach[kfs[i]].obs_lo = hr[i].x; c->kf_reach[kfs[i]].slot_lo = hr[i].y; }
    c->n_obs += n;
    return VO_OK;
}

extern "C" int vo_obs_kill(vo_ctx* c, const int64_t* ids, int n) {
    if (!c || n < 0 || (n && !ids)) return VO_E_INVALID;
    if (n == 0) return VO_OK;
    HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= c->n_obs) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) HIP_TRY(hipMemsetAsync(c->d_obs_alive + (size_t)ids[i], 0, 1, c->stream));     // a handful per local BA
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VO_OK;
}

// ---- keyframe commit ---------------------------------------------------------------------------------------------------------------
struct KfTabs { int32_t* obs_kf; int32_t* obs_mp; float2* obs_uv; uint8_t* obs_alive; int2* obs_link; int32_t* pt_last; int32_t* pt_first;
                double* map_pos; double* map_nrm; uint32_t* map_desc; uint8_t* map_flags; double* kf_pose; };
struct Pose12 { double v[12]; };

static KfTabs tabs_of(vo_ctx* c) {
    return KfTabs{c->d_obs_kf, c->d_obs_mp, reinterpret_cast<float2*>(c->d_obs_uv), c->d_obs_alive, reinterpret_cast<int2*>(c->d_obs_link), c->d_pt_last, c->d_pt_first,
                  c->d_map_pos, c->d_map_nrm, c->d_map_desc, c->d_map_flags, c->d_kf_pose};
}

// position of a set flag among the set flags of the workgroup's 1024 lanes (lane order), and their number
__device__ __forceinline__ int block_rank(bool f, int& total, int* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(f);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int before = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int v = s_w[w]; if (w < wave) before += v; tot += v; }
    __syncthreads();                                        // s_w is rewritten by the next call
    total = tot;
    return before;
}

// The commit proper, in two launches over as many workgroups as the lists need (a single workgroup walked the 3 000 matches and 2 000 keypoints
// of the bench workload in 55-85 us; a lane per record takes one trip through the dependent gathers):
//   k_kf_count   per 256 matches: how many are LM inliers / candidates of the triangulation loop; the keypoints the inliers explain are marked in a
//                bitmap in global memory (pnpMatchedKptSet_)
//   k_kf_place   match blocks: a record's place in the observation table = inliers in the blocks in front (sum of their counts) + rank inside the
//                block (ballot): the order of the host loop (src/frontend.cpp:366-370); keypoint blocks: the same for the new map points
//                (src/frontend.cpp:372-406) -- a block recounts the keypoints in front of it from the bitmap and the depth samples (<= 31 per lane)
__device__ __forceinline__ int kf_block_sum(int v, int* s_w) {      // sum over a 256-lane workgroup, returned to every lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    const int t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(256) void k_kf_count(KfTabs T, const vo_match* __restrict__ matches, int n_match, int2* __restrict__ cnt, unsigned* __restrict__ kp_bits, KfDev* __restrict__ hdr) {
    __shared__ int s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool inl = false, cnd = false;
    if (i < n_match) {
        const int4 m = *reinterpret_cast<const int4*>(matches + i);      // map_index, kp_index, distance, flags
        inl = (m.w & VO_MATCH_LM_INLIER) != 0;
        if (inl) {
            atomicOr(&kp_bits[m.y >> 5], 1u << (m.y & 31));
            cnd = !(T.map_flags[m.x] & (VO_MAP_FLAG_OUTLIER | VO_MAP_FLAG_TRIANGULATED | VO_MAP_FLAG_OPTIMIZED));
        }
    }
    const int a = kf_block_sum(inl ? 1 : 0, s_w), c = kf_block_sum(cnd ? 1 : 0, s_w);
    if (threadIdx.x == 0) {
        cnt[blockIdx.x] = make_int2(a, c);
        if (blockIdx.x == 0) { hdr->reach_obs = INT_MAX; hdr->reach_slot = INT_MAX; }
    }
}
__global__ __launch_bounds__(256) void k_kf_place(KfTabs T, const vo_match* __restrict__ matches, int n_match, int nb, const int2* __restrict__ cnt, const unsigned* __restrict__ kp_bits,
                                                  const vo_keypoint* __restrict__ kps, const int* __restrict__ nkp_p, int nfeat, const uint32_t* __restrict__ fdesc, int kf, int n_obs0,
                                                  int first_new, Pose12 P, CamD cam, double depth_scale, int32_t* __restrict__ cand, KfDev* __restrict__ hdr) {
    __shared__ int s_w[4], s_min[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const double* R = P.v; const double* t = P.v + 9;
    // camera centre = translation of T^-1 = (R^T t) * -1 (SE3::inverse of the host layer, operation for operation)
    double C[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) C[a] = (R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]) * -1.0;
    // inliers / candidates in the match blocks in front of this one (every block: the keypoint blocks need the totals)
    int fi = 0, fc = 0, ti = 0, tc = 0;
    for (int j = tid; j < nb; j += 256) { const int2 v = cnt[j]; ti += v.x; tc += v.y; if (j < b) { fi += v.x; fc += v.y; } }
    fi = kf_block_sum(fi, s_w); fc = kf_block_sum(fc, s_w); ti = kf_block_sum(ti, s_w); tc = kf_block_sum(tc, s_w);
    if (b < nb) {                                            // AddCurrentKeyframeObservations (src/frontend.cpp:366-370), match order
        if (tid < 2) s_min[tid] = INT_MAX;
        const int i = b * 256 + tid;
        int4 m = make_int4(0, 0, 0, 0);
        if (i < n_match) m = *reinterpret_cast<const int4*>(matches + i);
        const bool inl = i < n_match && (m.w & VO_MATCH_LM_INLIER);
        const bool cnd = inl && !(T.map_flags[m.x] & (VO_MAP_FLAG_OUTLIER | VO_MAP_FLAG_TRIANGULATED | VO_MAP_FLAG_OPTIMIZED));
        const unsigned long long mi = __ballot(inl), mc = __ballot(cnd);
        if (lane == 0) { s_w[wave] = __popcll(mi); }
        __syncthreads();
        int ri = fi + __popcll(mi & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) ri += s_w[w];
        __syncthreads();
        if (lane == 0) { s_w[wave] = __popcll(mc); }
        __syncthreads();
        int rc = fc + __popcll(mc & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) rc += s_w[w];
        if (inl) {
            const int o = n_obs0 + ri, sl = m.x;
            const float2 xy = *reinterpret_cast<const float2*>(kps + m.y);      // x, y lead the record
            T.obs_kf[o] = kf; T.obs_mp[o] = sl; T.obs_uv[o] = xy; T.obs_alive[o] = 1;
            T.obs_link[o] = make_int2(T.pt_last[sl], kf); T.pt_last[sl] = o;
            int f = T.pt_first[sl];
            if (f < 0) { f = o; T.pt_first[sl] = o; }
            atomicMin(&s_min[0], f); atomicMin(&s_min[1], sl);
            // Mappoint::AddObservedByKeyframe (src/mappoint.cpp:30-38): norm = (norm + (pos - centre).normalized()).normalized()
            const double* p = T.map_pos + 3 * (size_t)sl; double* nr = T.map_nrm + 3 * (size_t)sl;
            double d0 = p[0] - C[0], d1 = p[1] - C[1], d2 = p[2] - C[2];
            double n = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            d0 = d0 / n; d1 = d1 / n; d2 = d2 / n;
            double e0 = nr[0] + d0, e1 = nr[1] + d1, e2 = nr[2] + d2;
            n = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
            nr[0] = e0 / n; nr[1] = e1 / n; nr[2] = e2 / n;
            if (cnd) cand[rc] = sl;
        }
        __syncthreads();
        if (tid == 0 && s_min[0] != INT_MAX) { atomicMin(&hdr->reach_obs, s_min[0]); atomicMin(&hdr->reach_slot, s_min[1]); }
        return;
    }
    // CreateNewMappoints (src/frontend.cpp:372-406), keypoint order
    const int jb = b - nb, nkp = min(*nkp_p, nfeat), i = jb * 256 + tid;
    if (jb == 0 && tid < 12) T.kf_pose[12 * (size_t)kf + tid] = P.v[tid];      // the keyframe's pose (vo_kf_set_pose)
    int front = 0;                                           // new points among the keypoints in front of this block
    for (int q = tid; q < jb * 256; q += 256) front += (q < nkp && !((kp_bits[q >> 5] >> (q & 31)) & 1u) && kps[q].depth_raw != 0) ? 1 : 0;
    front = kf_block_sum(front, s_w);
    vo_keypoint k; k.depth_raw = 0; k.x = 0; k.y = 0;
    if (i < nkp) k = kps[i];
    const bool f = i < nkp && !((kp_bits[i >> 5] >> (i & 31)) & 1u) && k.depth_raw != 0;
    const unsigned long long mf = __ballot(f);
    if (lane == 0) s_w[wave] = __popcll(mf);
    __syncthreads();
    int r = front + __popcll(mf & ((1ull << lane) - 1ull)), own = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) r += s_w[w]; own += s_w[w]; }
    if (f) {
        const int slot = first_new + r, o = n_obs0 + ti + r;
        const double depth = double(k.depth_raw) / depth_scale;                       // Frame::GetDepth (src/frame.cpp:43-67)
        const double pc0 = ((double)k.x - cam.cx) * depth / cam.fx, pc1 = ((double)k.y - cam.cy) * depth / cam.fy, pc2 = depth;      // Camera::Pixel2Camera
        double pw[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) pw[a] = (R[a] * pc0 + R[3 + a] * pc1 + R[6 + a] * pc2) + C[a];      // T^-1 * p_c = R^T p_c + t'
        double d0 = pw[0] - C[0], d1 = pw[1] - C[1], d2 = pw[2] - C[2];
        double n = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        d0 = d0 / n; d1 = d1 / n; d2 = d2 / n;
        n = sqrt(d0 * d0 + d1 * d1 + d2 * d2);            // (0 + d).normalized() of the first observation
        double* mp = T.map_pos + 3 * (size_t)slot; double* nr = T.map_nrm + 3 * (size_t)slot;
        mp[0] = pw[0]; mp[1] = pw[1]; mp[2] = pw[2]; nr[0] = d0 / n; nr[1] = d1 / n; nr[2] = d2 / n;
        const uint4* src = reinterpret_cast<const uint4*>(fdesc + 8 * (size_t)i); uint4* dst = reinterpret_cast<uint4*>(T.map_desc + 8 * (size_t)slot);
        dst[0] = src[0]; dst[1] = src[1];
        T.map_flags[slot] = 0;
        T.obs_kf[o] = kf; T.obs_mp[o] = slot; T.obs_uv[o] = make_float2(k.x, k.y); T.obs_alive[o] = 1; T.obs_link[o] = make_int2(-1, kf);
        T.pt_last[slot] = o; T.pt_first[slot] = o;
    }
    if (tid == 0) {
        if (jb == 0) { hdr->n_matched = ti; hdr->n_tri = tc; }
        if (b == (int)gridDim.x - 1) {                       // the last keypoint block knows the number of new points
            hdr->n_new = front + own;
            if (front + own > 0) { atomicMin(&hdr->reach_obs, n_obs0 + ti); atomicMin(&hdr->reach_slot, first_new); }
        }
    }
}

// Two jobs in one launch (they are independent; side by side they cost the longer one).
// Blocks [0, nb_covis): src/frame.cpp:104-119 -- every keyframe that already sees a point the new keyframe observes gains one shared point with it:
//   one lane per new observation walks the point's chain; the counts are gathered in an LDS table over the newest KF_WLDS keyframe numbers
//   (thousands of global atomics on a few dozen addresses serialise: DESIGN 4a) and leave with one atomic per keyframe and workgroup.
// Blocks [nb_covis, ..): src/frontend.cpp:465-506, one lane per candidate: the live observations along the point's chain (newest first) are taken
//   oldest first -- the order of Mappoint's observation list -- by walking the chain once per view (2-3 views: no BA has touched these points yet).
#define KF_WLDS 2048
__global__ __launch_bounds__(256) void k_kf_covis_tri(KfTabs T, const KfDev* __restrict__ hdr, int n_obs0, int n_kf, int32_t* __restrict__ w, int nb_covis,
                                                      const int32_t* __restrict__ cand, CamD cam, uint8_t* __restrict__ tri_ok, double* __restrict__ tri_xyz) {
    __shared__ int s_cnt[KF_WLDS];
    if ((int)blockIdx.x < nb_covis) {
        const int j = blockIdx.x * 256 + threadIdx.x, k_lo = max(0, n_kf - KF_WLDS);
        if (blockIdx.x * 256 >= hdr->n_matched) return;
        for (int i = threadIdx.x; i < KF_WLDS; i += 256) s_cnt[i] = 0;
        __syncthreads();
        if (j < hdr->n_matched) {
            int q = T.obs_link[n_obs0 + j].x;
            while (q >= 0) {
                const int2 l = T.obs_link[q];
                if (T.obs_alive[q]) { if (l.y >= k_lo) atomicAdd(&s_cnt[l.y - k_lo], 1); else atomicAdd(&w[l.y], 1); }
                q = l.x;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < KF_WLDS; i += 256) { const int v = s_cnt[i]; if (v) atomicAdd(&w[k_lo + i], v); }
        return;
    }
    const int r = (blockIdx.x - nb_covis) * 256 + threadIdx.x;
    if (r >= hdr->n_tri) return;
    const int slot = cand[r];
    int n = 0;
    for (int q = T.pt_last[slot]; q >= 0; q = T.obs_link[q].x) n += T.obs_alive[q] ? 1 : 0;
    uint8_t good = 0;
    if (n >= 2) {
        double a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = 0.0;
        for (int v = 0; v < n; ++v) {                        // view v in ascending order = live link number n - 1 - v from the head
            int skip = n - 1 - v, q = T.pt_last[slot];
            for (;;) { if (T.obs_alive[q]) { if (skip == 0) break; --skip; } q = T.obs_link[q].x; }
            const double* Pq = T.kf_pose + 12 * (size_t)T.obs_link[q].y;
            double p[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) p[k] = Pq[k];
            const float2 uv = T.obs_uv[q];
            tri_accumulate(a, p, ((double)uv.x - cam.cx) * 1.0 / cam.fx, ((double)uv.y - cam.cy) * 1.0 / cam.fy);      // Camera::Pixel2Camera, depth 1
        }
        double x[3];
        if (tri_solve(a, x) && x[2] > 0) { good = 1; tri_xyz[3 * (size_t)r] = x[0]; tri_xyz[3 * (size_t)r + 1] = x[1]; tri_xyz[3 * (size_t)r + 2] = x[2]; }
    }
    tri_ok[r] = good;
}

// one workgroup: weights -> ascending (keyframe, weight) list in pinned memory (the counters are left at zero for the next keyframe); the first
// successful triangulation moves its point (src/frontend.cpp:496-501); counts and reach to the host
__global__ __launch_bounds__(1024) void k_kf_finish(KfTabs T, const KfDev* __restrict__ hdr, int n_kf, int kf, int n_obs0, int32_t* __restrict__ w, const int32_t* __restrict__ cand,
                                                    const uint8_t* __restrict__ tri_ok, const double* __restrict__ tri_xyz, int2* __restrict__ kf_reach, KfHost* __restrict__ h,
                                                    unsigned* __restrict__ kp_bits, int nfeat) {
    __shared__ int s_w[16], s_pick;
    const int tid = threadIdx.x;
    if (tid == 0) s_pick = INT_MAX;
    for (int i = tid; i < (nfeat + 31) / 32; i += 1024) kp_bits[i] = 0u;      // the keypoint bitmap is left empty for the next keyframe
    int base = 0;
    for (int k0 = 0; k0 < n_kf; k0 += 1024) {
        const int k = k0 + tid;
        const int v = k < n_kf ? w[k] : 0;
        int tot;
        const int r = block_rank(v > 0, tot, s_w);
        if (v > 0) { w[k] = 0; if (base + r < KF_COVIS_CAP) { h->covis_kf[base + r] = k; h->covis_w[base + r] = v; } }
        base += tot;
    }
    const int nt = hdr->n_tri;
    int best = INT_MAX;
    for (int r = tid; r < nt; r += 1024) if (tri_ok[r]) { best = r; break; }
    __syncthreads();
    if (best != INT_MAX) atomicMin(&s_pick, best);
    __syncthreads();
    if (tid == 0) {
        int slot = -1;
        if (s_pick != INT_MAX) {
            slot = cand[s_pick];
            T.map_pos[3 * (size_t)slot] = tri_xyz[3 * (size_t)s_pick]; T.map_pos[3 * (size_t)slot + 1] = tri_xyz[3 * (size_t)s_pick + 1]; T.map_pos[3 * (size_t)slot + 2] = tri_xyz[3 * (size_t)s_pick + 2];
            T.map_flags[slot] |= VO_MAP_FLAG_TRIANGULATED;
        }
        kf_reach[kf] = make_int2(hdr->reach_obs, hdr->reach_slot);
        h->r.n_matched = hdr->n_matched; h->r.n_new = hdr->n_new; h->r.first_obs = n_obs0; h->r.n_covisible = min(base, KF_COVIS_CAP);
        // the reference's loop stops looking at the first success: candidates behind it are not counted
        h->r.n_tri_candidates = s_pick != INT_MAX ? s_pick + 1 : nt; h->r.triangulated_slot = slot; h->r.n_covisible_total = base;
        h->reach_obs = hdr->reach_obs; h->reach_slot = hdr->reach_slot; h->n_covis_total = base;
    }
}

extern "C" int vo_keyframe_commit(vo_ctx* c, int lane, int frame_slot, int32_t kf, const double T_cw[12], int32_t first_new_slot,
                                  int32_t* covis_kf, int32_t* covis_weight, int cap_covis, vo_kf_commit_result* out) {
    if (!c || !T_cw || !out || frame_slot < 0 || frame_slot >= c->p.max_frames || first_new_slot < 0 || cap_covis < 0 || (cap_covis && (!covis_kf || !covis_weight))) return VO_E_INVALID;
    if (lane >= c->last_track_lanes || lane < -1) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;
    if (!c->slot_orb[frame_slot]) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    int rc = kf_state(c);
    if (rc) return rc;
    if (kf < 0 || kf >= c->kf_cap) return VO_E_INVALID;
    const int n_match = lane >= 0 ? std::min(c->h_track[lane].n_match, (int)c->lane_stride) : 0;
    const int nfeat = c->p.n_features;
    if ((long long)first_new_slot + nfeat > c->p.map_capacity) {      // every keypoint may become a map point: the map grows (vo_map_grow; VO_E_OVERFLOW beyond its ceiling)
        const int grc = vo_map_grow(c, (long long)first_new_slot + nfeat);
        if (grc) return grc;
    }
    if (n_match > c->kf->cand_cap) return VO_E_OVERFLOW;
    if ((rc = obs_room(c, (long long)n_match + nfeat))) return rc;
    if (c->n_obs + n_match + nfeat >= INT_MAX) return VO_E_OVERFLOW;
    KfState& K = *c->kf;
    hipStream_t st = c->stream;
    const KfTabs T = tabs_of(c);
    Pose12 P; memcpy(P.v, T_cw, 96);
    const CamD cam{(double)c->p.fx, (double)c->p.fy, (double)c->p.cx, (double)c->p.cy, c->p.width, c->p.height};
    const int n_obs0 = (int)c->n_obs, n_kf = std::max(c->n_kf, kf + 1);
    const vo_match* dm = c->d_matches + (size_t)std::max(lane, 0) * c->lane_stride;
    const int nbm = (n_match + 255) / 256, nbk = (nfeat + 255) / 256;
    { ProfScope ps(c, "k_kf_commit");                       // (two launches under one name: count, then place)
      if (nbm > 0) hipLaunchKernelGGL(k_kf_count, dim3(nbm), dim3(256), 0, st, T, dm, n_match, K.d_cnt, K.d_kp_bits, K.d_hdr);
      else HIP_TRY(hipMemsetAsync(K.d_hdr, 0x7F, sizeof(KfDev), st));      // no matches (the first keyframe): nothing counts, the reach starts at "none"
      hipLaunchKernelGGL(k_kf_place, dim3(nbm + nbk), dim3(256), 0, st, T, dm, n_match, nbm, (const int2*)K.d_cnt, (const unsigned*)K.d_kp_bits, (const vo_keypoint*)(c->d_kps + (size_t)frame_slot * nfeat),
                         (const int*)(c->d_nkp + frame_slot), nfeat, (const uint32_t*)(c->d_desc + (size_t)frame_slot * nfeat * 32), (int)kf, n_obs0, (int)first_new_slot, P, cam,
                         (double)c->p.depth_scale, K.d_cand, K.d_hdr); }
    if (n_match > 0) {
        ProfScope ps(c, "k_kf_covis_tri");
        const int nb = (n_match + 255) / 256;
        hipLaunchKernelGGL(k_kf_covis_tri, dim3(2 * nb), dim3(256), 0, st, T, (const KfDev*)K.d_hdr, n_obs0, n_kf, K.d_w, nb, (const int32_t*)K.d_cand, cam, K.d_tri_ok, K.d_tri_xyz);
    }
    { ProfScope ps(c, "k_kf_finish");
      hipLaunchKernelGGL(k_kf_finish, dim3(1), dim3(1024), 0, st, T, (const KfDev*)K.d_hdr, n_kf, (int)kf, n_obs0, K.d_w, (const int32_t*)K.d_cand, (const uint8_t*)K.d_tri_ok,
                         (const double*)K.d_tri_xyz, reinterpret_cast<int2*>(c->d_kf_reach), K.h, K.d_kp_bits, nfeat); }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));                      // the pinned block is complete; a back-end thread may read the tables from its own stream next
    const KfHost& H = *K.h;
    *out = H.r;
    c->n_obs += H.r.n_matched + H.r.n_new;
    c->n_kf = n_kf;
    c->map_hi = std::max(c->map_hi, first_new_slot + H.r.n_new);
    if (c->kf_reach.size() < (size_t)c->kf_cap) c->kf_reach.resize((size_t)c->kf_cap);
    if (H.r.n_matched + H.r.n_new > 0) { c->kf_reach[kf].obs_lo = H.reach_obs; c->kf_reach[kf].slot_lo = H.reach_slot; note_kf_first(c, kf, n_obs0); }
    const int take = std::min(H.r.n_covisible, cap_covis);
    if (take > 0) { memcpy(covis_kf, H.covis_kf, 4 * (size_t)take); memcpy(covis_weight, H.covis_w, 4 * (size_t)take); }
    out->n_covisible = take;
    out->n_covisible_total = H.n_covis_total;              // more than the caller (or the pinned block) holds: vo_kf_covisibility reads them all (nothing is lost on the device)
    return VO_OK;
}

// parity tap: the weights of one keyframe recounted from the tables (a scan of the whole table: not on the product path)
__global__ void k_kf_recount(KfTabs T, int n_obs, int kf, int32_t* __restrict__ w) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obs || !T.obs_alive[o] || T.obs_kf[o] != kf) return;
    for (int q = T.pt_last[T.obs_mp[o]]; q >= 0; q = T.obs_link[q].x) if (q != o && T.obs_alive[q]) atomicAdd(&w[T.obs_link[q].y], 1);
}
extern "C" int vo_kf_covisibility(vo_ctx* c, int32_t kf, int32_t* covis_kf, int32_t* covis_weight, int cap, int32_t* n) {
    if (!c || kf < 0 || !n || cap < 0 || (cap && (!covis_kf || !covis_weight))) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    int rc = kf_state(c);
    if (rc) return rc;
    *n = 0;
    if (c->n_obs == 0 || c->n_kf == 0) return VO_OK;
    KfState& K = *c->kf;
    const KfTabs T = tabs_of(c);
    hipLaunchKernelGGL(k_kf_recount, dim3((int)((c->n_obs + 255) / 256)), dim3(256), 0, c->stream, T, (int)c->n_obs, (int)kf, K.d_w);
    std::vector<int32_t> w((size_t)c->n_kf);
    HIP_TRY(hipMemcpyAsync(w.data(), K.d_w, 4 * w.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(K.d_w, 0, 4 * w.size(), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int k = 0;
    for (size_t q = 0; q < w.size(); ++q) if (w[q] > 0) { if (k < cap) { covis_kf[k] = (int32_t)q; covis_weight[k] = w[q]; } ++k; }
    *n = k;
    return k > cap ? VO_E_OVERFLOW : VO_OK;
}

// ---- local map (src/mapmanager.cpp:14-38, src/frontend.cpp:159-166) -----------------------------------------------------------------
// Window = the table from the first observation of the oldest listed keyframe.  A point's place in the list is that of its first live
// observation by a listed keyframe (keyframes ascending, observation order inside a keyframe): k_act_lead elects it with one 64-bit
// atomicMax per observation (epoch in the high word: no clearing between calls), k_act_flag marks the elected observations, a scan
// numbers them, k_act_emit writes the slots.
__global__ void k_act_mark(const int* __restrict__ list, int n, int32_t* __restrict__ mark, int epoch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mark[list[i]] = epoch;
}
__device__ __forceinline__ bool act_hit(const KfTabs& T, int o, const int32_t* __restrict__ mark, int epoch, int n_kf) {
    if (!T.obs_alive[o]) return false;
    const int k = T.obs_kf[o];
    return k < n_kf && mark[k] == epoch && !(T.map_flags[T.obs_mp[o]] & VO_MAP_FLAG_OUTLIER);
}
__global__ void k_act_lead(KfTabs T, int lo, int n_obs, const int32_t* __restrict__ mark, int epoch, int n_kf, unsigned long long* __restrict__ key) {
    const int o = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obs || !act_hit(T, o, mark, epoch, n_kf)) return;
    atomicMax(&key[T.obs_mp[o]], ((unsigned long long)(unsigned)epoch << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)o));
}
__global__ void k_act_flag(KfTabs T, int lo, int n_obs, int n_pad, const int32_t* __restrict__ mark, int epoch, int n_kf, const unsigned long long* __restrict__ key, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, o = lo + i;
    if (i >= n_pad) return;
    int f = 0;
    if (o < n_obs && act_hit(T, o, mark, epoch, n_kf))
        f = key[T.obs_mp[o]] == (((unsigned long long)(unsigned)epoch << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)o)) ? 1 : 0;
    flag[i] = f;
}
__global__ void k_act_emit(KfTabs T, int lo, int n_win, const int* __restrict__ flag, const int* __restrict__ pos, const int* __restrict__ total, int32_t* __restrict__ active, int cap, KfHost* __restrict__ h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) h->n_active = *total;
    if (i < n_win && flag[i] && pos[i] < cap) active[pos[i]] = T.obs_mp[lo + i];
}
__global__ void k_act_iota(int n, int32_t* __restrict__ active) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) active[i] = i;
}

extern "C" int vo_map_set_active_covisible(vo_ctx* c, const int32_t* kf, int n, int min_points, int32_t n_map_points, int32_t* n_active) {
    if (!c || n < 0 || (n && !kf) || n > KF_LIST_CAP || n_map_points < 0 || n_map_points > c->p.map_capacity || n_map_points > c->active_cap) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    int rc = kf_state(c);
    if (rc) return rc;
    KfState& K = *c->kf;
    hipStream_t st = c->stream;
    int found = 0;
    long long lo = c->n_obs;
    for (int i = 0; i < n; ++i) {
        if (kf[i] < 0 || kf[i] >= c->kf_cap) return VO_E_INVALID;
        if ((size_t)kf[i] < c->kf_first_obs.size() && c->kf_first_obs[kf[i]] >= 0) lo = std::min(lo, c->kf_first_obs[kf[i]]);
    }
    lo &= ~3ll;
    const int n_win = (int)(c->n_obs - lo);
    if (n > 0 && n_win > 0) {
        const int n_pad = (n_win + 3) & ~3;
        const size_t o_pos = ((size_t)4 * n_pad + 255) & ~(size_t)255, total = 2 * o_pos;
        if (!K.d_scan) {
            if (hipMalloc(&K.d_scan, 4096 + 256) != hipSuccess) { (void)hipGetLastError(); K.d_scan = nullptr; return VO_E_NOMEM; }
            HIP_TRY(hipMemsetAsync(K.d_scan, 0, 4096 + 256, st));
        }
        if (total > K.act_bytes) {
            if (K.d_act) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(K.d_act); }
            K.d_act = nullptr; K.act_bytes = 0;
            if (hipMalloc(&K.d_act, total + total / 2) != hipSuccess) { (void)hipGetLastError(); return VO_E_NOMEM; }
            K.act_bytes = total + total / 2;
        }
        int* flag = (int*)K.d_act; int* pos = (int*)((uint8_t*)K.d_act + o_pos); int* bsum = (int*)K.d_scan; int* tot = bsum + 1024;
        const int epoch = (int)(++K.epoch & 0x7FFFFFFF);
        memcpy(K.h->kf_list, kf, 4 * (size_t)n);            // (the previous call's kernels have been waited for)
        const KfTabs T = tabs_of(c);
        const int g = (n_pad + 255) / 256;
        { ProfScope ps(c, "k_act_lead");
          hipLaunchKernelGGL(k_act_mark, dim3((n + 255) / 256), dim3(256), 0, st, (const int*)K.h->kf_list, n, K.d_mark, epoch);
          hipLaunchKernelGGL(k_act_lead, dim3(g), dim3(256), 0, st, T, (int)lo, (int)c->n_obs, (const int32_t*)K.d_mark, epoch, c->n_kf, K.d_key); }
        { ProfScope ps(c, "k_act_emit");
          hipLaunchKernelGGL(k_act_flag, dim3(g), dim3(256), 0, st, T, (int)lo, (int)c->n_obs, n_pad, (const int32_t*)K.d_mark, epoch, c->n_kf, (const unsigned long long*)K.d_key, flag);
          if ((rc = vo_scan_i32(st, flag, n_pad, bsum, pos, tot))) return rc;
          hipLaunchKernelGGL(k_act_emit, dim3(g), dim3(256), 0, st, T, (int)lo, n_win, (const int*)flag, (const int*)pos, (const int*)tot, c->d_active, c->active_cap, K.h); }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        found = std::min(K.h->n_active, c->active_cap);
    }
    if (found < min_points) {                               // src/frontend.cpp:163-166: the whole map
        if (n_map_points > 0) hipLaunchKernelGGL(k_act_iota, dim3((n_map_points + 255) / 256), dim3(256), 0, st, (int)n_map_points, c->d_active);
        HIP_TRY(hipGetLastError());
        found = n_map_points;
    }
    c->n_active = found;
    if (n_active) *n_active = found;
    return VO_OK;
}

extern "C" int vo_tables_fetch(vo_ctx* c, int64_t obs0, int64_t obs_cap, int32_t* obs_kf, int32_t* obs_mp, float* obs_uv, uint8_t* obs_alive, int64_t* n_obs,
                               int32_t map0, int32_t map_cap, double* map_xyz, double* map_normal, uint8_t* map_desc, uint8_t* map_flags,
                               int32_t* active, int active_cap, int32_t* n_active) {
    if (!c || obs0 < 0 || obs_cap < 0 || map0 < 0 || map_cap < 0 || map0 + (int64_t)map_cap > c->p.map_capacity || active_cap < 0) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_obs) *n_obs = c->n_obs;
    const int64_t take = std::max<int64_t>(0, std::min<int64_t>(obs_cap, c->n_obs - obs0));
    if (take > 0 && c->d_obs_kf) {
        if (obs_kf) HIP_TRY(hipMemcpy(obs_kf, c->d_obs_kf + obs0, 4 * (size_t)take, hipMemcpyDeviceToHost));
        if (obs_mp) HIP_TRY(hipMemcpy(obs_mp, c->d_obs_mp + obs0, 4 * (size_t)take, hipMemcpyDeviceToHost));
        if (obs_uv) HIP_TRY(hipMemcpy(obs_uv, c->d_obs_uv + 2 * obs0, 8 * (size_t)take, hipMemcpyDeviceToHost));
        if (obs_alive) HIP_TRY(hipMemcpy(obs_alive, c->d_obs_alive + obs0, (size_t)take, hipMemcpyDeviceToHost));
    }
    if (map_cap > 0) {
        if (map_xyz) HIP_TRY(hipMemcpy(map_xyz, c->d_map_pos + 3 * (size_t)map0, 24 * (size_t)map_cap, hipMemcpyDeviceToHost));
        if (map_normal) HIP_TRY(hipMemcpy(map_normal, c->d_map_nrm + 3 * (size_t)map0, 24 * (size_t)map_cap, hipMemcpyDeviceToHost));
        if (map_desc) HIP_TRY(hipMemcpy(map_desc, c->d_map_desc + 8 * (size_t)map0, 32 * (size_t)map_cap, hipMemcpyDeviceToHost));
        if (map_flags) HIP_TRY(hipMemcpy(map_flags, c->d_map_flags + map0, (size_t)map_cap, hipMemcpyDeviceToHost));
    }
    if (n_active) *n_active = c->n_active;
    if (active && active_cap > 0 && c->n_active > 0) HIP_TRY(hipMemcpy(active, c->d_active, 4 * (size_t)std::min(active_cap, c->n_active), hipMemcpyDeviceToHost));
    return VO_OK;
}
