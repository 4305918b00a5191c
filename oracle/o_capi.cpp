// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_capi.cpp: the C-ABI of include/vo_hip.h implemented on the CPU by the restatement above.
// Built into oracle/_build/liboracle_vo.so; loaded only by tests/, smoke() and bench.py's
// cpu_baseline leg.  Same symbols as the HIP product library so that the same host code
// (rgbd_visualodometry_amd/host) and the same tests drive either side of the boundary.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../include/vo_hip.h"
#include "o_orb.h"
#include "o_track.h"

using namespace orc;

struct vo_ctx {
    vo_params p;
    long long win_obs = -1, win_slots = -1;                 // vo_ba_resident_window: this restatement walks the whole tables
    OrbPlan plan;
    Cam cam;
    struct Slot {
        std::vector<uint8_t> bgr; std::vector<uint16_t> depth;     // owned copies (upload)
        const uint8_t* bgr_p = nullptr; const uint16_t* depth_p = nullptr; int bstride = 0, dstride = 0;
        std::vector<vo_keypoint> kps; std::vector<uint8_t> desc; std::vector<OrbLevelDebug> dbg; bool has_orb = false;
    };
    std::vector<Slot> slots;
    MapStore map;
    Corr corr; std::vector<vo_match> last_matches;
    std::vector<std::vector<vo_match>> lane_matches;           // per lane of the last vo_track_batch (vo_track_fetch_matches)
    bool async_pending = false; std::vector<vo_track_result> async_res;      // vo_track_batch_begin / _end
    std::vector<int32_t> ransac_inliers;
    HypShard shard;
    BaShard ba_shard;
    // wall time per stage of the restatement (vo_profile_read): what bench.py's cpu_baseline breaks its frames/s down into
    double prof_ms[5] = {0, 0, 0, 0, 0}; long long prof_calls[5] = {0, 0, 0, 0, 0};      // orb, filter + match, ransac, pose lm, local ba
    // observation table: keyframe number, map slot, pixel, alive; keyframe poses
    std::vector<int32_t> obs_kf, obs_mp; std::vector<float> obs_uv; std::vector<uint8_t> obs_alive;
    std::vector<double> kf_pose;
    struct Pending { bool ready = false; std::vector<int32_t> pose_kf, point_slots, edge_pose, edge_point; std::vector<float> edge_uv; std::vector<int64_t> edge_obs;
                     std::vector<double> poses, pts; int n_free = 0; double huber = 0, chi2 = 0; } pend;      // between vo_local_ba_resident_cut and _solve
    struct Solved { bool ready = false, merged = false; std::vector<double> poses, pts; std::vector<int32_t> pose_kf, slots; std::vector<int64_t> culled;
                    int n_fixed = 0, n_edges = 0, n_culled = 0, lm_iters = 0; double chi0 = 0, chi1 = 0; } solved, staged;      // solved: between _solve and _merge; staged: _merge's copy for _fetch (a back-end thread may be in the next _cut / _solve meanwhile)
};

extern "C" {

const char* vo_backend_name(void) { return "cpu-oracle"; }
int vo_trace_level(void) { static const int v = [] { const char* e = std::getenv("VO_TRACE"); return e ? std::max(1, std::atoi(e)) : 0; }(); return v; }

const char* vo_strerror(int s) {
    switch (s) {
        case VO_OK: return "ok"; case VO_E_INVALID: return "invalid argument"; case VO_E_NOMEM: return "out of memory";
        case VO_E_DEVICE: return "device error"; case VO_E_OVERFLOW: return "capacity overflow"; case VO_E_STATE: return "bad call sequence";
        case VO_E_UNSUPPORTED: return "unsupported"; default: return "unknown";
    }
}

int vo_default_params(vo_params* p) {
    if (!p) return VO_E_INVALID;
    std::memset(p, 0, sizeof(*p));
    p->width = 640; p->height = 480; p->fx = 517.3f; p->fy = 516.5f; p->cx = 318.6f; p->cy = 255.3f; p->depth_scale = 5000.f;
    p->n_features = 500; p->scale_factor = 1.2f; p->n_levels = 8; p->fast_threshold = 20; p->edge_threshold = 31;
    p->max_frames = 1; p->map_capacity = 1 << 18; p->max_hypotheses = 2048; p->max_track_batch = 1;
    return VO_OK;
}

int vo_default_track_params(vo_track_params* t) {
    if (!t) return VO_E_INVALID;
    std::memset(t, 0, sizeof(*t));
    t->match_ratio = 2.0f; t->match_floor = 30.0f; t->n_hyp = 100; t->reproj_px = 4.0f; t->confidence = 0.99f; t->seed = 0x5eed5eedull;
    t->huber_delta = std::sqrt(7.815); t->chi2_cut = 1.0; t->it_robust = 10; t->it_plain = 10; t->passes = 2;
    return VO_OK;
}

int vo_ctx_create(const vo_params* p, int, vo_ctx** out) {
    if (!p || !out || p->width < 64 || p->height < 64 || p->n_levels < 1 || p->n_levels > 16 || p->max_frames < 1 ||
        p->n_features < 1 || !(p->scale_factor > 1.0f) || p->map_capacity < 1 || p->edge_threshold < 20) return VO_E_INVALID;      // (edge: the steered pattern reaches 19 pixels from the keypoint)
    vo_ctx* c = new (std::nothrow) vo_ctx();
    if (!c) return VO_E_NOMEM;
    c->p = *p;
    orb_build_plan(*p, c->plan);
    c->cam = {(double)p->fx, (double)p->fy, (double)p->cx, (double)p->cy, p->width, p->height};
    c->slots.resize(p->max_frames);
    c->map.pos.assign((size_t)3 * p->map_capacity, 0.0); c->map.nrm.assign((size_t)3 * p->map_capacity, 0.0);
    c->map.desc.assign((size_t)32 * p->map_capacity, 0); c->map.flags.assign(p->map_capacity, 0);
    *out = c;
    return VO_OK;
}
void vo_ctx_destroy(vo_ctx* c) { delete c; }

int vo_frame_upload(vo_ctx* c, int slot, const uint8_t* bgr, int bs, const uint16_t* depth, int ds) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !bgr || !depth || bs < 3 * c->p.width || ds < 2 * c->p.width) return VO_E_INVALID;
    auto& s = c->slots[slot];
    int W = c->p.width, H = c->p.height;
    s.bgr.resize((size_t)3 * W * H); s.depth.resize((size_t)W * H);
    for (int y = 0; y < H; ++y) {
        std::memcpy(&s.bgr[(size_t)3 * W * y], bgr + (size_t)bs * y, 3 * W);
        std::memcpy(&s.depth[(size_t)W * y], (const uint8_t*)depth + (size_t)ds * y, 2 * W);
    }
    s.bgr_p = s.bgr.data(); s.depth_p = s.depth.data(); s.bstride = 3 * W; s.dstride = 2 * W; s.has_orb = false;
    return VO_OK;
}

int vo_frames_preload(vo_ctx* c, int slot0, int n, const uint8_t* const* bgr, int bs, const uint16_t* const* depth, int ds) {      // (the CPU has nothing to overlap a copy with)
    if (!c || slot0 < 0 || n < 1 || slot0 + n > (int)c->slots.size() || !bgr || !depth || bs < 3 * c->p.width || ds < 2 * c->p.width) return VO_E_INVALID;
    return VO_OK;
}

int vo_frame_bind_device(vo_ctx* c, int slot, const void* b, int bs, const void* d, int ds) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !b || !d || bs < 3 * c->p.width || ds < 2 * c->p.width) return VO_E_INVALID;
    auto& s = c->slots[slot];
    s.bgr_p = (const uint8_t*)b; s.depth_p = (const uint16_t*)d; s.bstride = bs; s.dstride = ds; s.has_orb = false;
    return VO_OK;
}

namespace { struct StageTimer { vo_ctx* c; int k; std::chrono::steady_clock::time_point t0; StageTimer(vo_ctx* c_, int k_) : c(c_), k(k_), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() { c->prof_ms[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c->prof_calls[k]++; } }; }
int vo_orb_detect_describe(vo_ctx* c, int slot0, int n) {
    if (!c || slot0 < 0 || n < 1 || slot0 + n > (int)c->slots.size()) return VO_E_INVALID;
    StageTimer tm(c, 0);
    for (int i = slot0; i < slot0 + n; ++i) {
        auto& s = c->slots[i];
        if (!s.bgr_p) return VO_E_STATE;
        orb_detect_describe(c->plan, s.bgr_p, s.bstride, s.depth_p, s.dstride, s.kps, s.desc, &s.dbg);
        s.has_orb = true;
    }
    return VO_OK;
}

int vo_orb_fetch(vo_ctx* c, int slot, vo_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !n_out) return VO_E_INVALID;
    auto& s = c->slots[slot];
    if (!s.has_orb) return VO_E_STATE;
    int n = std::min<int>(cap, (int)s.kps.size());
    if (kps) std::memcpy(kps, s.kps.data(), sizeof(vo_keypoint) * n);
    if (desc) std::memcpy(desc, s.desc.data(), (size_t)32 * n);
    *n_out = (int)s.kps.size();
    return VO_OK;
}

int vo_orb_level_size(vo_ctx* c, int l, int* w, int* h, int* quota) {
    if (!c || l < 0 || l >= c->plan.nlevels) return VO_E_INVALID;
    if (w) *w = c->plan.lw[l];
    if (h) *h = c->plan.lh[l];
    if (quota) *quota = c->plan.quota[l];
    return VO_OK;
}

int vo_orb_fetch_level(vo_ctx* c, int slot, int l, uint8_t* out) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || l < 0 || l >= c->plan.nlevels || !out) return VO_E_INVALID;
    auto& s = c->slots[slot];
    if (!s.has_orb) return VO_E_STATE;
    std::memcpy(out, s.dbg[l].gray.data(), s.dbg[l].gray.size());
    return VO_OK;
}

int vo_orb_fetch_blur_level(vo_ctx* c, int slot, int l, uint8_t* out) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || l < 0 || l >= c->plan.nlevels || !out) return VO_E_INVALID;
    auto& s = c->slots[slot];
    if (!s.has_orb) return VO_E_STATE;
    std::memcpy(out, s.dbg[l].blurred.data(), s.dbg[l].blurred.size());
    return VO_OK;
}

int vo_map_upsert(vo_ctx* c, const int32_t* idx, const double* xyz, const double* nrm, const uint8_t* desc, const uint8_t* flags, int n) {
    if (!c || n < 0 || (n && !idx)) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= c->p.map_capacity) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) {
        size_t k = idx[i];
        if (xyz) std::memcpy(&c->map.pos[3 * k], xyz + 3 * (size_t)i, 24);
        if (nrm) std::memcpy(&c->map.nrm[3 * k], nrm + 3 * (size_t)i, 24);
        if (desc) std::memcpy(&c->map.desc[32 * k], desc + 32 * (size_t)i, 32);
        if (flags) c->map.flags[k] = flags[i];
    }
    return VO_OK;
}

int vo_map_upsert_from_frame(vo_ctx* c, int slot, const int32_t* kp, const int32_t* idx, const double* xyz, const double* nrm, const uint8_t* flags, int n) {
    if (!c || n < 0 || (n && (!idx || !kp)) || slot < 0 || slot >= (int)c->slots.size()) return VO_E_INVALID;
    auto& s = c->slots[slot];
    if (!s.has_orb) return VO_E_STATE;
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= c->p.map_capacity || kp[i] < 0 || kp[i] >= (int)s.kps.size()) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) {
        size_t k = idx[i];
        if (xyz) std::memcpy(&c->map.pos[3 * k], xyz + 3 * (size_t)i, 24);
        if (nrm) std::memcpy(&c->map.nrm[3 * k], nrm + 3 * (size_t)i, 24);
        std::memcpy(&c->map.desc[32 * k], &s.desc[(size_t)32 * kp[i]], 32);     // the keypoint's descriptor row (frontend.cpp:390)
        if (flags) c->map.flags[k] = flags[i];
    }
    return VO_OK;
}

int vo_map_set_active(vo_ctx* c, const int32_t* idx, int n) {
    if (!c || n < 0 || (n && !idx)) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= c->p.map_capacity) return VO_E_INVALID;
    c->map.active.assign(idx, idx + n);
    return VO_OK;
}

static void corr_from_matches(vo_ctx* c, int slot) {
    auto& s = c->slots[slot];
    c->corr.n = (int)c->last_matches.size();
    c->corr.xyz.resize((size_t)3 * c->corr.n); c->corr.uv.resize((size_t)2 * c->corr.n);
    for (int i = 0; i < c->corr.n; ++i) {
        const vo_match& m = c->last_matches[i];
        for (int a = 0; a < 3; ++a) c->corr.xyz[3 * i + a] = (float)c->map.pos[3 * (size_t)m.map_index + a];   // frontend.cpp:228
        c->corr.uv[2 * i] = s.kps[m.kp_index].x; c->corr.uv[2 * i + 1] = s.kps[m.kp_index].y;                  // frontend.cpp:229
    }
}

int vo_match_active_map(vo_ctx* c, int slot, const double T[12], float ratio, float floor_dist, vo_match* out, int cap,
                        int* n_out, int* n_cand, int* min_distance) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !T) return VO_E_INVALID;
    auto& s = c->slots[slot];
    if (!s.has_orb) return VO_E_STATE;
    int nc = 0, md = -1;
    StageTimer tm(c, 1);
    match_active(c->cam, c->map, SE3::from12(T), s.desc.data(), (int)s.kps.size(), ratio, floor_dist, c->last_matches, nc, md);
    corr_from_matches(c, slot);
    c->ransac_inliers.clear();
    if (out) std::memcpy(out, c->last_matches.data(), sizeof(vo_match) * std::min<int>(cap, (int)c->last_matches.size()));
    if (n_out) *n_out = (int)c->last_matches.size();
    if (n_cand) *n_cand = nc;
    if (min_distance) *min_distance = md;
    return VO_OK;
}

int vo_matches_set(vo_ctx* c, const float* xyz, const float* uv, int n) {
    if (!c || n < 0 || (n && (!xyz || !uv))) return VO_E_INVALID;
    c->corr.n = n; c->corr.xyz.assign(xyz, xyz + 3 * (size_t)n); c->corr.uv.assign(uv, uv + 2 * (size_t)n);
    c->last_matches.clear(); c->ransac_inliers.clear();
    return VO_OK;
}

int vo_pnp_ransac(vo_ctx* c, int n_hyp, float reproj_px, float conf, uint64_t seed, double T[12], int32_t* inl, int cap,
                  int* n_inl, int32_t* hyp_counts, int* iters_used, int* best_hyp) {
    if (!c || !T || n_hyp < 1 || n_hyp > c->p.max_hypotheses) return VO_E_INVALID;
    RansacOut r;
    StageTimer tm(c, 2);
    pnp_ransac(c->cam, c->corr, n_hyp, reproj_px, conf, seed, SE3::from12(T), r, &c->shard);
    r.T.to12(T);
    c->ransac_inliers = r.inliers;
    if (inl) std::memcpy(inl, r.inliers.data(), 4 * std::min<size_t>(cap, r.inliers.size()));
    if (n_inl) *n_inl = (int)r.inliers.size();
    if (hyp_counts) std::memcpy(hyp_counts, r.hyp_counts.data(), 4 * (size_t)n_hyp);
    if (iters_used) *iters_used = r.iters_used;
    if (best_hyp) *best_hyp = r.best;
    return VO_OK;
}

int vo_pose_refine_lm(vo_ctx* c, double T[12], double delta, double cut, int it_r, int it_p, uint8_t* mask, int cap,
                      int* n_edges, int* lm_iters) {
    if (!c || !T) return VO_E_INVALID;
    LmOut o;
    StageTimer tm(c, 3);
    pose_lm(c->cam, c->corr, c->ransac_inliers, SE3::from12(T), delta, cut, it_r, it_p, o);
    o.T.to12(T);
    if (mask) std::memcpy(mask, o.inlier_mask.data(), std::min<size_t>(cap, o.inlier_mask.size()));
    if (n_edges) *n_edges = (int)o.inlier_mask.size();
    if (lm_iters) *lm_iters = o.iters;
    return VO_OK;
}

int vo_track_frame(vo_ctx* c, int slot, const double T0[12], const vo_track_params* tp, vo_track_result* res,
                   vo_match* matches, int cap) {
    if (!c || !T0 || !tp || !res || slot < 0 || slot >= (int)c->slots.size() || tp->passes < 1) return VO_E_INVALID;
    if (!c->slots[slot].has_orb) return VO_E_STATE;
    std::memset(res, 0, sizeof(*res));
    double T[12];
    std::memcpy(T, T0, sizeof(T));
    for (int pass = 0; pass < tp->passes; ++pass) {           // coarse, fine (frontend.cpp:100-108)
        int nm = 0, nc = 0, md = -1, ni = 0, iu = 0, bh = -1, ne = 0, li = 0;
        int rc = vo_match_active_map(c, slot, T, tp->match_ratio, tp->match_floor, nullptr, 0, &nm, &nc, &md);
        if (rc) return rc;
        rc = vo_pnp_ransac(c, tp->n_hyp, tp->reproj_px, tp->confidence, tp->seed + (uint64_t)pass, T, nullptr, 0, &ni, nullptr, &iu, &bh);
        if (rc) return rc;
        std::vector<uint8_t> mask(std::max(ni, 1));
        rc = vo_pose_refine_lm(c, T, tp->huber_delta, tp->chi2_cut, tp->it_robust, tp->it_plain, mask.data(), ni, &ne, &li);
        if (rc) return rc;
        for (auto& m : c->last_matches) m.flags = 0;
        int nlm = 0;
        for (int i = 0; i < ni; ++i) {
            c->last_matches[c->ransac_inliers[i]].flags |= VO_MATCH_RANSAC_INLIER;
            if (mask[i]) { c->last_matches[c->ransac_inliers[i]].flags |= VO_MATCH_LM_INLIER; ++nlm; }
        }
        res->n_candidates = nc; res->n_matches = nm; res->n_ransac_inliers = ni; res->n_lm_inliers = nlm;
        res->min_distance = md; res->ransac_iters = iu; res->best_hypothesis = bh; res->lm_iters += li;
    }
    std::memcpy(res->T_cw, T, sizeof(T));
    if (matches) std::memcpy(matches, c->last_matches.data(), sizeof(vo_match) * std::min<size_t>(cap, c->last_matches.size()));
    if ((int)c->last_matches.size() > cap && matches) res->status = VO_E_OVERFLOW;
    c->lane_matches.assign(1, c->last_matches);
    return VO_OK;
}

int vo_track_batch(vo_ctx* c, int n, const int* slots, const double T0[12], const vo_track_params* tp, const uint64_t* seeds,
                   vo_track_result* res, vo_match* matches, int cap) {
    if (!c || n < 1 || !slots || !T0 || !tp || !res || cap < 0) return VO_E_INVALID;
    if (n > std::max(1, c->p.max_track_batch)) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;          // as the HIP library: a chain begun by vo_track_batch_begin is in flight
    std::vector<std::vector<vo_match>> lanes(n);
    for (int i = 0; i < n; ++i) {                  // frames sharing prior + map are independent: a plain loop on the CPU
        vo_track_params t = *tp;
        if (seeds) t.seed = seeds[i];
        int rc = vo_track_frame(c, slots[i], T0, &t, &res[i], matches ? matches + (size_t)i * cap : nullptr, cap);
        if (rc) return rc;
        lanes[i] = c->last_matches;
    }
    c->lane_matches.swap(lanes);
    return VO_OK;
}

// vo_track_batch in two halves: the CPU restatement computes at _begin and hands the results out at _end
int vo_track_batch_begin(vo_ctx* c, int n, const int* slots, const double T0[12], const vo_track_params* tp, const uint64_t* seeds, int cap) {
    if (!c || n < 1 || !slots || !T0 || !tp) return VO_E_INVALID;
    if (c->async_pending) return VO_E_STATE;
    c->async_res.assign((size_t)n, vo_track_result());
    const int rc = vo_track_batch(c, n, slots, T0, tp, seeds, c->async_res.data(), nullptr, cap);
    if (rc != VO_OK) return rc;
    c->async_pending = true;
    return VO_OK;
}
int vo_track_batch_end(vo_ctx* c, vo_track_result* res) {
    if (!c || !res) return VO_E_INVALID;
    if (!c->async_pending) return VO_E_STATE;
    c->async_pending = false;
    for (size_t i = 0; i < c->async_res.size(); ++i) res[i] = c->async_res[i];
    return VO_OK;
}
int vo_track_fetch_matches(vo_ctx* c, int lane, vo_match* matches, int cap, int* n_out) {
    if (!c || lane < 0 || lane >= (int)c->lane_matches.size() || !matches || cap < 0 || !n_out) return VO_E_INVALID;
    const int n = (int)std::min<size_t>((size_t)cap, c->lane_matches[lane].size());
    std::memcpy(matches, c->lane_matches[lane].data(), sizeof(vo_match) * (size_t)n);
    *n_out = n;
    return VO_OK;
}

int vo_set_hypothesis_shard_stream(vo_ctx*, int, int, vo_stream_allreduce_fn, void*) { return VO_E_UNSUPPORTED; }      // no streams on the CPU
int vo_set_hypothesis_shard(vo_ctx* c, int rank, int world, vo_exchange_fn fn, void* user) {
    if (!c || world < 0 || (world > 1 && (rank < 0 || rank >= world || !fn))) return VO_E_INVALID;
    c->shard.rank = rank; c->shard.world = world > 1 ? world : 1; c->shard.exchange = fn; c->shard.user = user;
    return VO_OK;
}

int vo_triangulate_batch(vo_ctx* c, int n, const int32_t* vs, const double* T, const double* xy, double* xyz, uint8_t* ok) {
    if (!c || n < 0 || (n && (!vs || !T || !xy || !xyz || !ok))) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) {
        const int nv = vs[i + 1] - vs[i];
        if (nv < 0) return VO_E_INVALID;
        ok[i] = 0;
        if (nv < 2) continue;
        ok[i] = triangulate_point(nv, T + 12 * (size_t)vs[i], xy + 2 * (size_t)vs[i], xyz + 3 * (size_t)i) ? 1 : 0;
    }
    return VO_OK;
}

// (the FAST candidate lists of the CPU restatement are unbounded vectors; the HIP capacity is sized so that it cannot overflow)
// ---- observation table and resident graph cut: plain loops restating reference src/backend.cpp:36-135 ----------
int vo_kf_set_pose(vo_ctx* c, const int32_t* kf, const double* T, int n) {
    if (!c || n < 0 || (n && (!kf || !T))) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) {
        if (kf[i] < 0) return VO_E_INVALID;
        if ((size_t)12 * (kf[i] + 1) > c->kf_pose.size()) c->kf_pose.resize((size_t)12 * (kf[i] + 1), 0.0);
        std::memcpy(&c->kf_pose[12 * (size_t)kf[i]], T + 12 * (size_t)i, 96);
    }
    return VO_OK;
}
int vo_obs_append(vo_ctx* c, const int32_t* kf, const int32_t* mp, const float* uv, int n, int64_t* first) {
    if (!c || n < 0 || (n && (!kf || !mp || !uv))) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) if (kf[i] < 0 || mp[i] < 0 || mp[i] >= c->p.map_capacity) return VO_E_INVALID;
    {   // same capacity rule as the HIP library (32 Mi observations; VO_OBS_CAP shrinks it for tests)
        const char* env = getenv("VO_OBS_CAP");
        const long long cap = env && atoll(env) > 0 ? std::min<long long>(atoll(env), 32ll << 20) : (32ll << 20);
        if ((long long)c->obs_kf.size() + n > cap) return VO_E_OVERFLOW;
    }
    if (first) *first = (int64_t)c->obs_kf.size();
    for (int i = 0; i < n; ++i) { c->obs_kf.push_back(kf[i]); c->obs_mp.push_back(mp[i]); c->obs_uv.push_back(uv[2 * i]); c->obs_uv.push_back(uv[2 * i + 1]); c->obs_alive.push_back(1); }
    return VO_OK;
}
int vo_obs_kill(vo_ctx* c, const int64_t* ids, int n) {
    if (!c || n < 0 || (n && !ids)) return VO_E_INVALID;
    for (int i = 0; i < n; ++i) { if (ids[i] < 0 || ids[i] >= (int64_t)c->obs_alive.size()) return VO_E_INVALID; c->obs_alive[ids[i]] = 0; }
    return VO_OK;
}
namespace {
struct ResidentGraph { std::vector<int32_t> pose_kf, point_slots, edge_pose, edge_point; std::vector<float> edge_uv; std::vector<int64_t> edge_obs; int n_free = 0; };
int cut_graph(vo_ctx* t, const int32_t* free_kf, int n_free, ResidentGraph& g) {
    const int nkf = (int)(t->kf_pose.size() / 12);
    std::vector<int> kfi(nkf, -1);
    for (int i = 0; i < n_free; ++i) { if (free_kf[i] < 0 || free_kf[i] >= nkf || kfi[free_kf[i]] >= 0) return VO_E_INVALID; kfi[free_kf[i]] = i; }
    g.n_free = n_free; g.pose_kf.assign(free_kf, free_kf + n_free);
    const size_t no = t->obs_kf.size();
    std::vector<char> pflag(t->p.map_capacity, 0);
    for (size_t o = 0; o < no; ++o)                         // points some free keyframe observes, outliers excluded (backend.cpp:62-81)
        if (t->obs_alive[o] && t->obs_kf[o] < nkf && kfi[t->obs_kf[o]] >= 0 && !(t->map.flags[t->obs_mp[o]] & VO_MAP_FLAG_OUTLIER)) pflag[t->obs_mp[o]] = 1;
    std::vector<int> pidx(t->p.map_capacity, -1);
    for (int m = 0; m < t->p.map_capacity; ++m) if (pflag[m]) { pidx[m] = (int)g.point_slots.size(); g.point_slots.push_back(m); }
    std::vector<char> fixed(nkf, 0);
    std::vector<std::vector<int64_t>> per(g.point_slots.size());
    for (size_t o = 0; o < no; ++o) {                       // every live observation of those points (backend.cpp:88-135)
        if (!t->obs_alive[o] || pidx[t->obs_mp[o]] < 0 || t->obs_kf[o] >= nkf) continue;
        per[pidx[t->obs_mp[o]]].push_back((int64_t)o);
        if (kfi[t->obs_kf[o]] < 0) fixed[t->obs_kf[o]] = 1;
    }
    for (int k = 0; k < nkf; ++k) if (fixed[k]) { kfi[k] = (int)g.pose_kf.size(); g.pose_kf.push_back(k); }
    for (size_t p = 0; p < per.size(); ++p) {
        std::stable_sort(per[p].begin(), per[p].end(), [&](int64_t a, int64_t b) { return t->obs_kf[a] < t->obs_kf[b]; });
        for (int64_t o : per[p]) {
            g.edge_pose.push_back(kfi[t->obs_kf[o]]); g.edge_point.push_back((int32_t)p);
            g.edge_uv.push_back(t->obs_uv[2 * o]); g.edge_uv.push_back(t->obs_uv[2 * o + 1]); g.edge_obs.push_back(o);
        }
    }
    return VO_OK;
}
}  // namespace
int vo_ba_resident_set_slab_budget(vo_ctx* c, int64_t bytes) { return (c && bytes > 0) ? VO_OK : VO_E_INVALID; }      // (this restatement has no slab)
int vo_ba_resident_window(vo_ctx* c, int64_t* observations_visited, int64_t* map_slots_visited) {
    if (!c || !observations_visited || !map_slots_visited) return VO_E_INVALID;
    if (c->win_obs < 0) return VO_E_STATE;
    *observations_visited = c->win_obs; *map_slots_visited = c->win_slots;
    return VO_OK;
}
int vo_ba_resident_graph(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int n_free, int32_t* n_poses, int32_t* pose_kf, int cap_poses, int32_t* n_points,
                         int32_t* point_slots, int cap_points, int32_t* n_edges, int32_t* edge_pose, int32_t* edge_point, float* edge_uv, int64_t* edge_obs, int cap_edges) {
    if (!c || !t || n_free < 0 || (n_free && !free_kf) || !n_poses || !n_points || !n_edges) return VO_E_INVALID;
    ResidentGraph g;
    int rc = cut_graph(t, free_kf, n_free, g);
    if (rc == VO_OK) { c->win_obs = (long long)t->obs_kf.size(); c->win_slots = t->p.map_capacity; }
    if (rc) return rc;
    *n_poses = (int)g.pose_kf.size(); *n_points = (int)g.point_slots.size(); *n_edges = (int)g.edge_pose.size();
    for (int i = 0; i < *n_poses && i < cap_poses; ++i) if (pose_kf) pose_kf[i] = g.pose_kf[i];
    for (int i = 0; i < *n_points && i < cap_points; ++i) if (point_slots) point_slots[i] = g.point_slots[i];
    for (int e = 0; e < *n_edges && e < cap_edges; ++e) {
        if (edge_pose) edge_pose[e] = g.edge_pose[e];
        if (edge_point) edge_point[e] = g.edge_point[e];
        if (edge_uv) { edge_uv[2 * e] = g.edge_uv[2 * e]; edge_uv[2 * e + 1] = g.edge_uv[2 * e + 1]; }
        if (edge_obs) edge_obs[e] = g.edge_obs[e];
    }
    return VO_OK;
}
int vo_local_ba_resident_cut(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int n_free, double huber_delta, double chi2_th, int32_t* n_points, int32_t* n_fixed, int32_t* n_edges) {
    if (!c || !t || n_free < 0 || (n_free && !free_kf)) return VO_E_INVALID;
    ResidentGraph g;
    int rc = cut_graph(t, free_kf, n_free, g);
    if (rc == VO_OK) { c->win_obs = (long long)t->obs_kf.size(); c->win_slots = t->p.map_capacity; }
    if (rc) return rc;
    auto& P = c->pend;
    P.pose_kf = g.pose_kf; P.point_slots = g.point_slots; P.edge_pose = g.edge_pose; P.edge_point = g.edge_point; P.edge_uv = g.edge_uv; P.edge_obs = g.edge_obs;
    P.n_free = n_free; P.huber = huber_delta; P.chi2 = chi2_th;
    P.poses.resize(12 * g.pose_kf.size()); P.pts.resize(3 * std::max<size_t>(g.point_slots.size(), 1));
    for (size_t p = 0; p < g.pose_kf.size(); ++p) std::memcpy(&P.poses[12 * p], &t->kf_pose[12 * (size_t)g.pose_kf[p]], 96);      // values are taken NOW: the tables may change afterwards
    for (size_t k = 0; k < g.point_slots.size(); ++k) std::memcpy(&P.pts[3 * k], &t->map.pos[3 * (size_t)g.point_slots[k]], 24);
    P.ready = true;
    if (n_points) *n_points = (int)g.point_slots.size();
    if (n_fixed) *n_fixed = (int)g.pose_kf.size() - n_free;
    if (n_edges) *n_edges = (int)g.edge_pose.size();
    return VO_OK;
}
int vo_local_ba_resident_solve(vo_ctx* c, int it_robust, int it_plain, vo_ba_resident_result* out) {
    if (!c || !out || !c->pend.ready || !out->culled_obs) return VO_E_INVALID;
    const bool deferred = !out->poses && !out->points && !out->point_slots;       // vo_local_ba_resident_merge / _fetch follow
    if (!deferred && (!out->poses || !out->points || !out->point_slots)) return VO_E_INVALID;
    auto& P = c->pend;
    P.ready = false; c->solved.ready = false;
    const int n_free = P.n_free, np = (int)P.pose_kf.size(), nx = (int)P.point_slots.size(), ne = (int)P.edge_pose.size();
    out->n_points = nx; out->n_fixed = np - n_free; out->n_edges = ne; out->n_culled = 0; out->chi2_initial = out->chi2_final = 0; out->lm_iters = 0;
    if (!deferred && nx > out->cap_points) return VO_E_OVERFLOW;
    auto& S = c->solved;
    S.merged = false; S.poses.clear(); S.pts.clear(); S.slots.clear(); S.pose_kf.clear(); S.culled.clear(); S.n_fixed = np - n_free; S.n_edges = ne; S.n_culled = 0; S.lm_iters = 0; S.chi0 = S.chi1 = 0;
    if (nx == 0 || ne == 0 || n_free == 0) { S.ready = true; return VO_OK; }
    std::vector<double> po(12 * (size_t)std::max(n_free, 1)), xo(3 * (size_t)std::max(nx, 1));
    std::vector<uint8_t> fl(std::max(ne, 1));
    if (!deferred) for (int k = 0; k < nx; ++k) out->point_slots[k] = P.point_slots[k];
    vo_ba_problem pr;
    pr.n_poses = np; pr.n_free = n_free; pr.n_points = nx; pr.n_edges = ne; pr.poses = P.poses.data(); pr.points = P.pts.data();
    pr.edge_pose = P.edge_pose.data(); pr.edge_point = P.edge_point.data(); pr.edge_uv = P.edge_uv.data();
    pr.huber_delta = P.huber; pr.chi2_th = P.chi2; pr.it_robust = it_robust; pr.it_plain = it_plain;
    vo_ba_result r;
    std::memset(&r, 0, sizeof(r));
    r.poses = po.data(); r.points = xo.data(); r.edge_flags = fl.data();
    int rc = vo_local_ba(c, &pr, &r);
    if (rc) return rc;
    if (!deferred) { std::memcpy(out->poses, po.data(), 96 * (size_t)n_free); std::memcpy(out->points, xo.data(), 24 * (size_t)nx); }
    for (int e = 0; e < ne; ++e) if (fl[e] & 3) { if (out->n_culled < out->cap_culled) out->culled_obs[out->n_culled] = P.edge_obs[e]; ++out->n_culled; S.culled.push_back(P.edge_obs[e]); }
    std::sort(out->culled_obs, out->culled_obs + std::min(out->n_culled, out->cap_culled));
    out->chi2_initial = r.chi2_initial; out->chi2_final = r.chi2_final; out->lm_iters = r.lm_iters; out->n_pairs = 0;
    po.resize(12 * (size_t)n_free); xo.resize(3 * (size_t)nx);
    S.poses = po; S.pts = xo; S.slots = P.point_slots; S.pose_kf.assign(P.pose_kf.begin(), P.pose_kf.begin() + n_free);
    S.n_culled = out->n_culled; S.lm_iters = r.lm_iters; S.chi0 = r.chi2_initial; S.chi1 = r.chi2_final; S.ready = true;
    return out->n_culled > out->cap_culled ? VO_E_OVERFLOW : VO_OK;
}
// the write-back of reference src/backend.cpp:183-194 on the tables' side: positions of the non-outlier points, free poses, culled observations
int vo_local_ba_resident_merge(vo_ctx* c, vo_ctx* t) {
    if (!c || !t || !c->solved.ready || c->solved.merged) return VO_E_STATE;
    auto& S = c->solved;
    for (size_t k = 0; k < S.slots.size(); ++k) {
        const size_t slot = (size_t)S.slots[k];
        if (t->map.flags[slot] & VO_MAP_FLAG_OUTLIER) continue;
        std::memcpy(&t->map.pos[3 * slot], &S.pts[3 * k], 24);
    }
    for (size_t p = 0; p < S.pose_kf.size(); ++p) std::memcpy(&t->kf_pose[12 * (size_t)S.pose_kf[p]], &S.poses[12 * p], 96);
    for (int64_t id : S.culled) t->obs_alive[(size_t)id] = 0;
    S.merged = true;
    c->staged = S;
    return VO_OK;
}
int vo_local_ba_resident_fetch(vo_ctx* c, vo_ba_resident_result* out) {
    if (!c || !out || !c->staged.merged || !out->poses || !out->points || !out->point_slots) return VO_E_STATE;
    auto& S = c->staged;
    const int nx = (int)S.slots.size();
    out->n_points = nx; out->n_fixed = S.n_fixed; out->n_edges = S.n_edges; out->n_culled = S.n_culled; out->chi2_initial = S.chi0; out->chi2_final = S.chi1; out->lm_iters = S.lm_iters;
    if (nx > out->cap_points) return VO_E_OVERFLOW;
    if (!S.poses.empty()) std::memcpy(out->poses, S.poses.data(), 8 * S.poses.size());
    if (nx) { std::memcpy(out->points, S.pts.data(), 24 * (size_t)nx); std::memcpy(out->point_slots, S.slots.data(), 4 * (size_t)nx); }
    return VO_OK;
}
int vo_local_ba_resident(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int n_free, double huber_delta, double chi2_th, int it_robust, int it_plain,
                         vo_ba_resident_result* out) {
    int rc0 = vo_local_ba_resident_cut(c, t, free_kf, n_free, huber_delta, chi2_th, nullptr, nullptr, nullptr);
    if (rc0) return rc0;
    return vo_local_ba_resident_solve(c, it_robust, it_plain, out);
}
// ---- keyframe bookkeeping on the tables: plain loops over the whole observation table, restating reference
// src/frontend.cpp:366-406 (observations of the LM inliers, new map points), src/frame.cpp:93-152 (covisibility weights),
// src/mappoint.cpp:30-45 (mean viewing direction, outlier when the last observation goes), src/frontend.cpp:465-506 (first-success
// triangulation) and src/mapmanager.cpp:14-38 (local map).  The HIP library walks per-point chains instead; same results.
namespace {
struct KV3 { double v[3]; };
inline KV3 kv3_norm(const KV3& a) { const double n = std::sqrt(a.v[0] * a.v[0] + a.v[1] * a.v[1] + a.v[2] * a.v[2]); return KV3{{a.v[0] / n, a.v[1] / n, a.v[2] / n}}; }
// camera centre = translation of T^-1 = (R^T t) * -1 (SE3::inverse of the host layer, operation for operation)
inline KV3 cam_center(const double T[12]) {
    KV3 c;
    for (int i = 0; i < 3; ++i) c.v[i] = (T[i] * T[9] + T[3 + i] * T[10] + T[6 + i] * T[11]) * -1.0;
    return c;
}
// live observations of a map slot in table order (= ascending keyframe number)
void observers_of(const vo_ctx* c, int slot, std::vector<int64_t>& out) {
    out.clear();
    for (size_t o = 0; o < c->obs_mp.size(); ++o) if (c->obs_alive[o] && c->obs_mp[o] == slot) out.push_back((int64_t)o);
}
}  // namespace

int vo_keyframe_commit(vo_ctx* c, int lane, int frame_slot, int32_t kf, const double T[12], int32_t first_new_slot,
                       int32_t* covis_kf, int32_t* covis_weight, int cap_covis, vo_kf_commit_result* out) {
    if (!c || !T || !out || kf < 0 || frame_slot < 0 || frame_slot >= (int)c->slots.size() || first_new_slot < 0 || cap_covis < 0 || (cap_covis && (!covis_kf || !covis_weight))) return VO_E_INVALID;
    if (lane >= (int)c->lane_matches.size()) return VO_E_INVALID;
    auto& s = c->slots[frame_slot];
    if (!s.has_orb) return VO_E_STATE;
    static const std::vector<vo_match> none;
    const std::vector<vo_match>& ms = lane >= 0 ? c->lane_matches[lane] : none;
    const int nkp = (int)s.kps.size();
    const KV3 C = cam_center(T);
    std::memset(out, 0, sizeof(*out));
    out->first_obs = (int64_t)c->obs_kf.size(); out->triangulated_slot = -1;
    std::vector<char> matched(nkp, 0);
    std::vector<int32_t> inl;                                // map slots of the LM inliers, match order
    for (const vo_match& m : ms) if (m.flags & VO_MATCH_LM_INLIER) { if (m.kp_index < 0 || m.kp_index >= nkp || m.map_index < 0 || m.map_index >= c->p.map_capacity) return VO_E_INVALID; inl.push_back(m.map_index); }
    // covisibility (src/frame.cpp:104-119): every keyframe that already sees an inlier point gains one shared point with the new keyframe
    std::vector<int32_t> w;
    {
        std::vector<char> is_inl(c->p.map_capacity, 0);
        for (int32_t sl : inl) is_inl[sl] = 1;
        for (size_t o = 0; o < c->obs_mp.size(); ++o)
            if (c->obs_alive[o] && is_inl[c->obs_mp[o]] && c->obs_kf[o] != kf) { if ((size_t)c->obs_kf[o] >= w.size()) w.resize((size_t)c->obs_kf[o] + 1, 0); ++w[c->obs_kf[o]]; }
    }
    // AddCurrentKeyframeObservations (src/frontend.cpp:366-370)
    for (const vo_match& m : ms) {
        if (!(m.flags & VO_MATCH_LM_INLIER)) continue;
        const size_t sl = (size_t)m.map_index;
        matched[m.kp_index] = 1;
        c->obs_kf.push_back(kf); c->obs_mp.push_back(m.map_index); c->obs_uv.push_back(s.kps[m.kp_index].x); c->obs_uv.push_back(s.kps[m.kp_index].y); c->obs_alive.push_back(1);
        const KV3 d = kv3_norm(KV3{{c->map.pos[3 * sl] - C.v[0], c->map.pos[3 * sl + 1] - C.v[1], c->map.pos[3 * sl + 2] - C.v[2]}});      // src/mappoint.cpp:30-38
        const KV3 n = kv3_norm(KV3{{c->map.nrm[3 * sl] + d.v[0], c->map.nrm[3 * sl + 1] + d.v[1], c->map.nrm[3 * sl + 2] + d.v[2]}});
        for (int a = 0; a < 3; ++a) c->map.nrm[3 * sl + a] = n.v[a];
        ++out->n_matched;
    }
    // CreateNewMappoints (src/frontend.cpp:372-406): Pixel2World = T^-1 * ((u - cx) d / fx, (v - cy) d / fy, d)
    const double fx = c->p.fx, fy = c->p.fy, cx = c->p.cx, cy = c->p.cy;
    const KV3 tinv = C;
    for (int i = 0; i < nkp; ++i) {
        if (matched[i] || s.kps[i].depth_raw == 0) continue;
        const int32_t sl = first_new_slot + out->n_new;
        if (sl >= c->p.map_capacity) {                      // the map grows (as the product's vo_map_grow: doubling; the reference's container has no size)
            long long cap = c->p.map_capacity;
            while (cap <= sl) cap *= 2;
            if (cap > (1ll << 28)) return VO_E_OVERFLOW;
            c->map.pos.resize((size_t)3 * cap, 0.0); c->map.nrm.resize((size_t)3 * cap, 0.0); c->map.desc.resize((size_t)32 * cap, 0); c->map.flags.resize((size_t)cap, 0);
            c->p.map_capacity = (int32_t)cap;
        }
        const double depth = double(s.kps[i].depth_raw) / c->p.depth_scale;
        const double pc[3] = {((double)s.kps[i].x - cx) * depth / fx, ((double)s.kps[i].y - cy) * depth / fy, depth};
        KV3 pw;
        for (int a = 0; a < 3; ++a) pw.v[a] = (T[a] * pc[0] + T[3 + a] * pc[1] + T[6 + a] * pc[2]) + tinv.v[a];       // R^T p + t'
        const KV3 n = kv3_norm(kv3_norm(KV3{{pw.v[0] - C.v[0], pw.v[1] - C.v[1], pw.v[2] - C.v[2]}}));                    // (0 + d).normalized(), d normalised
        for (int a = 0; a < 3; ++a) { c->map.pos[3 * (size_t)sl + a] = pw.v[a]; c->map.nrm[3 * (size_t)sl + a] = n.v[a]; }
        std::memcpy(&c->map.desc[32 * (size_t)sl], &s.desc[(size_t)32 * i], 32);
        c->map.flags[sl] = 0;
        c->obs_kf.push_back(kf); c->obs_mp.push_back(sl); c->obs_uv.push_back(s.kps[i].x); c->obs_uv.push_back(s.kps[i].y); c->obs_alive.push_back(1);
        ++out->n_new;
    }
    // the keyframe's pose
    if ((size_t)12 * (kf + 1) > c->kf_pose.size()) c->kf_pose.resize((size_t)12 * (kf + 1), 0.0);
    std::memcpy(&c->kf_pose[12 * (size_t)kf], T, 96);
    // TriangulateMappointsInTrackingMap (src/frontend.cpp:465-506): match order, first success wins
    std::vector<int64_t> ob; std::vector<double> Ts, xy;
    for (int32_t sl : inl) {
        if (c->map.flags[sl] & (VO_MAP_FLAG_OUTLIER | VO_MAP_FLAG_TRIANGULATED | VO_MAP_FLAG_OPTIMIZED)) continue;
        ++out->n_tri_candidates;
        observers_of(c, sl, ob);
        if (ob.size() < 2) continue;
        Ts.clear(); xy.clear();
        for (int64_t o : ob) {
            const double* P = &c->kf_pose[12 * (size_t)c->obs_kf[o]];
            Ts.insert(Ts.end(), P, P + 12);
            xy.push_back(((double)c->obs_uv[2 * o] - cx) * 1.0 / fx); xy.push_back(((double)c->obs_uv[2 * o + 1] - cy) * 1.0 / fy);      // Camera::Pixel2Camera, depth 1
        }
        double x[3];
        if (triangulate_point((int)ob.size(), Ts.data(), xy.data(), x) && x[2] > 0) {
            for (int a = 0; a < 3; ++a) c->map.pos[3 * (size_t)sl + a] = x[a];
            c->map.flags[sl] |= VO_MAP_FLAG_TRIANGULATED;
            out->triangulated_slot = sl;
            break;                                           // src/frontend.cpp:501
        }
    }
    int k = 0;
    for (size_t q = 0; q < w.size(); ++q) if (w[q] > 0) { if (k < cap_covis) { covis_kf[k] = (int32_t)q; covis_weight[k] = w[q]; } ++k; }
    out->n_covisible = std::min(k, cap_covis);
    out->n_covisible_total = k;                              // (a cut list: vo_kf_covisibility reads them all)
    return VO_OK;
}

int vo_kf_covisibility(vo_ctx* c, int32_t kf, int32_t* covis_kf, int32_t* covis_weight, int cap, int32_t* n) {
    if (!c || kf < 0 || !n || cap < 0 || (cap && (!covis_kf || !covis_weight))) return VO_E_INVALID;
    std::vector<char> seen(c->p.map_capacity, 0);
    for (size_t o = 0; o < c->obs_mp.size(); ++o) if (c->obs_alive[o] && c->obs_kf[o] == kf) seen[c->obs_mp[o]] = 1;
    std::vector<int32_t> w;
    for (size_t o = 0; o < c->obs_mp.size(); ++o)
        if (c->obs_alive[o] && c->obs_kf[o] != kf && seen[c->obs_mp[o]]) { if ((size_t)c->obs_kf[o] >= w.size()) w.resize((size_t)c->obs_kf[o] + 1, 0); ++w[c->obs_kf[o]]; }
    int k = 0;
    for (size_t q = 0; q < w.size(); ++q) if (w[q] > 0) { if (k < cap) { covis_kf[k] = (int32_t)q; covis_weight[k] = w[q]; } ++k; }
    *n = k;
    return k > cap ? VO_E_OVERFLOW : VO_OK;
}

int vo_map_set_active_covisible(vo_ctx* c, const int32_t* kf, int n, int min_points, int32_t n_map_points, int32_t* n_active) {
    if (!c || n < 0 || (n && !kf) || n_map_points < 0 || n_map_points > c->p.map_capacity) return VO_E_INVALID;
    std::vector<int32_t> ks(kf, kf + n);
    std::sort(ks.begin(), ks.end());
    std::vector<char> seen(c->p.map_capacity, 0);
    c->map.active.clear();
    for (int32_t k : ks)                                     // keyframes in ascending order, each one's observations in table order (src/mapmanager.cpp:14-38)
        for (size_t o = 0; o < c->obs_mp.size(); ++o) {
            if (!c->obs_alive[o] || c->obs_kf[o] != k) continue;
            const int32_t sl = c->obs_mp[o];
            if (seen[sl] || (c->map.flags[sl] & VO_MAP_FLAG_OUTLIER)) continue;
            seen[sl] = 1; c->map.active.push_back(sl);
        }
    if ((int)c->map.active.size() < min_points) {            // src/frontend.cpp:163-166: the whole map
        c->map.active.resize((size_t)n_map_points);
        for (int32_t i = 0; i < n_map_points; ++i) c->map.active[i] = i;
    }
    if (n_active) *n_active = (int32_t)c->map.active.size();
    return VO_OK;
}

int vo_local_ba_resident_merge_ledger(vo_ctx* c, vo_ctx* t, int32_t* pair_a, int32_t* pair_b, int cap_pairs, int32_t* n_pairs, double* poses, int cap_poses) {
    if (!c || !t || !n_pairs || cap_pairs < 0 || (cap_pairs && (!pair_a || !pair_b)) || cap_poses < 0 || (cap_poses && !poses)) return VO_E_INVALID;
    if (!c->solved.ready || c->solved.merged) return VO_E_STATE;
    auto& S = c->solved;
    // Frame::RemoveObservedMappoint (src/frame.cpp:122-152), one culled observation after the other in ascending id order: the keyframes that
    // STILL see the point lose one shared point with the culled observation's keyframe; a point without observations becomes an outlier
    std::vector<int64_t> cu(S.culled.begin(), S.culled.end());
    std::sort(cu.begin(), cu.end());
    int np = 0;
    std::vector<int64_t> ob, undo_alive; std::vector<size_t> undo_flag;
    for (int64_t id : cu) {
        if (id < 0 || id >= (int64_t)t->obs_alive.size() || !t->obs_alive[(size_t)id]) continue;
        t->obs_alive[(size_t)id] = 0; undo_alive.push_back(id);
        observers_of(t, t->obs_mp[(size_t)id], ob);
        for (int64_t o : ob) { if (np < cap_pairs) { pair_a[np] = t->obs_kf[(size_t)id]; pair_b[np] = t->obs_kf[(size_t)o]; } ++np; }
        if (ob.empty() && !(t->map.flags[t->obs_mp[(size_t)id]] & VO_MAP_FLAG_OUTLIER)) { t->map.flags[t->obs_mp[(size_t)id]] |= VO_MAP_FLAG_OUTLIER; undo_flag.push_back((size_t)t->obs_mp[(size_t)id]); }       // src/mappoint.cpp:40-45
    }
    if (np > cap_pairs) {                                    // the caller's arrays are too short: nothing is merged, *n_pairs says what the call needs (include/vo_hip.h)
        for (int64_t id : undo_alive) t->obs_alive[(size_t)id] = 1;
        for (size_t s : undo_flag) t->map.flags[s] &= (uint8_t)~VO_MAP_FLAG_OUTLIER;
        *n_pairs = np;
        return VO_E_OVERFLOW;
    }
    for (size_t k = 0; k < S.slots.size(); ++k) {            // src/backend.cpp:188-194
        const size_t slot = (size_t)S.slots[k];
        t->map.flags[slot] |= VO_MAP_FLAG_OPTIMIZED;
        if (t->map.flags[slot] & VO_MAP_FLAG_OUTLIER) continue;
        std::memcpy(&t->map.pos[3 * slot], &S.pts[3 * k], 24);
    }
    for (size_t p = 0; p < S.pose_kf.size(); ++p) std::memcpy(&t->kf_pose[12 * (size_t)S.pose_kf[p]], &S.poses[12 * p], 96);      // src/backend.cpp:183-187
    if (poses) std::memcpy(poses, S.poses.data(), 8 * std::min(S.poses.size(), (size_t)12 * cap_poses));
    S.merged = true;
    c->staged = S;
    *n_pairs = np;
    return VO_OK;
}

long long vo_scan_call_number(long long) { return 0; }      // (the restatement's prefix sums are loops: nothing is published)

int vo_tables_fetch(vo_ctx* c, int64_t obs0, int64_t obs_cap, int32_t* obs_kf, int32_t* obs_mp, float* obs_uv, uint8_t* obs_alive, int64_t* n_obs,
                    int32_t map0, int32_t map_cap, double* map_xyz, double* map_normal, uint8_t* map_desc, uint8_t* map_flags,
                    int32_t* active, int active_cap, int32_t* n_active) {
    if (!c || obs0 < 0 || obs_cap < 0 || map0 < 0 || map_cap < 0 || map0 + (int64_t)map_cap > c->p.map_capacity || active_cap < 0) return VO_E_INVALID;
    const int64_t no = (int64_t)c->obs_kf.size();
    if (n_obs) *n_obs = no;
    const int64_t take = std::max<int64_t>(0, std::min(obs_cap, no - obs0));
    if (take > 0) {
        if (obs_kf) std::memcpy(obs_kf, &c->obs_kf[(size_t)obs0], 4 * (size_t)take);
        if (obs_mp) std::memcpy(obs_mp, &c->obs_mp[(size_t)obs0], 4 * (size_t)take);
        if (obs_uv) std::memcpy(obs_uv, &c->obs_uv[2 * (size_t)obs0], 8 * (size_t)take);
        if (obs_alive) std::memcpy(obs_alive, &c->obs_alive[(size_t)obs0], (size_t)take);
    }
    if (map_cap > 0) {
        if (map_xyz) std::memcpy(map_xyz, &c->map.pos[3 * (size_t)map0], 24 * (size_t)map_cap);
        if (map_normal) std::memcpy(map_normal, &c->map.nrm[3 * (size_t)map0], 24 * (size_t)map_cap);
        if (map_desc) std::memcpy(map_desc, &c->map.desc[32 * (size_t)map0], 32 * (size_t)map_cap);
        if (map_flags) std::memcpy(map_flags, &c->map.flags[(size_t)map0], (size_t)map_cap);
    }
    if (n_active) *n_active = (int32_t)c->map.active.size();
    if (active) std::memcpy(active, c->map.active.data(), 4 * (size_t)std::min<size_t>(active_cap, c->map.active.size()));
    return VO_OK;
}

// Stream groups: on the CPU every call is computed on the spot; the group only counts (the fused launch chain is a property
// of the HIP implementation, the results are defined to be those of un-grouped calls).
struct vo_group { long long requests = 0; int members = 0; };
int vo_group_create(int, int max_lanes, vo_group** out) { if (!out || max_lanes < 1 || max_lanes > 128) return VO_E_INVALID; *out = new (std::nothrow) vo_group(); return *out ? VO_OK : VO_E_NOMEM; }
void vo_group_destroy(vo_group* g) { delete g; }
int vo_group_join(vo_group* g, vo_ctx* c) { if (!g || !c) return VO_E_INVALID; ++g->members; return VO_OK; }
int vo_group_leave(vo_group* g, vo_ctx* c) { if (!g || !c) return VO_E_INVALID; --g->members; return VO_OK; }
int vo_group_set_gather(vo_group* g, int min_requests, int timeout_us) { return (!g || min_requests < 1 || timeout_us < 0) ? VO_E_INVALID : VO_OK; }
int vo_group_stats(vo_group* g, int64_t* chains, int64_t* lanes, int64_t* requests) { if (!g) return VO_E_INVALID; if (chains) *chains = 0; if (lanes) *lanes = 0; if (requests) *requests = 0; return VO_OK; }

int vo_local_ba(vo_ctx* c, const vo_ba_problem* in, vo_ba_result* out) {
    if (!c || !in || !out || !out->poses || !out->points || !out->edge_flags) return VO_E_INVALID;
    StageTimer tm(c, 4);
    return local_ba(c->cam, *in, *out, c->ba_shard.world > 1 ? &c->ba_shard : nullptr);
}
int vo_set_ba_shard(vo_ctx* c, int rank, int world, vo_exchange_f64_fn fn, void* user) {
    if (!c || world < 0 || world > 64 || (world > 1 && (rank < 0 || rank >= world || !fn))) return VO_E_INVALID;
    c->ba_shard.rank = world > 1 ? rank : 0; c->ba_shard.world = world > 1 ? world : 1; c->ba_shard.exchange = world > 1 ? fn : nullptr; c->ba_shard.user = user;
    return VO_OK;
}
int vo_set_ba_shard_stream(vo_ctx*, int, int, vo_stream_allreduce_f64_fn, void*) { return VO_E_UNSUPPORTED; }      // no streams on the CPU

int vo_sync(vo_ctx*) { return VO_OK; }
int vo_profile_enable(vo_ctx*, int) { return VO_OK; }
int vo_profile_read(vo_ctx* c, char (*names)[48], double* ms, int64_t* calls, int cap, int* n) {      // the restatement's stages (always on): wall ms and calls
    static const char* nm[5] = {"cpu_orb", "cpu_filter_match", "cpu_ransac", "cpu_pose_lm", "cpu_local_ba"};
    if (!c || !n) return VO_E_INVALID;
    *n = 0;
    for (int k = 0; k < 5 && k < cap; ++k) { std::snprintf(names[k], 48, "%s", nm[k]); ms[k] = c->prof_ms[k]; calls[k] = c->prof_calls[k]; *n = k + 1; }
    return VO_OK;
}

}  // extern "C"
