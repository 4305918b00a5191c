// myslam_c.cpp -- C wrapper of the host layer (see include/myslam_c.h).
#include "myslam_c.h"
#include "myslam/util.h"

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>

#include "myslam/backend.h"
#include "myslam/config.h"
#include "myslam/frontend.h"
#include "myslam/mapmanager.h"

using namespace myslam;

struct myslam_system {
    MapManager map;
    Camera::Ptr camera;
    FrontEnd::Ptr frontend;
    Backend::Ptr backend;
    std::deque<Frame::Ptr> queue;
    myslam_options opt;
};

namespace {
thread_local std::string g_err;
template <typename F>
int guarded(myslam_system* s, F&& f) {
    try { if (s) MapManager::BindToThread(&s->map); f(); MapManager::BindToThread(nullptr); return 0; }
    catch (const std::exception& e) { g_err = e.what(); MapManager::BindToThread(nullptr); return -1; }
}
Frame::Ptr make_frame(myslam_system* s, double stamp, const void* bgr, const void* depth, int bs, int ds, int on_device) {
    Image c, d;
    c.data = bgr; c.rows = s->opt.height; c.cols = s->opt.width; c.stride = bs; c.on_device = on_device != 0;
    d.data = depth; d.rows = s->opt.height; d.cols = s->opt.width; d.stride = ds; d.on_device = on_device != 0;
    // With the re-observation pass on, keyframes are detected again at later keyframes (frontend.cpp:408-463): a host frame is then
    // deep-copied like Frame::CreateFrame does, because the caller's buffer is only promised until the frame has been consumed
    // (myslam_c.h).  Device frames are never copied: the caller keeps them alive for the run in that mode (documented there).
    if (s->opt.reobserve_new_mappoints && !on_device) return Frame::CreateFrame(stamp, s->camera, c, d);
    return Frame::CreateFrameView(stamp, s->camera, c, d);      // the caller's buffers stay valid until the frame has been consumed (myslam_c.h)
}
void out_pose(const Frame::Ptr& f, double T_wc[12]) { if (T_wc) f->GetPose().inverse().to12(T_wc); }
}  // namespace

extern "C" {

const char* myslam_last_error(void) { return g_err.c_str(); }
const char* myslam_backend_name(void) { return vo_backend_name(); }

int myslam_default_options(myslam_options* o) {
    if (!o) return -1;
    std::memset(o, 0, sizeof(*o));
    o->width = 640; o->height = 480; o->fx = 517.3f; o->fy = 516.5f; o->cx = 318.6f; o->cy = 255.3f; o->depth_scale = 5000.f;
    o->number_of_features = 500; o->scale_factor = 1.2f; o->level_pyramid = 8; o->match_ratio = 2.0f; o->max_num_lost = 10;
    o->min_inliers = 10; o->keyframe_rotation = 0.05; o->keyframe_translation = 0.05; o->enable_local_optimization = 1; o->chi2_th = 1.f;
    o->ransac_iterations = 100; o->backend_lag_frames = 0; o->max_frames_in_flight = 1; o->track_batch = 1; o->map_capacity = 1 << 22; o->device = 0; o->verbose = 0;
    return 0;
}

int myslam_system_create(const myslam_options* o, const char* yaml, myslam_system** out) {
    if (!o || !out) return -1;
    myslam_system* s = nullptr;
    int rc = guarded(nullptr, [&]() {
        s = new myslam_system();
        s->opt = *o;
        auto S = [](auto v) { return std::to_string(v); };
        Config::set("camera.fx", S(o->fx)); Config::set("camera.fy", S(o->fy)); Config::set("camera.cx", S(o->cx)); Config::set("camera.cy", S(o->cy));
        Config::set("camera.depth_scale", S(o->depth_scale)); Config::set("number_of_features", S(o->number_of_features));
        Config::set("scale_factor", S(o->scale_factor)); Config::set("level_pyramid", S(o->level_pyramid)); Config::set("match_ratio", S(o->match_ratio));
        Config::set("max_num_lost", S(o->max_num_lost)); Config::set("min_inliers", S(o->min_inliers));
        Config::set("keyframe_rotation", S(o->keyframe_rotation)); Config::set("keyframe_translation", S(o->keyframe_translation));
        Config::set("enable_local_optimization", S(o->enable_local_optimization)); Config::set("chi2_th", S(o->chi2_th));
        Config::set("ransac_iterations", S(o->ransac_iterations)); Config::set("track_batch", S(o->track_batch)); Config::set("map_capacity", S(o->map_capacity));
        Config::set("ba_device_graph", S(o->ba_device_graph)); Config::set("triangulate_all", S(o->triangulate_all)); Config::set("reobserve_new_mappoints", S(o->reobserve_new_mappoints));
        Config::set("map_descriptors_on_device", S(o->map_descriptors_on_device)); Config::set("device_keyframes", S(o->device_keyframes));
        if (yaml) Config::setParameterFile(yaml);
        MapManager::BindToThread(&s->map);
        s->camera = Camera::Ptr(new Camera);
        s->frontend = FrontEnd::Ptr(new FrontEnd(o->device, o->width, o->height, o->max_frames_in_flight));
        s->frontend->verbose_ = o->verbose != 0;
        if (Config::get<int>("enable_local_optimization")) { s->backend = Backend::Ptr(new Backend(s->camera)); s->backend->SetLag(o->backend_lag_frames); s->frontend->SetBackend(s->backend); }
    });
    if (rc) { delete s; return rc; }
    *out = s;
    return 0;
}

void myslam_system_destroy(myslam_system* s) {
    if (!s) return;
    MapManager::BindToThread(&s->map);
    if (s->backend) { try { s->backend->Stop(); } catch (...) {} }
    s->queue.clear(); s->frontend.reset(); s->backend.reset();
    MapManager::BindToThread(nullptr);
    delete s;
}

int myslam_prefetch(myslam_system* s, int n, const double* stamps, const void* const* bgr, const void* const* depth, int bs, int ds, int on_device) {
    if (!s || n < 1 || !bgr || !depth) return -1;
    VO_SCOPE("c.prefetch");
    return guarded(s, [&]() {
        if (!s->queue.empty()) throw std::runtime_error("previous prefetched frames not consumed yet");
        std::vector<Frame::Ptr> fr;
        for (int i = 0; i < n; ++i) fr.push_back(make_frame(s, stamps ? stamps[i] : 0.0, bgr[i], depth[i], bs, ds, on_device));
        int done = s->frontend->PrefetchFrames(fr);
        for (int i = 0; i < done; ++i) s->queue.push_back(fr[i]);
    });
}

int myslam_preload(myslam_system* s, int n, const void* const* bgr, const void* const* depth, int bs, int ds) {
    if (!s || n < 1 || !bgr || !depth) return -1;
    return guarded(s, [&]() {
        std::vector<const void*> b(bgr, bgr + n), d(depth, depth + n);
        s->frontend->PreloadFrames(b, d, bs, ds);
    });
}

int myslam_add_prefetched(myslam_system* s, int* tracked, double T_wc[12]) {
    if (!s) return -1;
    VO_SCOPE("c.add_prefetched");
    return guarded(s, [&]() {
        if (s->queue.empty()) throw std::runtime_error("no prefetched frame queued");
        Frame::Ptr f = s->queue.front(); s->queue.pop_front();
        bool ok = s->frontend->AddFrame(f);
        if (tracked) *tracked = ok ? 1 : 0;
        out_pose(f, T_wc);
    });
}

int myslam_add_frame(myslam_system* s, double stamp, const void* bgr, const void* depth, int bs, int ds, int on_device, int* tracked, double T_wc[12]) {
    if (!s || !bgr || !depth) return -1;
    return guarded(s, [&]() {
        if (!s->queue.empty()) throw std::runtime_error("myslam_add_frame: prefetched frames are still queued (consume them with myslam_add_prefetched first)");
        Frame::Ptr f = make_frame(s, stamp, bgr, depth, bs, ds, on_device);
        bool ok = s->frontend->AddFrame(f);
        if (tracked) *tracked = ok ? 1 : 0;
        out_pose(f, T_wc);
    });
}

struct myslam_group { vo_group* g = nullptr; };
int myslam_group_create(int device, int max_lanes, myslam_group** out) {
    if (!out) return -1;
    myslam_group* g = new myslam_group();
    int rc = vo_group_create(device, max_lanes, &g->g);
    if (rc != VO_OK) { g_err = std::string("vo_group_create failed: ") + vo_strerror(rc); delete g; return -1; }
    *out = g;
    return 0;
}
void myslam_group_destroy(myslam_group* g) { if (g) { vo_group_destroy(g->g); delete g; } }
int myslam_group_join(myslam_group* g, myslam_system* s) {
    if (!g || !s) return -1;
    return guarded(s, [&]() { s->frontend->JoinGroup(g->g); });
}
int myslam_group_stats(myslam_group* g, int64_t* chains, int64_t* lanes, int64_t* requests) { return g ? vo_group_stats(g->g, chains, lanes, requests) : -1; }

int myslam_flush(myslam_system* s) {
    if (!s) return -1;
    return guarded(s, [&]() { if (s->backend) s->backend->Flush(); });
}

// ---- taps for parity tests (see include/myslam_c.h) ----------------------------------------------------------------
int myslam_triangulate(int n, const double* T_cw, const double* pts, double out_xyz[3], int* ok) {
    if (n < 1 || !T_cw || !pts || !out_xyz || !ok) return -1;
    std::vector<SE3> poses; std::vector<Vec3> points;
    for (int i = 0; i < n; ++i) { poses.push_back(SE3::from12(T_cw + 12 * i)); points.push_back(Vec3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])); }
    Vec3 p = Vec3::Zero();
    *ok = Triangulation(poses, points, p) ? 1 : 0;
    out_xyz[0] = p[0]; out_xyz[1] = p[1]; out_xyz[2] = p[2];
    return 0;
}
int myslam_se3_log(const double T[12], double out6[6]) { if (!T || !out6) return -1; Vector6d d = SE3::from12(T).log(); for (int i = 0; i < 6; ++i) out6[i] = d[i]; return 0; }
int myslam_se3_exp(const double d6[6], double T_out[12]) { if (!d6 || !T_out) return -1; Vector6d d; for (int i = 0; i < 6; ++i) d[i] = d6[i]; SE3::exp(d).to12(T_out); return 0; }

int myslam_keyframe_policy(myslam_system* s, const double T_ref[12], const double T_cur[12], int num_inliers, int* flags) {
    if (!s || !T_ref || !T_cur || !flags) return -1;
    return guarded(s, [&]() { *flags = s->frontend->PolicyFlags(SE3::from12(T_ref), SE3::from12(T_cur), num_inliers); });
}

namespace {
Frame::Ptr scn_keyframe(myslam_system* s, int64_t id) {
    Frame::Ptr f = s->map.GetKeyframe((size_t)id);
    if (!f) throw std::runtime_error("no such keyframe");
    return f;
}
}  // namespace

int myslam_scn_add_keyframe(myslam_system* s, const double T_cw[12], int64_t* id_out) {
    if (!s || !T_cw || !id_out) return -1;
    return guarded(s, [&]() {
        Image c, d; c.rows = d.rows = s->opt.height; c.cols = d.cols = s->opt.width;
        Frame::Ptr f = Frame::CreateFrame(0.0, s->camera, c, d);
        f->SetPose(SE3::from12(T_cw));
        s->map.InsertKeyframe(f);
        *id_out = (int64_t)f->GetId();
    });
}
int myslam_scn_add_mappoint(myslam_system* s, const double xyz[3], int64_t* id_out) {
    if (!s || !xyz || !id_out) return -1;
    return guarded(s, [&]() {
        Descriptor d; d.fill(0);
        Mappoint::Ptr mp = Mappoint::CreateMappoint(Vector3d(xyz[0], xyz[1], xyz[2]), d);
        s->map.InsertMappoint(mp);
        *id_out = (int64_t)mp->GetId();
    });
}
int myslam_scn_observe(myslam_system* s, int64_t kf, int64_t mp, float u, float v) {
    if (!s) return -1;
    return guarded(s, [&]() { scn_keyframe(s, kf)->AddObservedMappoint((size_t)mp, Point2f(u, v)); });
}
int myslam_scn_unobserve(myslam_system* s, int64_t kf, int64_t mp) {
    if (!s) return -1;
    return guarded(s, [&]() { scn_keyframe(s, kf)->RemoveObservedMappoint((size_t)mp); });
}
int myslam_scn_covisibility(myslam_system* s, int64_t kf, int64_t* ids, int32_t* weights, uint8_t* active, int cap, int* n) {
    if (!s || !n) return -1;
    return guarded(s, [&]() {
        Frame::Ptr f = scn_keyframe(s, kf);
        auto w = f->GetCovisibleKeyframeWeights();
        auto act = f->GetCovisibleKeyframes();
        std::map<size_t, int> ordered(w.begin(), w.end());
        int k = 0;
        for (auto& e : ordered) { if (k < cap) { if (ids) ids[k] = (int64_t)e.first; if (weights) weights[k] = e.second; if (active) active[k] = act.count(e.first) ? 1 : 0; } ++k; }
        for (size_t id : act) if (!w.count(id)) throw std::runtime_error("active covisible keyframe without a weight");
        *n = k;
    });
}
int myslam_scn_local_map(myslam_system* s, int64_t kf, int64_t* mp_ids, int cap, int* n) {
    if (!s || !n) return -1;
    return guarded(s, [&]() {
        Frame::Ptr f = scn_keyframe(s, kf);
        std::vector<Mappoint*> v = s->map.CollectMappointsAroundKeyframe(f);
        auto dict = s->map.GetMappointsAroundKeyframe(f);              // the reference-shaped container must hold the same set
        if (dict.size() != v.size()) throw std::runtime_error("local map: vector and dictionary forms differ");
        for (Mappoint* mp : v) if (!dict.count(mp->GetId())) throw std::runtime_error("local map: vector and dictionary forms differ");
        for (size_t i = 0; i < v.size() && (int)i < cap; ++i) if (mp_ids) mp_ids[i] = (int64_t)v[i]->GetId();
        *n = (int)v.size();
    });
}
int myslam_scn_ba_graph(myslam_system* s, int64_t kf, int64_t* pose_ids, int cap_poses, int* n_poses, int* n_free, int64_t* point_ids, int cap_points,
                        int* n_points, int32_t* edge_pose, int32_t* edge_point, float* edge_uv, int cap_edges, int* n_edges) {
    if (!s || !n_poses || !n_free || !n_points || !n_edges) return -1;
    return guarded(s, [&]() {
        if (!s->backend) throw std::runtime_error("local optimisation is disabled for this system");
        Backend::GraphView g = s->backend->DescribeGraph(scn_keyframe(s, kf));
        *n_poses = (int)g.poseIds.size(); *n_free = g.nFree; *n_points = (int)g.pointIds.size(); *n_edges = (int)g.edgePose.size();
        for (size_t i = 0; i < g.poseIds.size() && (int)i < cap_poses; ++i) if (pose_ids) pose_ids[i] = (int64_t)g.poseIds[i];
        for (size_t i = 0; i < g.pointIds.size() && (int)i < cap_points; ++i) if (point_ids) point_ids[i] = (int64_t)g.pointIds[i];
        for (size_t e = 0; e < g.edgePose.size() && (int)e < cap_edges; ++e) {
            if (edge_pose) edge_pose[e] = g.edgePose[e];
            if (edge_point) edge_point[e] = g.edgePoint[e];
            if (edge_uv) { edge_uv[2 * e] = g.edgeUv[2 * e]; edge_uv[2 * e + 1] = g.edgeUv[2 * e + 1]; }
        }
    });
}
int myslam_scn_mappoint(myslam_system* s, int64_t id, int* outlier, int* n_obs, double xyz[3], double normal[3]) {
    if (!s) return -1;
    return guarded(s, [&]() {
        Mappoint::Ptr mp = s->map.GetMappoint((size_t)id);
        if (!mp) throw std::runtime_error("no such map point");
        if (outlier) *outlier = mp->outlier_ ? 1 : 0;
        if (n_obs) *n_obs = (int)mp->GetObservedByKeyframesMap().size();
        Vector3d p = mp->GetPosition(), nr = mp->GetNormDirection();
        for (int a = 0; a < 3; ++a) { if (xyz) xyz[a] = p[a]; if (normal) normal[a] = nr[a]; }
    });
}
int myslam_scn_run_ba(myslam_system* s, int64_t kf) {
    if (!s) return -1;
    return guarded(s, [&]() {
        if (!s->backend) throw std::runtime_error("local optimisation is disabled for this system");
        s->backend->OptimizeCovisibleGraphOfKeyframe(scn_keyframe(s, kf));
        s->backend->Flush();
    });
}
int myslam_scn_keyframe_pose(myslam_system* s, int64_t kf, double T_cw[12]) {
    if (!s || !T_cw) return -1;
    return guarded(s, [&]() { scn_keyframe(s, kf)->GetPose().to12(T_cw); });
}

int myslam_materialize(myslam_system* s, int64_t* keyframe_ids, int cap, int* n, int* on_device) {
    if (!s || !n) return -1;
    return guarded(s, [&]() {
        s->frontend->MaterializeMap();
        if (on_device) *on_device = s->frontend->KeyframesOnDevice() ? 1 : 0;
        std::map<size_t, Frame::Ptr> ordered;
        for (auto& e : s->map.GetAllKeyframes()) ordered[e.first] = e.second;
        int k = 0;
        for (auto& e : ordered) { if (k < cap && keyframe_ids) keyframe_ids[k] = (int64_t)e.first; ++k; }
        *n = k;
    });
}
int myslam_mappoint_ids(myslam_system* s, int64_t* ids, int cap, int* n) {
    if (!s || !n) return -1;
    return guarded(s, [&]() {
        const auto& all = s->map.AllMappointsOrdered();
        for (size_t i = 0; i < all.size() && (int)i < cap; ++i) if (ids) ids[i] = (int64_t)all[i]->GetId();
        *n = (int)all.size();
    });
}

void* myslam_get_context(myslam_system* s) { return s ? (void*)s->frontend->GetContext() : nullptr; }

int myslam_get_stats(myslam_system* s, myslam_stats* st) {
    if (!s || !st) return -1;
    std::memset(st, 0, sizeof(*st));
    const auto& f = s->frontend->GetStats();
    st->frames = f.frames; st->keyframes = f.keyframes; st->lost = f.lost; st->state = (int)s->frontend->GetState();
    st->last_keypoints = f.last_keypoints; st->last_candidates = f.last_candidates; st->last_matches = f.last_matches;
    st->ms_extract = f.ms_extract; st->ms_track = f.ms_track; st->ms_keyframe = f.ms_keyframe; st->ms_backend = f.ms_backend;
    if (myslam::TraceScope::on()) fprintf(stderr, "[vo_trace] frontend ms: extract %.1f track %.1f (refresh %.1f flush %.1f) keyframe %.1f backend %.1f\n", f.ms_extract, f.ms_track, f.ms_refresh, f.ms_flush, f.ms_keyframe, f.ms_backend);
    st->tracked_frames = f.tracked; st->sum_active = f.sum_active; st->sum_candidates = f.sum_cand; st->sum_matches = f.sum_match; st->sum_ransac_inliers = f.sum_ransac;
    st->sum_lm_inliers = f.sum_lm; st->sum_lm_iters = f.sum_lm_iters; st->track_launches = f.track_launches;
    st->triangulated = f.triangulated; st->reobserved_matches = f.reobserved;
    st->last_ransac_inliers = f.last_ransac; st->last_lm_inliers = f.last_lm; st->map_points = (int)s->map.MappointCount();
    if (s->backend) {
        const auto& b = s->backend->GetStats();
        st->ba_runs = b.runs; st->ba_poses = b.poses; st->ba_fixed = b.fixed; st->ba_points = b.points; st->ba_edges = b.edges; st->ba_outliers = b.outliers; st->ba_ms = b.ms; st->ba_failed = b.failed; st->ba_capped = b.capped;
        st->ba_sum_d3 = b.sum_d3; st->ba_sum_d2 = b.sum_d2; st->ba_sum_edges = b.sum_edges; st->ba_sum_points = b.sum_points; st->ba_sum_pairs = b.sum_pairs;
        if (myslam::TraceScope::on()) fprintf(stderr, "[vo_trace] BA runs %d build %.2f ms solve %.2f ms wait %.2f ms (%d waits: solve done -> tracker awake %.1f us, awake -> merge launched %.1f us per wait)\n", b.runs, b.ms_build, b.ms_solve, b.ms_wait, b.waited, b.waited ? 1e3 * b.ms_wake / b.waited : 0.0, b.waited ? 1e3 * b.ms_to_merge / b.waited : 0.0);
    }
    if (myslam::TraceScope::on()) myslam::TraceScope::dump();
    return 0;
}

}  // extern "C"
