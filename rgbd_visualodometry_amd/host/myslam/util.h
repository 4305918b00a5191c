// util.h -- linear N-view triangulation (reference include/myslam/util.h:16-34).
#ifndef MYSLAM_UTIL_H
#define MYSLAM_UTIL_H
#include <chrono>

#include "myslam/common_include.h"

namespace myslam {
// DLT: rows x*m2 - m0, y*m2 - m1 per view; smallest right singular vector; success iff
// sigma4 / sigma3 < 1e-2.
bool Triangulation(const std::vector<SE3>& poses, const std::vector<Vec3>& points, Vec3& pt_world);
// symmetric eigen-decomposition (Jacobi), ascending eigenvalues; V columns = eigenvectors
void SymmetricEigen4(const double A[16], double evals[4], double V[16]);
struct KeyPointSet {            // identity set of keypoints of the current frame (reference: hashed cv::KeyPoint), flat bitmap
    std::vector<char> hit; size_t n = 0;
    size_t count(const KeyPoint& k) const { return (size_t)k.index < hit.size() && hit[k.index] ? 1 : 0; }
    void insert(const KeyPoint& k) { if ((size_t)k.index >= hit.size()) hit.resize(k.index + 1, 0); if (!hit[k.index]) { hit[k.index] = 1; ++n; } }
    void clear() { hit.assign(hit.size(), 0); n = 0; }
    void reset(size_t cap) { hit.assign(cap, 0); n = 0; }
    size_t size() const { return n; }
};
// Named wall-clock scopes for VO_TRACE=1 runs (host-side cost accounting; off otherwise: one predictable branch).
struct TraceScope {
    static bool on();
    static void add(const char* name, double ms);
    static void dump();                                     // prints and clears the table (stderr)
    const char* name; std::chrono::steady_clock::time_point t0; bool live;
    explicit TraceScope(const char* n) : name(n), live(on()) { if (live) t0 = std::chrono::steady_clock::now(); }
    ~TraceScope() { if (live) add(name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};
#define VO_SCOPE_CAT2(a, b) a##b
#define VO_SCOPE_CAT(a, b) VO_SCOPE_CAT2(a, b)
#define VO_SCOPE(name) ::myslam::TraceScope VO_SCOPE_CAT(vo_scope_, __LINE__)(name)
}  // namespace myslam
#endif
