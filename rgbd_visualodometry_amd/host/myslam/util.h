// util.h -- linear N-view triangulation (reference include/myslam/util.h:16-34).
#ifndef MYSLAM_UTIL_H
#define MYSLAM_UTIL_H
#include "myslam/common_include.h"

namespace myslam {
// DLT: rows x*m2 - m0, y*m2 - m1 per view; smallest right singular vector; success iff
// sigma4 / sigma3 < 1e-2.
bool Triangulation(const std::vector<SE3>& poses, const std::vector<Vec3>& points, Vec3& pt_world);
// symmetric eigen-decomposition (Jacobi), ascending eigenvalues; V columns = eigenvectors
void SymmetricEigen4(const double A[16], double evals[4], double V[16]);
struct KeyPointSet {            // identity set of keypoints of the current frame (reference: hashed cv::KeyPoint)
    std::unordered_set<int> idx;
    size_t count(const KeyPoint& k) const { return idx.count(k.index); }
    void insert(const KeyPoint& k) { idx.insert(k.index); }
    void clear() { idx.clear(); }
    size_t size() const { return idx.size(); }
};
}  // namespace myslam
#endif
