// util.cpp -- DLT triangulation (reference include/myslam/util.h:16-34; Eigen::bdcSvd there, a
// Jacobi eigen-decomposition of A^T A here: singular values = sqrt(eigenvalues)).
#include "myslam/util.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace myslam {

void SymmetricEigen4(const double A[16], double ev[4], double V[16]) {
    double a[16];
    std::memcpy(a, A, sizeof(a));
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += a[4 * p + q] * a[4 * p + q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (std::fabs(a[4 * p + q]) < 1e-300) continue;
                double th = (a[4 * q + q] - a[4 * p + p]) / (2 * a[4 * p + q]);
                double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1));
                double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; ++k) { double x = a[4 * k + p], y = a[4 * k + q]; a[4 * k + p] = c * x - s * y; a[4 * k + q] = s * x + c * y; }
                for (int k = 0; k < 4; ++k) { double x = a[4 * p + k], y = a[4 * q + k]; a[4 * p + k] = c * x - s * y; a[4 * q + k] = s * x + c * y; }
                for (int k = 0; k < 4; ++k) { double x = V[4 * k + p], y = V[4 * k + q]; V[4 * k + p] = c * x - s * y; V[4 * k + q] = s * x + c * y; }
            }
    }
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) if (a[5 * order[j]] < a[5 * order[i]]) std::swap(order[i], order[j]);
    double Vs[16];
    for (int i = 0; i < 4; ++i) { ev[i] = a[5 * order[i]]; for (int k = 0; k < 4; ++k) Vs[4 * k + i] = V[4 * k + order[i]]; }
    std::memcpy(V, Vs, sizeof(Vs));
}

bool Triangulation(const std::vector<SE3>& poses, const std::vector<Vec3>& points, Vec3& pt_world) {
    double AtA[16] = {0};
    for (size_t i = 0; i < poses.size(); ++i) {
        double m[12];
        poses[i].matrix3x4(m);
        for (int r = 0; r < 2; ++r) {
            double row[4];
            for (int c = 0; c < 4; ++c) row[c] = points[i][r] * m[8 + c] - m[4 * r + c];
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) AtA[4 * a + b] += row[a] * row[b];
        }
    }
    double ev[4], V[16];
    SymmetricEigen4(AtA, ev, V);                  // ascending: ev[0] = sigma4^2, ev[1] = sigma3^2
    const double w = V[4 * 3 + 0];
    pt_world = Vec3(V[0] / w, V[4] / w, V[8] / w);
    const double s4 = std::sqrt(std::max(ev[0], 0.0)), s3 = std::sqrt(std::max(ev[1], 0.0));
    return s4 / s3 < 1e-2;
}

namespace {
struct TraceRow { const char* name; double ms; long calls; };
std::mutex g_trace_mu;
std::vector<TraceRow> g_trace_rows;
}  // namespace

bool TraceScope::on() { return vo_trace_level() != 0; }
void TraceScope::add(const char* name, double ms) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    for (auto& r : g_trace_rows) if (r.name == name || !strcmp(r.name, name)) { r.ms += ms; ++r.calls; return; }
    g_trace_rows.push_back({name, ms, 1});
}
void TraceScope::dump() {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    for (auto& r : g_trace_rows) fprintf(stderr, "[vo_trace] scope %-28s %9.2f ms  %7ld calls  %8.2f us/call\n", r.name, r.ms, r.calls, 1e3 * r.ms / r.calls);
    g_trace_rows.clear();
}
}  // namespace myslam
