// synth.cpp -- see synth.h.  Plain C++17, no dependencies; deterministic for a given seed.
#include "synth.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t cell_hash(uint64_t seed, int face, int level, int a, int b) {
    return mix(seed ^ mix(((uint64_t)(uint32_t)a << 32) ^ (uint32_t)b ^ ((uint64_t)face << 58) ^ ((uint64_t)level << 52)));
}

struct Pose { double R[9], t[3]; };

void rot_xyz(double rx, double ry, double rz, double R[9]) {   // R = Ry(yaw) * Rx(pitch) * Rz(roll)
    double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
    double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, T[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T[i * 3 + j] = 0; for (int k = 0; k < 3; ++k) T[i * 3 + j] += Rx[i * 3 + k] * Rz[k * 3 + j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { R[i * 3 + j] = 0; for (int k = 0; k < 3; ++k) R[i * 3 + j] += Ry[i * 3 + k] * T[k * 3 + j]; }
}

Pose pose_at(const synth_params& p, int frame) {
    double t = p.speed * frame / p.fps;
    uint64_t h = mix(p.seed ^ 0xA5A5A5A5ull);
    auto ph = [&](int k) { return 6.283185307179586 * (double)((mix(h + k) >> 11) & 0xFFFFF) / 1048576.0; };
    Pose P;
    P.t[0] = 0.40 * std::sin(0.50 * t + ph(0)); P.t[1] = 0.25 * std::sin(0.37 * t + ph(1)); P.t[2] = 0.30 * std::sin(0.29 * t + ph(2));
    // slow continuous turn (0.25 deg/frame at 30 Hz: the camera sweeps all four walls in 1440 frames, so old keyframes
    // drop out of the covisible set instead of the whole run staying inside one local map) plus small oscillations
    double yaw = 0.1309 * t + 0.15 * std::sin(0.23 * t + ph(3)), pitch = 0.15 * std::sin(0.31 * t + ph(4)), roll = 0.08 * std::sin(0.41 * t + ph(5));
    rot_xyz(pitch, yaw, roll, P.R);
    return P;
}

// room half extents (camera: x right, y down, z forward)
const double HX = 3.0, HY = 2.0, HZ = 3.5;

inline void shade(uint64_t seed, int face, double a, double b, float out[3]) {
    int a0 = (int)std::floor(a / 0.25), b0 = (int)std::floor(b / 0.25);
    uint64_t hA = cell_hash(seed, face, 0, a0, b0);
    float v = 40.f + (float)(hA & 0xFF) * (175.f / 255.f);
    int a1 = (int)std::floor(a / 0.06), b1 = (int)std::floor(b / 0.06);
    uint64_t hB = cell_hash(seed, face, 1, a1, b1);
    if (hB & 1) v += ((float)((hB >> 8) & 0xFF) - 127.5f) * (90.f / 255.f);
    int a2 = (int)std::floor(a / 0.02), b2 = (int)std::floor(b / 0.02);
    uint64_t hC = cell_hash(seed, face, 2, a2, b2);
    if (hC & 2) v += ((float)((hC >> 8) & 0xFF) - 127.5f) * (40.f / 255.f);
    v = std::min(255.f, std::max(0.f, v));
    out[0] = v * (0.70f + 0.30f * (float)((hA >> 8) & 0xFF) / 255.f);
    out[1] = v * (0.70f + 0.30f * (float)((hA >> 16) & 0xFF) / 255.f);
    out[2] = v * (0.70f + 0.30f * (float)((hA >> 24) & 0xFF) / 255.f);
}

// returns ray parameter t (= camera-frame depth because the ray's camera-z component is 1)
inline double cast(const Pose& P, double dxc, double dyc, int& face, double& a, double& b) {
    double d[3] = {P.R[0] * dxc + P.R[1] * dyc + P.R[2], P.R[3] * dxc + P.R[4] * dyc + P.R[5], P.R[6] * dxc + P.R[7] * dyc + P.R[8]};
    const double H[3] = {HX, HY, HZ};
    double best = 1e30; int bf = 0;
    for (int k = 0; k < 3; ++k) {
        if (d[k] == 0) continue;
        double bound = d[k] > 0 ? H[k] : -H[k];
        double tt = (bound - P.t[k]) / d[k];
        if (tt > 0 && tt < best) { best = tt; bf = 2 * k + (d[k] > 0); }
    }
    double hit[3] = {P.t[0] + best * d[0], P.t[1] + best * d[1], P.t[2] + best * d[2]};
    face = bf;
    int k = bf >> 1;
    a = hit[(k + 1) % 3]; b = hit[(k + 2) % 3];
    return best;
}

void render_one(const synth_params& p, int frame, uint8_t* bgr, uint16_t* depth) {
    const Pose P = pose_at(p, frame);
    const int W = p.width, H = p.height, ss = p.supersample > 1 ? 2 : 1;
    const uint64_t fseed = mix(p.seed * 0x100000001B3ull + (uint64_t)frame);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float acc[3] = {0, 0, 0};
            for (int sy = 0; sy < ss; ++sy)
                for (int sx = 0; sx < ss; ++sx) {
                    double ox = ss == 1 ? 0.0 : (sx - 0.5) * 0.5, oy = ss == 1 ? 0.0 : (sy - 0.5) * 0.5;
                    int face; double a, b; float c[3];
                    cast(P, (x + ox - p.cx) / p.fx, (y + oy - p.cy) / p.fy, face, a, b);
                    shade(p.seed, face, a, b, c);
                    acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
                }
            int face; double a, b;
            double z = cast(P, (x - p.cx) / p.fx, (y - p.cy) / p.fy, face, a, b);
            uint64_t h = mix(fseed ^ ((uint64_t)y << 20) ^ (uint64_t)x);
            // Irwin-Hall(4) approximates a unit Gaussian: (sum of 4 U(0,1) - 2) * sqrt(3)
            float g1 = ((float)(h & 0xFFF) + (float)((h >> 12) & 0xFFF) + (float)((h >> 24) & 0xFFF) + (float)((h >> 36) & 0xFFF)) / 4096.f - 2.f;
            g1 *= 1.7320508f;
            uint64_t h2 = mix(h);
            float g2 = ((float)(h2 & 0xFFF) + (float)((h2 >> 12) & 0xFFF) + (float)((h2 >> 24) & 0xFFF) + (float)((h2 >> 36) & 0xFFF)) / 4096.f - 2.f;
            g2 *= 1.7320508f;
            float inv = 1.f / (ss * ss);
            for (int c = 0; c < 3; ++c) {
                float v = acc[c] * inv + p.noise_sigma * g1;
                bgr[((size_t)y * W + x) * 3 + c] = (uint8_t)std::min(255.f, std::max(0.f, v + 0.5f));
            }
            double zn = z * (1.0 + (double)p.depth_noise_rel * g2);
            long dq = std::lrint(zn * p.depth_scale);
            bool invalid = (float)((h2 >> 48) & 0xFFFF) / 65536.f < p.invalid_frac;
            depth[(size_t)y * W + x] = (invalid || dq <= 0 || dq > 65535) ? 0 : (uint16_t)dq;
        }
}

}  // namespace

extern "C" {

int synth_default_params(synth_params* p) {
    if (!p) return -1;
    std::memset(p, 0, sizeof(*p));
    p->width = 640; p->height = 480; p->fx = 517.3f; p->fy = 516.5f; p->cx = 318.6f; p->cy = 255.3f; p->depth_scale = 5000.f;
    p->seed = 0; p->noise_sigma = 2.0f; p->depth_noise_rel = 0.005f; p->invalid_frac = 0.05f; p->supersample = 2; p->fps = 30.0; p->speed = 1.0;
    return 0;
}

int synth_pose(const synth_params* p, int frame, double T[12], double* stamp) {
    if (!p || !T) return -1;
    Pose P = pose_at(*p, frame);
    std::memcpy(T, P.R, 72); T[9] = P.t[0]; T[10] = P.t[1]; T[11] = P.t[2];
    if (stamp) *stamp = 1000.0 + frame / p->fps;
    return 0;
}

int synth_render_range(const synth_params* p, int i0, int n, uint8_t* bgr, uint16_t* depth, double* T_wc, double* stamps, int threads) {
    if (!p || n < 0 || !bgr || !depth || p->width < 8 || p->height < 8) return -1;
    const size_t px = (size_t)p->width * p->height;
    threads = std::max(1, std::min(threads, n));
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([=]() { for (int i = t; i < n; i += threads) render_one(*p, i0 + i, bgr + px * 3 * i, depth + px * i); });
    for (auto& th : pool) th.join();
    for (int i = 0; i < n; ++i) {
        double tmp[12];
        synth_pose(p, i0 + i, T_wc ? T_wc + 12 * (size_t)i : tmp, stamps ? stamps + i : nullptr);
    }
    return 0;
}

}  // extern "C"
