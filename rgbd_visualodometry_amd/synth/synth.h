/* synth.h -- seeded synthetic RGB-D stream generator (SURVEY.md 8d "concrete synthetic inputs").
 * CPU-side data generation only: a textured box room ray-cast along a smooth 6-DoF path, with
 * exact ground-truth poses.  Output formats are the ones the reference's run_vo feeds to
 * Frame::CreateFrame (app/run_vo.cpp:91-100): BGR 8UC3 and depth 16UC1 (depth_scale units/m,
 * 0 = invalid). */
#ifndef VO_SYNTH_H
#define VO_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct synth_params {
    int32_t width, height;
    float fx, fy, cx, cy, depth_scale;
    uint64_t seed;
    float noise_sigma;        /* grey-level noise sigma (2.0)           */
    float depth_noise_rel;    /* depth noise sigma as a fraction of z (0.005) */
    float invalid_frac;       /* fraction of depth pixels forced to 0 (0.05)  */
    int32_t supersample;      /* 1 or 2 (2x2 rays per pixel)            */
    double fps;               /* 30                                     */
    double speed;             /* trajectory speed multiplier (1.0)       */
} synth_params;
int synth_default_params(synth_params* p);
/* Ground-truth camera-to-world pose T_w_c (12 doubles: R row-major, t) and timestamp of frame i. */
int synth_pose(const synth_params* p, int frame, double T_wc[12], double* stamp);
/* Render frames [i0, i0+n) into tightly packed arrays (n*H*W*3 bytes, n*H*W u16), poses n*12. */
int synth_render_range(const synth_params* p, int i0, int n, uint8_t* bgr, uint16_t* depth, double* T_wc, double* stamps, int threads);
#ifdef __cplusplus
}
#endif
#endif
