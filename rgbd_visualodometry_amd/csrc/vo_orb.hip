// vo_orb.hip -- ORB detect + describe on gfx950, replacing cv::ORB::detectAndCompute behind
// FrontEnd::ExtractKeyPointsAndComputeDescriptors (reference src/frontend.cpp:150-154; detector
// parameters :35-37).  All kernels are batched over frame slots (blockIdx.z = slot) so that the
// look-ahead batch of a stream -- or many streams -- runs as one launch chain:
//
//   k_gray      BGR8 -> gray8 level 0            HBM-bound, 16-B per lane stores
//   k_resize    level l from level l-1           11-bit fixed-point bilinear (cv::resize 8u semantics)
//   k_fast_nms  FAST-9/16 score + 3x3 NMS        one 64x16 tile per workgroup, gray tile + halo in LDS,
//                                                scores never leave LDS; survivors appended to a list
//   k_select    retain-best by FAST then Harris  one workgroup per (level, slot): LDS histogram cut,
//                                                exact integer Harris, LDS bitonic sort
//   k_blur      7x7 sigma-2 Gaussian, 8-bit      streaming pass over the whole pyramid (LDS tile + halo)
//   k_describe  IC angle + rBRIEF                one wavefront per keypoint: moments over the radius-15 disc, 256 steered
//                                                tests on the blurred level packed with 4 x __ballot (64 lanes), depth sample
//
// Integer pipeline end to end: bit-exact against oracle/o_orb.cpp by construction.
//
// Frame-to-XCD affinity.  MI355X = 8 XCDs with a private 4 MiB L2 each; workgroups of a 1-D grid are dealt to the XCDs
// round-robin (workgroup b runs on XCD b % 8).  With 8 or more frames in a batch every kernel of the chain sends ALL
// workgroups of frame slot s to XCD s % 8 (vo_slot_block below): a frame's pyramid and its blurred copy (~1 MB each at
// 640x480) are produced and consumed inside one L2 -- level l feeds level l+1, FAST, Harris, blur and the descriptor
// gathers without a trip through HBM, and tile halos are never fetched by two L2s.  Batches of fewer than 8 frames
// spread each frame over all XCDs instead (latency matters there, traffic does not).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cstdio>
#include <vector>

#include "vo_brief_pattern.h"
#include "vo_internal.h"


// rBRIEF test pattern: statically initialised, so every device of the process gets its copy with the code object
// (a run-time upload guarded by a process-wide flag left the second GPU of a process with zeros)
__constant__ int8_t c_pattern[256][4] = VO_BRIEF_PATTERN_INIT;

// blockIdx.x -> (frame slot relative to the batch, block index inside the slot's share of the grid)
__device__ __forceinline__ bool vo_slot_block(int per_slot, int n, int affinity, int& slot_rel, int& j) {
    const int b = blockIdx.x;
    if (affinity) { const int q = b >> 3; slot_rel = (b & 7) + 8 * (q / per_slot); j = q % per_slot; }
    else { slot_rel = b / per_slot; j = b % per_slot; }
    return slot_rel < n;
}
static inline int vo_slot_grid(int per_slot, int n, int affinity) { return affinity ? 8 * per_slot * ((n + 7) / 8) : per_slot * n; }

struct __attribute__((packed)) U32u { uint32_t v; };    // a dword at any byte address (gfx950 global loads take any alignment)

// ------------------------------------------------------------------------------------------
// 16 pixels per lane: 3 x 16-byte loads of BGR, one 16-byte store of gray (the row pitches are multiples of 16 for the
// common widths; otherwise the byte path below).  The image is flattened over (row, 16-pixel group), so every
// workgroup streams 256 x 48 B regardless of the image width.
__device__ __forceinline__ uint32_t gray4(uint32_t w0, uint32_t w1, uint32_t w2) {      // 12 bytes BGRBGRBGRBGR -> 4 gray bytes
    const uint32_t b0 = w0 & 255, g0 = (w0 >> 8) & 255, r0 = (w0 >> 16) & 255, b1 = w0 >> 24;
    const uint32_t g1 = w1 & 255, r1 = (w1 >> 8) & 255, b2 = (w1 >> 16) & 255, g2 = w1 >> 24;
    const uint32_t r2 = w2 & 255, b3 = (w2 >> 8) & 255, g3 = (w2 >> 16) & 255, r3 = w2 >> 24;
    const uint32_t v0 = (b0 * 1868u + g0 * 9617u + r0 * 4899u + 8192u) >> 14, v1 = (b1 * 1868u + g1 * 9617u + r1 * 4899u + 8192u) >> 14;
    const uint32_t v2 = (b2 * 1868u + g2 * 9617u + r2 * 4899u + 8192u) >> 14, v3 = (b3 * 1868u + g3 * 9617u + r3 * 4899u + 8192u) >> 14;
    return v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
}
__global__ __launch_bounds__(256) void k_gray(DevPlan P, const SlotDesc* __restrict__ slots, uint8_t* __restrict__ pyr, int slot0, int n, int aff, int per_slot) {
    int srel, jb;
    if (!vo_slot_block(per_slot, n, aff, srel, jb)) return;
    const int slot = slot0 + srel;
    const int gpr = (P.W + 15) >> 4;                   // 16-pixel groups per row
    const int id = jb * 256 + threadIdx.x;
    if (id >= gpr * P.H) return;
    const int y = id / gpr, x = (id - y * gpr) * 16;
    const SlotDesc sd = slots[slot];
    const uint8_t* row = sd.bgr + (size_t)y * sd.bgr_stride + 3 * (size_t)x;
    uint8_t* out = pyr + (size_t)slot * P.pyr_stride + P.loff[0] + (size_t)y * P.pitch[0] + x;
    if (x + 16 <= P.W && (((uintptr_t)row | (uintptr_t)out) & 15) == 0) {
        const uint4* r4 = (const uint4*)row;
        const uint4 a = r4[0], b = r4[1], c = r4[2];
        uint4 o;
        o.x = gray4(a.x, a.y, a.z); o.y = gray4(a.w, b.x, b.y); o.z = gray4(b.z, b.w, c.x); o.w = gray4(c.y, c.z, c.w);
        *(uint4*)out = o;
    } else {
        const int n = min(16, P.W - x);
        for (int i = 0; i < n; ++i) out[i] = (uint8_t)((row[3 * i] * 1868u + row[3 * i + 1] * 9617u + row[3 * i + 2] * 4899u + 8192u) >> 14);
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(DevPlan P, int l, uint8_t* __restrict__ pyr, const int* __restrict__ tab,
                                                const short* __restrict__ tabs, int slot0, int n, int aff, int gx, int gy) {
    int srel, jb;
    if (!vo_slot_block(gx * gy, n, aff, srel, jb)) return;
    const int slot = slot0 + srel;
    const int dx = (jb % gx) * 64 + threadIdx.x, dy = (jb / gx) * 4 + threadIdx.y;
    const int dw = P.lw[l], dh = P.lh[l], sw = P.lw[l - 1], sh = P.lh[l - 1];
    if (dx >= dw || dy >= dh) return;
    const uint8_t* src = pyr + (size_t)slot * P.pyr_stride + P.loff[l - 1];
    uint8_t* dst = pyr + (size_t)slot * P.pyr_stride + P.loff[l];
    const int sp = P.pitch[l - 1];
    const int sx = tab[P.tabx[l] + dx], sy = tab[P.taby[l] + dy];
    const int a0 = tabs[2 * (P.tabx[l] + dx)], a1 = tabs[2 * (P.tabx[l] + dx) + 1];
    const int b0 = tabs[2 * (P.taby[l] + dy)], b1 = tabs[2 * (P.taby[l] + dy) + 1];
    const int sx1 = min(sx + 1, sw - 1);
    const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* S0 = src + (size_t)r0 * sp;
    const uint8_t* S1 = src + (size_t)r1 * sp;
    const int h0 = S0[sx] * a0 + S0[sx1] * a1;
    const int h1 = S1[sx] * a0 + S1[sx1] * a1;
    int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    dst[(size_t)dy * P.pitch[l] + dx] = (uint8_t)min(255, max(0, v));
}

// ------------------------------------------------------------------------------------------
// The whole pyramid below level 0 in ONE launch.  Level l is a bilinear resize of level l - 1 (cv::ORB resizes level by level), which made
// seven dependent launches of 16-18 us each.  Here a workgroup owns the same fraction (tile tx, ty of gx x gy) of EVERY level and carries it
// down the pyramid in LDS: it loads its part of level 0 plus the halo the seven levels need (vo_ctx::d_pyr_rng: the ranges are the same
// for every frame and are computed once on the host from the resize tables), resizes level after level from one LDS buffer into the
// other and stores the pixels of each level that fall into dwords touching its own tile.  Halo pixels are computed by two neighbours --
// with the same integer formula from the same inputs, so the duplicated dword stores carry identical bytes.  No workgroup waits for
// another one.  Arithmetic = k_resize's, bit for bit.
#define PYR_BUF_A 14592              // bytes: levels 0, 2, 4, 6 of a tile + halo (640 x 480, 5 x 8 tiles: 14504; five workgroups per CU with these sizes)
#define PYR_BUF_B 9984               // levels 1, 3, 5, 7 (9760)
#define PYR_TAB 200                  // table entries per level and axis (196)
struct PyrRng { short x0, x1, y0, y1; };                    // the part of a level a tile needs (half-open; x0 is a multiple of 4)
__global__ __launch_bounds__(256) void k_pyramid(DevPlan P, uint8_t* __restrict__ pyr, const int* __restrict__ tab, const short* __restrict__ tabs,
                                                 const PyrRng* __restrict__ rng, int slot0, int n, int aff, int gx, int gy) {
    __shared__ __align__(16) uint8_t s_a[PYR_BUF_A];
    __shared__ __align__(16) uint8_t s_b[PYR_BUF_B];
    __shared__ int2 s_tx[256], s_ty[PYR_TAB];               // x: (local sx | local sx1 << 16, a0 | a1 << 16), column x at (x & 3) * 64 + (x >> 2): a lane reads columns 4 q + k, so lanes of one k read consecutive entries (a plain table: 32-byte stride, four lanes per bank); y: (local r0 | local r1 << 16, b0 | b1 << 16)
    int srel, jb;
    if (!vo_slot_block(gx * gy, n, aff, srel, jb)) return;
    const int slot = slot0 + srel, tx = jb % gx, ty = jb / gx, tid = threadIdx.x;
    const PyrRng* R = rng + (size_t)jb * VO_MAX_LEVELS;
    uint8_t* base = pyr + (size_t)slot * P.pyr_stride;
    {   // this tile's part of level 0 (written by k_gray) -> LDS, one dword per lane and step
        const PyrRng r = R[0];
        const int wq = (r.x1 - r.x0 + 3) >> 2, h = r.y1 - r.y0;
        const uint8_t* src = base + P.loff[0];
        for (int i = tid; i < wq * h; i += 256) {
            const int j = i / wq, q = i - j * wq;
            reinterpret_cast<uint32_t*>(s_a)[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)(r.y0 + j) * P.pitch[0] + r.x0 + 4 * q);      // (rows are padded to 64 bytes: the last dword of a row stays inside it)
        }
    }
    for (int l = 1; l < P.L; ++l) {
        const PyrRng r = R[l], rp = R[l - 1];
        const int w = r.x1 - r.x0, h = r.y1 - r.y0, wq = (w + 3) >> 2, pp = ((rp.x1 - rp.x0 + 3) >> 2) << 2;      // pp: pitch of the previous level's buffer
        const int sw = P.lw[l - 1], sh = P.lh[l - 1];
        if (tid < w) {
            const int dx = r.x0 + tid, sx = tab[P.tabx[l] + dx], sx1 = min(sx + 1, sw - 1);
            s_tx[(tid & 3) * 64 + (tid >> 2)] = make_int2((sx - rp.x0) | ((sx1 - rp.x0) << 16), (int)(unsigned short)tabs[2 * (P.tabx[l] + dx)] | ((int)tabs[2 * (P.tabx[l] + dx) + 1] << 16));
        }
        if (tid >= 256 - h) {                                // (the y table from the other end of the workgroup: w + h may exceed 256)
            const int j = 255 - tid, dy = r.y0 + j, sy = tab[P.taby[l] + dy];
            const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
            s_ty[j] = make_int2((r0 - rp.y0) | ((r1 - rp.y0) << 16), (int)(unsigned short)tabs[2 * (P.taby[l] + dy)] | ((int)tabs[2 * (P.taby[l] + dy) + 1] << 16));
        }
        __syncthreads();                                     // tables + the previous level's buffer are complete
        const uint8_t* prev = (l & 1) ? s_a : s_b;
        uint8_t* cur = (l & 1) ? s_b : s_a;
        // the tile's own pixels of this level: the same fraction of every level
        const int ox0 = tx * P.lw[l] / gx, ox1 = (tx + 1) * P.lw[l] / gx, oy0 = ty * P.lh[l] / gy, oy1 = (ty + 1) * P.lh[l] / gy;
        uint8_t* dst = base + P.loff[l];
        const int pitch = P.pitch[l];
        // a lane keeps ONE dword column (four pixels) of the tile and walks down its rows: the four x-table entries are read and unpacked once per level
        // instead of once per pixel (half of the kernel's vector instructions were that)
        const int rpp = max(1, 256 / wq), q = tid % wq, rg = tid / wq;      // rows per pass; this lane's column group and first row
        if (rg < rpp) {
            int lx[4], lx1[4], a0[4], a1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * q + k;
                const int2 tx2 = s_tx[k * 64 + q];
                lx[k] = x < w ? (tx2.x & 0xFFFF) : 0; lx1[k] = x < w ? (int)((unsigned)tx2.x >> 16) : 0;
                a0[k] = x < w ? (int)(short)(tx2.y & 0xFFFF) : 0; a1[k] = x < w ? (tx2.y >> 16) : 0;      // (columns past the level: weight 0 -> pixel 0, as before)
            }
            const int gx_ = r.x0 + 4 * q;
            const bool colmine = gx_ < ox1 && gx_ + 4 > ox0;
            for (int j = rg; j < h; j += rpp) {
                const int2 ty2 = s_ty[j];
                const uint8_t* S0 = prev + (ty2.x & 0xFFFF) * pp;
                const uint8_t* S1 = prev + ((unsigned)ty2.x >> 16) * pp;
                const int b0 = (short)(ty2.y & 0xFFFF), b1 = ty2.y >> 16;
                uint32_t o = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int h0 = S0[lx[k]] * a0[k] + S0[lx1[k]] * a1[k];
                    const int h1 = S1[lx[k]] * a0[k] + S1[lx1[k]] * a1[k];
                    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    o |= (4 * q + k < w ? (uint32_t)min(255, max(0, v)) : 0u) << (8 * k);
                }
                reinterpret_cast<uint32_t*>(cur)[j * wq + q] = o;
                const int gy_ = r.y0 + j;
                if (colmine && gy_ >= oy0 && gy_ < oy1) *reinterpret_cast<uint32_t*>(dst + (size_t)gy_ * pitch + gx_) = o;
            }
        }
        __syncthreads();                                     // this level is complete before its tables are overwritten
    }
}
// the ranges of k_pyramid for every tile and level, from the resize tables (host, once per context); false: some tile does not fit the LDS buffers
static bool pyr_ranges(const DevPlan& P, const std::vector<int>& tab, int gx, int gy, std::vector<PyrRng>& out) {
    out.assign((size_t)gx * gy * VO_MAX_LEVELS, PyrRng{0, 0, 0, 0});
    for (int ty = 0; ty < gy; ++ty)
        for (int tx = 0; tx < gx; ++tx) {
            PyrRng* R = &out[(size_t)(ty * gx + tx) * VO_MAX_LEVELS];
            int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;           // what level l + 1 needs of level l
            for (int l = P.L - 1; l >= 0; --l) {
                int x0 = tx * P.lw[l] / gx, x1 = (tx + 1) * P.lw[l] / gx, y0 = ty * P.lh[l] / gy, y1 = (ty + 1) * P.lh[l] / gy;
                if (l == 0) { x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1; }       // level 0 is only read
                else if (l < P.L - 1) { x0 = std::min(x0, nx0); x1 = std::max(x1, nx1); y0 = std::min(y0, ny0); y1 = std::max(y1, ny1); }
                x0 &= ~3;
                x1 = std::min(P.lw[l], x0 + ((x1 - x0 + 3) & ~3));      // whole dwords: every byte a tile stores has been computed (or lies in the row padding)
                if (x1 <= x0 || y1 <= y0) return false;
                R[l] = PyrRng{(short)x0, (short)x1, (short)y0, (short)y1};
                const size_t bytes = (size_t)(((x1 - x0 + 3) >> 2) << 2) * (y1 - y0);
                if (bytes > ((l & 1) ? PYR_BUF_B : PYR_BUF_A) || x1 - x0 > PYR_TAB || y1 - y0 > PYR_TAB || x1 - x0 > 256 || y1 - y0 > 256) return false;
                if (l == 0) break;
                const int sw = P.lw[l - 1], sh = P.lh[l - 1];
                const int* txb = &tab[P.tabx[l]]; const int* tyb = &tab[P.taby[l]];
                nx0 = txb[x0]; nx1 = std::min(txb[x1 - 1] + 1, sw - 1) + 1;
                ny0 = std::min(std::max(tyb[y0], 0), sh - 1); ny1 = std::min(std::max(tyb[y1 - 1] + 1, 0), sh - 1) + 1;
            }
        }
    return true;
}
int vo_orb_pyramid_plan(vo_ctx* c, const std::vector<int>& tab) {
    const DevPlan& P = c->plan;
    c->pyr_gx = 0; c->pyr_gy = 0;
    if (P.L < 2) return VO_OK;
    std::vector<PyrRng> rng;
    for (int shrink = 0; shrink < 4; ++shrink) {               // 128 x 60 pixel tiles of level 0 at 640 x 480; smaller ones if the halo of a deep pyramid does not fit
        const int gx = std::max(1, (P.W + 127) / 128) + shrink, gy = std::max(1, (P.H + 59) / 60) + 2 * shrink;
        if (gx > P.lw[P.L - 1] || gy > P.lh[P.L - 1]) break;
        if (!pyr_ranges(P, tab, gx, gy, rng)) continue;
        if (hipMalloc((void**)&c->d_pyr_rng, rng.size() * sizeof(PyrRng)) != hipSuccess) { c->d_pyr_rng = nullptr; return VO_E_NOMEM; }
        HIP_TRY(hipMemcpy(c->d_pyr_rng, rng.data(), rng.size() * sizeof(PyrRng), hipMemcpyHostToDevice));
        c->pyr_gx = gx; c->pyr_gy = gy;
        if (vo_trace_level()) {
            size_t ma = 0, mb = 0;
            for (int t = 0; t < gx * gy; ++t) for (int l = 0; l < P.L; ++l) { const PyrRng& r = rng[(size_t)t * VO_MAX_LEVELS + l]; const size_t b = (size_t)((r.x1 - r.x0 + 3) & ~3) * (r.y1 - r.y0); if (l & 1) mb = std::max(mb, b); else ma = std::max(ma, b); }
            fprintf(stderr, "[vo_trace] k_pyramid: largest tile parts %zu / %zu bytes (buffers %d / %d)\n", ma, mb, PYR_BUF_A, PYR_BUF_B);
        }
        if (vo_trace_level()) { const PyrRng& a = rng[0]; const PyrRng& b = rng[1]; fprintf(stderr, "[vo_trace] k_pyramid: %d x %d tiles per level; tile 0 needs %d x %d of level 0, %d x %d of level 1\n", gx, gy, a.x1 - a.x0, a.y1 - a.y0, b.x1 - b.x0, b.y1 - b.y0); }
        return VO_OK;
    }
    return VO_OK;                                            // no plan: vo_orb_launch keeps the level-by-level kernels
}

// ------------------------------------------------------------------------------------------
// FAST-9/16 corner score: the largest threshold for which the pixel is still a corner =
// max over the 16 arcs of 9 contiguous ring pixels of min(+diff) or min(-diff), minus 1.
#define TW 64
#define TH VO_FAST_TH
#define GP (TW + 8)          // gray tile pitch (halo 4)
#define SP (TW + 2)          // score tile pitch (halo 1)

// The differences fit 16 bits, so two arcs are walked per instruction: register i holds (d[i], d[i + 8]) and "index i + 8" of any
// intermediate is the same register with its halves swapped (v_pk_min_i16 / v_pk_max_i16; ~125 instructions where the scalar form took ~190).
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s pk_swap(v2s a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2s pk_min(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s pk_max(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }

__device__ __forceinline__ int fast_score_lds(const uint8_t* g, int idx) {
    // ring offsets in a GP-pitch tile, same order as the oracle's RING table
    const int off[16] = {3 * GP, 3 * GP + 1, 2 * GP + 2, GP + 3, 3, -GP + 3, -2 * GP + 2, -3 * GP + 1,
                         -3 * GP, -3 * GP - 1, -2 * GP - 2, -GP - 3, -3, GP - 3, 2 * GP - 2, 3 * GP - 1};
    const short p = g[idx];
    const v2s pp = {p, p};
    v2s A[8], As[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const v2s r = {(short)g[idx + off[i]], (short)g[idx + off[i + 8]]}; A[i] = r - pp; As[i] = pk_swap(A[i]); }
    v2s n2[10], x2[10], n4[12], x4[12], n8[8], x8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const v2s nx = i < 7 ? A[i + 1] : As[0]; n2[i] = pk_min(A[i], nx); x2[i] = pk_max(A[i], nx); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { n2[8 + i] = pk_swap(n2[i]); x2[8 + i] = pk_swap(x2[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { n4[i] = pk_min(n2[i], n2[i + 2]); x4[i] = pk_max(x2[i], x2[i + 2]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { n4[8 + i] = pk_swap(n4[i]); x4[8 + i] = pk_swap(x4[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { n8[i] = pk_min(n4[i], n4[i + 4]); x8[i] = pk_max(x4[i], x4[i + 4]); }
    const v2s zero = {0, 0};
    v2s best = {-256, -256};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const v2s mn9 = pk_min(n8[i], As[i]), mx9 = pk_max(x8[i], As[i]);      // arcs starting at i (low half) and at i + 8 (high half)
        best = pk_max(best, pk_max(mn9, zero - mx9));
    }
    return max((int)best.x, (int)best.y) - 1;
}

__global__ __launch_bounds__(256) void k_fast_nms(DevPlan P, const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                                  int* __restrict__ cand_cnt, int* __restrict__ status, int slot0, int n, int aff, int per_slot) {
    __shared__ __align__(4) uint8_t s_gray[GP * (TH + 8)];
    __shared__ __align__(4) uint8_t s_score[SP * (TH + 2)];
    static_assert(SP * (TH + 2) % 4 == 0 && TH % 4 == 0, "score tile is cleared by dwords");
    int srel, jb;
    if (!vo_slot_block(per_slot, n, aff, srel, jb)) return;
    const int slot = slot0 + srel;
    // few frames: a frame's tiles are spread over the XCDs in 8 contiguous runs (neighbouring tiles share halo rows in one L2)
    const int per_xcd = per_slot >> 3, tile = (!aff && P.xcd_map) ? (jb & 7) * per_xcd + (jb >> 3) : jb;
    if (tile >= P.tile_prefix[P.L]) return;
    int l = 0;
    while (l + 1 < P.L && tile >= P.tile_prefix[l + 1]) ++l;
    const int t = tile - P.tile_prefix[l];
    const int w = P.lw[l], h = P.lh[l], pitch = P.pitch[l];
    const int x0 = P.edge + (t % P.tiles_x[l]) * TW, y0 = P.edge + (t / P.tiles_x[l]) * TH;
    const uint8_t* img = pyr + (size_t)slot * P.pyr_stride + P.loff[l];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    // gray tile with halo 4 (coordinates clamped; clamped positions never produce a score): one dword per lane and
    // step; tile columns start at edge + 64 k - 4, i.e. not dword aligned -- gfx950 global loads take any alignment
    for (int i = tid; i < (GP / 4) * (TH + 8); i += 256) {
        const int d = i % (GP / 4), r = i / (GP / 4);
        const int gx0 = x0 - 4 + 4 * d;
        const int gy = min(max(y0 - 4 + r, 0), h - 1);
        const uint8_t* rowp = img + (size_t)gy * pitch;
        uint32_t v;
        if (gx0 >= 0 && gx0 + 3 < w) v = reinterpret_cast<const U32u*>(rowp + gx0)->v;
        else {
            v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)rowp[min(max(gx0 + b, 0), w - 1)] << (8 * b);
        }
        reinterpret_cast<uint32_t*>(s_gray)[r * (GP / 4) + d] = v;
    }
    __syncthreads();
    // scores on the tile + halo 1, in two phases so that the expensive arc evaluation runs on a dense list:
    // (A) every position: cheap reject (a 9-arc always contains one pixel of each opposite pair) -> LDS work list,
    // (B) lanes walk the list and evaluate the full corner score.  Without the split, one candidate per
    //     wavefront makes all 64 lanes wait for the ~250-instruction score at every position.
    __shared__ __align__(4) uint16_t s_list[SP * (TH + 2)];       // the work list of (B); the survivors of the suppression reuse it
    __shared__ int s_nlist;
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    const int thr = P.fast_thr;
    // (A) four positions per lane: the lane takes one dword of a gray row (pixels c .. c+3 of the tile), the dwords 3 rows above and below
    // and its two neighbours (v_alignbyte gives the pixels 3 to the left and right), and tests the four with packed 16-bit arithmetic
    // (even bytes in one register, odd bytes in another).  Survivors are appended with one LDS atomic per wavefront.
    for (int i = tid; i < SP * (TH + 2) / 4; i += 256) reinterpret_cast<uint32_t*>(s_score)[i] = 0;
    {
        const uint32_t* g32 = reinterpret_cast<const uint32_t*>(s_gray);
        const int xlo = max(3, x0 - 1), xhi = min(w - 3, x0 + TW + 1), ylo = max(3, y0 - 1), yhi = min(h - 3, y0 + TH + 1);   // where a score is wanted
        const short ts = (short)thr; const v2s thr2 = {ts, ts};
        const unsigned long long lt = (1ull << threadIdx.x) - 1;
        constexpr int GD = GP / 4, NA = GD * (TH + 2);
        for (int i0 = 0; i0 < NA; i0 += 256) {
            const int i = min(i0 + tid, NA - 1);
            const int r = i / GD, dq = i - r * GD;              // score row r = gray row r + 3; gray dword dq = tile columns 4 dq .. 4 dq + 3
            const uint32_t* row = g32 + (r + 3) * GD + dq;
            const uint32_t C = row[0], U = row[-3 * GD], D = row[3 * GD], Cp = row[dq > 0 ? -1 : 0], Cn = row[dq < GD - 1 ? 1 : 0];
            const uint32_t Lf = __builtin_amdgcn_alignbyte(C, Cp, 1), Rt = __builtin_amdgcn_alignbyte(Cn, C, 3);
            uint32_t sgn[2];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const uint32_t sel = hb ? 0x0c030c01u : 0x0c020c00u;        // bytes (1, 3) or (0, 2), zero-extended to 16 bits
                const v2s pc = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, C, sel));
                const v2s du = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, U, sel)) - pc, dd = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, D, sel)) - pc;
                const v2s dl = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, Lf, sel)) - pc, dr = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, Rt, sel)) - pc;
                const v2s zero = {0, 0};
                const v2s v = thr2 - pk_max(pk_max(du, zero - du), pk_max(dd, zero - dd));     // negative <=> one of the vertical pair differs by more than thr
                const v2s hh = thr2 - pk_max(pk_max(dl, zero - dl), pk_max(dr, zero - dr));
                sgn[hb] = __builtin_bit_cast(uint32_t, v) & __builtin_bit_cast(uint32_t, hh);
            }
            uint32_t km = ((sgn[0] >> 15) & 1u) | ((sgn[1] >> 14) & 2u) | ((sgn[0] >> 29) & 4u) | ((sgn[1] >> 28) & 8u);
            const int xb = x0 - 4 + 4 * dq, gy = y0 - 1 + r;       // image column of byte 0, image row
            const int jmin = min(max(xlo - xb, 0), 4), jmax = min(max(xhi - xb, 0), 4);
            const uint32_t vm = (jmax > jmin && gy >= ylo && gy < yhi && i0 + tid < NA) ? ((1u << jmax) - (1u << jmin)) : 0u;
            km &= vm;
            const unsigned long long m0 = __ballot(km & 1u), m1 = __ballot(km & 2u), m2 = __ballot(km & 4u), m3 = __ballot(km & 8u);
            const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2), c3 = __popcll(m3);
            if (c0 + c1 + c2 + c3 == 0) continue;                 // wave-uniform
            int wbase = 0;
            if (threadIdx.x == 0) wbase = atomicAdd(&s_nlist, c0 + c1 + c2 + c3);
            wbase = __shfl(wbase, 0, 64);
            const int si = r * SP + 4 * dq - 3;                    // score-tile index of byte 0 (column sx = 4 dq - 3)
            if (km & 1u) s_list[wbase + __popcll(m0 & lt)] = (uint16_t)si;
            if (km & 2u) s_list[wbase + c0 + __popcll(m1 & lt)] = (uint16_t)(si + 1);
            if (km & 4u) s_list[wbase + c0 + c1 + __popcll(m2 & lt)] = (uint16_t)(si + 2);
            if (km & 8u) s_list[wbase + c0 + c1 + c2 + __popcll(m3 & lt)] = (uint16_t)(si + 3);
        }
    }
    __syncthreads();
    const int nlist = s_nlist;
    for (int q = tid; q < nlist; q += 256) {
        const int i = s_list[q];
        const int sx = i % SP, sy = i / SP;
        const int sc = fast_score_lds(s_gray, (sy + 3) * GP + sx + 3);
        if (sc >= thr) s_score[i] = (uint8_t)min(sc, 255);
    }
    __syncthreads();
    // 3x3 non-max suppression (strict), border filter; survivors are gathered in LDS and appended with ONE
    // global atomic per workgroup (a per-survivor atomic on the shared (slot, level) counter serialises at L2:
    // it was 80 % of this kernel's time)
    uint32_t* s_surv = reinterpret_cast<uint32_t*>(s_list);       // [TW * TH / 2]
    static_assert(sizeof(uint16_t) * SP * (TH + 2) >= sizeof(uint32_t) * TW * TH / 2, "survivors fit the work list");
    __shared__ int s_nsurv, s_base;
    if (tid == 0) s_nsurv = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TH / 4; ++k) {
        const int ty = threadIdx.y + 4 * k, tx = threadIdx.x;
        const int x = x0 + tx, y = y0 + ty;
        const uint8_t* q = &s_score[(ty + 1) * SP + tx + 1];
        const int sc = (x < w - P.edge && y < h - P.edge) ? q[0] : 0;
        if (sc == 0) continue;                          // most wavefronts of most tiles leave here
        const int nb = max(max(max((int)q[-1], (int)q[1]), max((int)q[-SP - 1], (int)q[-SP])), max(max((int)q[-SP + 1], (int)q[SP - 1]), max((int)q[SP], (int)q[SP + 1])));
        if (sc > nb) {
            const int pos = atomicAdd(&s_nsurv, 1);      // strict 3x3 maxima: at most every other pixel -> fits TW*TH/2
            s_surv[pos] = ((uint32_t)sc << 24) | ((uint32_t)y << 12) | (uint32_t)x;
        }
    }
    __syncthreads();
    const int ns = s_nsurv;
    if (ns == 0) return;
    if (tid == 0) {
        s_base = atomicAdd(&cand_cnt[slot * VO_MAX_LEVELS + l], ns);
    }
    __syncthreads();
    const int base = s_base;
    for (int i = tid; i < ns; i += 256) {
        const int pos = base + i;
        if (pos < P.ccap[l]) cand[(size_t)slot * P.cprefix[P.L] + P.cprefix[l] + pos] = s_surv[i];
        else *status = VO_E_OVERFLOW;
    }
}

// ------------------------------------------------------------------------------------------
// Harris response of the 7x7 block around (x, y) on Sobel gradients, as an integer key (25 (ab - c^2) - (a + b)^2: k = 0.04 without a division).
// Separable: per row of the 9x9 window the horizontal difference d and the horizontal [1 2 1] sum s of its seven interior columns, then
// ix = d(above) + 2 d + d(below), iy = s(below) - s(above); three rows of d / s roll through registers.  |ix|, |iy| <= 1020 and 49 terms: the
// sums fit 32 bits (5.1e7), only the key needs 64.  (The direct form -- twelve pixels fetched per position -- took twice the instructions.)
__device__ __forceinline__ long long harris_key_dev(const uint8_t* img, int pitch, int x, int y) {
    int a = 0, b = 0, c = 0;
    int d[3][7], s[3][7];
    const uint8_t* q = img + (size_t)(y - 4) * pitch + x - 4;
#pragma unroll
    for (int r = 0; r < 9; ++r, q += pitch) {
        const uint32_t w0 = reinterpret_cast<const U32u*>(q)->v, w1 = reinterpret_cast<const U32u*>(q + 4)->v;
        const int p[9] = {(int)(w0 & 255), (int)((w0 >> 8) & 255), (int)((w0 >> 16) & 255), (int)(w0 >> 24),
                          (int)(w1 & 255), (int)((w1 >> 8) & 255), (int)((w1 >> 16) & 255), (int)(w1 >> 24), (int)q[8]};
        int (&dn)[7] = d[r % 3]; int (&sn)[7] = s[r % 3];
#pragma unroll
        for (int k = 0; k < 7; ++k) { dn[k] = p[k + 2] - p[k]; sn[k] = p[k] + 2 * p[k + 1] + p[k + 2]; }
        if (r >= 2) {
            const int (&du)[7] = d[(r + 1) % 3]; const int (&dm)[7] = d[(r + 2) % 3]; const int (&su)[7] = s[(r + 1) % 3];      // rows r - 2 and r - 1
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int ix = du[k] + 2 * dm[k] + dn[k], iy = sn[k] - su[k];
                a += ix * ix; b += iy * iy; c += ix * iy;
            }
        }
    }
    const long long A = a, B = b, C = c;
    return 25 * (A * B - C * C) - (A + B) * (A + B);
}

__device__ __forceinline__ bool sel_before(long long ka, uint32_t ia, long long kb, uint32_t ib) {
    return ka > kb || (ka == kb && ia < ib);
}

// one workgroup per (level, slot)
__global__ __launch_bounds__(1024) void k_select(DevPlan P, const uint8_t* __restrict__ pyr, const uint32_t* __restrict__ cand,
                                                 const int* __restrict__ cand_cnt, uint32_t* __restrict__ sel, long long* __restrict__ sel_key,
                                                 int* __restrict__ sel_cnt, int slot0, int nslots, int aff) {
    extern __shared__ __align__(16) unsigned char smem[];
    long long* s_key = (long long*)smem;                            // [sel_cap]
    uint32_t* s_idx = (uint32_t*)(smem + (size_t)P.sel_cap * 8);    // [sel_cap]  packed (y<<12|x) doubles as the tie-break index y*4096+x
    int* s_hist = (int*)(smem + (size_t)P.sel_cap * 12);            // [256] + misc
    int* s_misc = s_hist + 256;
    int srel, l;
    if (!vo_slot_block(P.L, nslots, aff, srel, l)) return;
    const int slot = slot0 + srel, tid = threadIdx.x;
    const int n = min(cand_cnt[slot * VO_MAX_LEVELS + l], P.ccap[l]);
    const int quota = P.quota[l];
    const uint32_t* cl = cand + (size_t)slot * P.cprefix[P.L] + P.cprefix[l];
    const uint8_t* img = pyr + (size_t)slot * P.pyr_stride + P.loff[l];
    const int pitch = P.pitch[l];
    if (tid < 256) s_hist[tid] = 0;
    if (tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&s_hist[cl[i] >> 24], 1);
    __syncthreads();
    // the cut: the largest score s whose suffix sum hist[s] + .. + hist[255] reaches 2 x quota.  Four wavefronts scan the 256 bins (one lane walking them
    // was 256 dependent LDS reads: 7 us of the level-0 workgroup's 43, and that workgroup is the launch's duration)
    int* const s_scr = s_misc + 8;                              // (the two 4096-bin passes below clear what they use of this area)
    const int lane_c = tid & 63, wave_c = tid >> 6;
    int suf = 0;
    if (tid < 256) {
        suf = s_hist[tid];                                      // inclusive suffix sum inside the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(suf, o, 64); if (lane_c + o < 64) suf += t; }
        if (lane_c == 0) s_scr[wave_c] = suf;
    }
    __syncthreads();
    if (tid < 256) {
        for (int w = wave_c + 1; w < 4; ++w) suf += s_scr[w];
        const bool reach = n > 2 * quota && suf >= 2 * quota;   // true for every bin up to the cut (suffix sums do not grow with s)
        const unsigned long long m = __ballot(reach);
        if (lane_c == 0) s_scr[4 + wave_c] = __popcll(m);
        s_scr[16 + tid] = suf;
    }
    __syncthreads();
    if (tid == 0) {
        int thr = 0, need = -1;
        if (n > 2 * quota) {
            thr = s_scr[4] + s_scr[5] + s_scr[6] + s_scr[7] - 1;
            const int acc = s_scr[16 + thr];
            if (acc > 4 * quota) need = 2 * quota - (acc - s_hist[thr]);      // the ties at the cut do not fit: only `need` of them stay
        }
        s_misc[1] = thr; s_misc[2] = need;
    }
    __syncthreads();
    const int thr = s_misc[1], need = s_misc[2];
    // Overflowing cut bin (saturated / repetitive images: thousands of corners with one score): the ties are ranked by pixel index
    // (row, then column) and exactly the first `need` stay -- two histogram passes over the bin, independent of the arrival order
    // of the candidate list.  (The bin used to be dropped whole, which could leave a level with fewer keypoints than its quota.)
    int ycut = -1, xcut = -1;
    if (need >= 0) {
        int* s_h2 = s_misc + 8;                              // [4096]: rows, then columns (coordinates are 12 bits)
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = tid; i < 4096; i += 1024) s_h2[i] = 0;
            __syncthreads();
            const int yc = s_misc[3];
            for (int i = tid; i < n; i += 1024) {
                const uint32_t c = cl[i];
                if ((int)(c >> 24) != thr) continue;
                const int x = c & 0xFFF, y = (c >> 12) & 0xFFF;
                if (pass == 0) atomicAdd(&s_h2[y], 1); else if (y == yc) atomicAdd(&s_h2[x], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int want = pass == 0 ? need : s_misc[4], cum = 0, k = 0;
                for (; k < 4096; ++k) { if (cum + s_h2[k] >= want) break; cum += s_h2[k]; }
                if (pass == 0) { s_misc[3] = k; s_misc[4] = want - cum; } else s_misc[5] = k;      // row ycut holds the rank; `want - cum` of its columns stay
            }
            __syncthreads();
        }
        ycut = s_misc[3]; xcut = s_misc[5];
    }
    // two steps: first the survivors of the cut are gathered (one LDS atomic per wavefront), then every lane takes ONE of them to its Harris key.  (Computed
    // inside the gathering loop, the ~600-instruction key ran in each of its trips with the quarter of the lanes that had a survivor there.)
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const uint32_t c = i < n ? cl[i] : 0u;
        const int sc = (int)(c >> 24), cx = c & 0xFFF, cy = (c >> 12) & 0xFFF;
        const bool pass = i < n && (sc > thr || (sc == thr && (need < 0 || cy < ycut || (cy == ycut && cx <= xcut))));
        const unsigned long long m = __ballot(pass);
        int wbase = 0;
        if (lane_c == 0 && m) wbase = atomicAdd(&s_misc[0], __popcll(m));      // at most 4 x quota <= sel_cap entries pass the cut (see above): the order of arrival only
        wbase = __shfl(wbase, 0, 64);                                         // permutes the list that the sort below orders by (Harris key, pixel index)
        const int pos = wbase + __popcll(m & ((1ull << lane_c) - 1ull));
        if (pass && pos < P.sel_cap) s_idx[pos] = c & 0xFFFFFF;
    }
    __syncthreads();
    const int kept = min(s_misc[0], P.sel_cap);
    for (int p = tid; p < kept; p += 1024) { const uint32_t c = s_idx[p]; s_key[p] = harris_key_dev(img, pitch, c & 0xFFF, (c >> 12) & 0xFFF); }
    __syncthreads();
    int m = 64;
    while (m < kept) m <<= 1;
    for (int i = kept + tid; i < m; i += 1024) { s_key[i] = LLONG_MIN; s_idx[i] = 0xFFFFFFFFu; }
    __syncthreads();
    // Bitonic sort into "before" order.  Compare-exchange steps whose partners are less than 64 apart stay inside an aligned block of 64
    // elements: a wavefront takes the block into registers and runs all of them (for one k, or for every k up to 64 at the start) with
    // lane exchanges -- 21 workgroup barriers for 2048 elements instead of 66.  Only the steps with partners >= 64 apart go through LDS.
    const int lane = tid & 63, wv = tid >> 6;
    auto reg_pass = [&](int k_lo, int k_hi) {                // the steps j = min(k / 2, 32) .. 1 of every k in [k_lo, k_hi] on blocks of 64
        for (int blk = wv; blk < (m >> 6); blk += 16) {
            const int i = (blk << 6) + lane;
            long long key = s_key[i]; uint32_t idx = s_idx[i];
            for (int k = k_lo; k <= k_hi; k <<= 1) {
                const bool up = (i & k) == 0;
                for (int j = min(k >> 1, 32); j > 0; j >>= 1) {
                    const long long pk = __shfl_xor(key, j, 64); const uint32_t pi = (uint32_t)__shfl_xor((int)idx, j, 64);
                    const bool want_before = ((lane & j) == 0) == up;       // the lower position of an ascending pair (or the upper of a descending one) keeps the element that comes first
                    const bool take = want_before ? sel_before(pk, pi, key, idx) : sel_before(key, idx, pk, pi);
                    if (take) { key = pk; idx = pi; }
                }
            }
            s_key[i] = key; s_idx[i] = idx;
        }
    };
    reg_pass(2, 64);
    __syncthreads();
    for (int k = 128; k <= m; k <<= 1) {
        for (int j = k >> 1; j >= 64; j >>= 1) {
            for (int i = tid; i < m; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const long long ka = s_key[i], kb = s_key[p];
                    const uint32_t ia = s_idx[i], ib = s_idx[p];
                    const bool up = (i & k) == 0;          // ascending position = "before" order
                    const bool swap = up ? sel_before(kb, ib, ka, ia) : sel_before(ka, ia, kb, ib);
                    if (swap) { s_key[i] = kb; s_key[p] = ka; s_idx[i] = ib; s_idx[p] = ia; }
                }
            }
            __syncthreads();
        }
        reg_pass(k, k);
        __syncthreads();
    }
    const int out_n = min(kept, quota);
    for (int i = tid; i < out_n; i += 1024) {
        sel[(size_t)slot * P.nfeat + P.qprefix[l] + i] = s_idx[i];
        sel_key[(size_t)slot * P.nfeat + P.qprefix[l] + i] = s_key[i];
    }
    if (tid == 0) sel_cnt[slot * VO_MAX_LEVELS + l] = out_n;
}

// ------------------------------------------------------------------------------------------
// 7x7 sigma-2 Gaussian blur of every pyramid level in 8-bit fixed point (cv::GaussianBlur semantics for 8-bit images:
// integer kernel round(k*256), row pass in int, column pass (sum + 2^15) >> 16), BORDER_REFLECT_101.  Separable in
// LDS: 128x16 output tile, (128+8)x(16+6) input halo, u16 intermediate.  Streaming kernel: reads P, writes P.
#define BTW 128
#define BTH 16
#define BSTR (BTW + 16)                                 // s_in row stride in bytes: 4 halo bytes left, 4+ right, 16-byte multiple
__global__ __launch_bounds__(256) void k_blur(DevPlan P, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int slot0, int n, int aff, int per_slot) {
    // 4 pixels per lane in every phase: dword loads/stores to HBM and LDS (rows and level offsets are 64-byte aligned)
    __shared__ uint32_t s_in[(BTH + 6) * BSTR / 4];
    __shared__ uint32_t s_h[(BTH + 6) / 2 * BTW];      // u16 row sums: rows 2 k and 2 k + 1 of a column in one dword
    int srel, jb;
    if (!vo_slot_block(per_slot, n, aff, srel, jb)) return;
    const int slot = slot0 + srel;
    // few frames: workgroups go round-robin over the 8 XCDs (each with its own L2), so block j of the frame works on
    // tile (j % 8) * (tiles / 8) + j / 8 -- neighbouring tiles, which share halo rows, meet in the same L2
    const int per_xcd = per_slot >> 3, tile = (!aff && P.xcd_map) ? (jb & 7) * per_xcd + (jb >> 3) : jb;
    if (tile >= P.btile_prefix[P.L]) return;
    int l = 0;
    while (l + 1 < P.L && tile >= P.btile_prefix[l + 1]) ++l;
    const int t = tile - P.btile_prefix[l];
    const int w = P.lw[l], h = P.lh[l], pitch = P.pitch[l];
    const int x0 = (t % P.btiles_x[l]) * BTW, y0 = (t / P.btiles_x[l]) * BTH;
    const uint8_t* img = pyr + (size_t)slot * P.pyr_stride + P.loff[l];
    uint8_t* out = blur + (size_t)slot * P.pyr_stride + P.loff[l];
    const int tx = threadIdx.x, ty = threadIdx.y;      // 32 x 8
    // halo tile: image columns x0-4 .. x0+BTW+3 as (BTW+8)/4 dwords per row, reflect-101 at the image border
    {
        constexpr int HD = (BTW + 8) / 4;                   // dwords per halo row
        const int tid = ty * 32 + tx;
        for (int i = tid; i < HD * (BTH + 6); i += 256) {   // (flat: a row is 34 dwords, a lane row of 32 would idle through a second trip)
            const int r = i / HD, d = i - r * HD;
            int gy = y0 - 3 + r;
            gy = gy < 0 ? -gy : (gy >= h ? 2 * h - 2 - gy : gy);
            gy = min(max(gy, 0), h - 1);
            const uint8_t* rowp = img + (size_t)gy * pitch;
            const int gx0 = x0 - 4 + 4 * d;
            uint32_t v;
            if (gx0 >= 0 && gx0 + 3 < w) v = *(const uint32_t*)(rowp + gx0);
            else {
                v = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int gx = gx0 + k;
                    gx = gx < 0 ? -gx : (gx >= w ? 2 * w - 2 - gx : gx);
                    gx = min(max(gx, 0), w - 1);
                    v |= (uint32_t)rowp[gx] << (8 * k);
                }
            }
            s_in[r * (BSTR / 4) + d] = v;
        }
    }
    __syncthreads();
    const uint32_t g0 = P.gk[0], g1 = P.gk[1], g2 = P.gk[2], g3 = P.gk[3];      // symmetric 7-tap kernel, round(k * 256): every tap fits a byte
    // Row pass on bytes: output i of a lane's four needs bytes i+1 .. i+7 of its 12-byte window [a | b | c]: two v_dot4_u32_u8 on windows cut out with
    // v_alignbyte (weights (g0 g1 g2 g3) and (g2 g1 g0 0)).  A lane takes TWO rows (2 rp, 2 rp + 1) and stores the sums of a column as one dword
    // (even row low, odd row high): the column pass then reads row PAIRS and is four v_dot2_u32_u16 per output.  Columns are stored permuted
    // (column 4 tx + i at i * 32 + tx) so that both passes touch consecutive dwords in consecutive lanes.
    const uint32_t G1 = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24), G2 = g2 | (g1 << 8) | (g0 << 16);
    for (int rp = ty; rp < (BTH + 6) / 2; rp += 8) {
        uint32_t hs[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t* q = &s_in[(2 * rp + e) * (BSTR / 4) + tx];
            const uint32_t a = q[0], b = q[1], c = q[2];   // bytes x-4 .. x+7 of this lane's 4 outputs x .. x+3
            hs[e][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 1), G1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 1), G2, 0u, false), false);
            hs[e][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 2), G1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 2), G2, 0u, false), false);
            hs[e][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 3), G1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 3), G2, 0u, false), false);
            hs[e][3] = __builtin_amdgcn_udot4(b, G1, __builtin_amdgcn_udot4(c, G2, 0u, false), false);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s_h[rp * BTW + i * 32 + tx] = hs[0][i] | (hs[1][i] << 16);      // (a row sum is at most 255 * 256)
    }
    __syncthreads();
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    // output row y needs the row sums y .. y+6: for even y three whole pairs and the low half of a fourth, for odd y the high half of a
    // pair and three whole ones -- four dot products either way, with the weights chosen by the parity of the lane's rows (ty: BTH and 8 are even)
    const bool odd = ty & 1;
    const us2 W0 = __builtin_bit_cast(us2, odd ? (g0 << 16) : (g0 | (g1 << 16))), W1 = __builtin_bit_cast(us2, odd ? (g1 | (g2 << 16)) : (g2 | (g3 << 16)));
    const us2 W2 = __builtin_bit_cast(us2, odd ? (g3 | (g2 << 16)) : (g2 | (g1 << 16))), W3 = __builtin_bit_cast(us2, odd ? (g1 | (g0 << 16)) : g0);
    for (int r = ty; r < BTH; r += 8) {
        const int x = x0 + 4 * tx, y = y0 + r;
        if (x >= w || y >= h) continue;
        const uint32_t* hp = &s_h[(r >> 1) * BTW + tx];
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hp[i * 32]), W0, 1u << 15, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hp[BTW + i * 32]), W1, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hp[2 * BTW + i * 32]), W2, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hp[3 * BTW + i * 32]), W3, acc, false);
            o |= min(255u, acc >> 16) << (8 * i);
        }
        uint8_t* op = out + (size_t)y * pitch + x;
        if (x + 3 < w) *(uint32_t*)op = o;
        else for (int i = 0; x + i < w; ++i) op[i] = (uint8_t)(o >> (8 * i));
    }
}

__device__ __forceinline__ float fast_atan2_deg_dev(float y, float x) {
    const float k = 57.29577951308232f;
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + 2.220446e-16f); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + 2.220446e-16f); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wavefront per keypoint.  The chain of a keypoint is position -> 31 rows of the level image (intensity-centroid moments over the
// radius-15 disc, two 31-pixel rows per wave instruction) -> angle -> 512 steered samples of the blurred level -> 256 bits (4 x __ballot).
// The samples are where the time goes: 8 byte gathers per lane whose 64 addresses of one instruction spread over ~35 cache lines.  The
// pattern stays inside radius 13*sqrt(2) < 19 whatever the angle, so the wave first copies the 39 x 40-byte window of the blurred level
// around the keypoint into LDS with 7 row-contiguous dword loads (issued before the moments so both trips to L2 overlap) and gathers from
// there.  (Tried and dropped: several keypoints per wave in flight -- 2000 features x 32 frames, 1: 120 us, 4: 132 us.)
#define DP_R 19                     // window radius (rows/columns either side of the keypoint)
#define DP_ROWS (2 * DP_R + 1)
#define DP_DW 10                    // dwords per window row: columns x-19 .. x+20
#define DP_LOADS ((DP_ROWS * DP_DW + 63) / 64)
__global__ __launch_bounds__(256) void k_describe(DevPlan P, const SlotDesc* __restrict__ slots, const uint8_t* __restrict__ pyr,
                                                  const uint8_t* __restrict__ blurp, const uint32_t* __restrict__ sel,
                                                  const long long* __restrict__ sel_key, const int* __restrict__ sel_cnt,
                                                  vo_keypoint* __restrict__ kps, uint8_t* __restrict__ desc, int* __restrict__ nkp, int slot0, int n, int aff, int per_slot) {
    __shared__ uint32_t s_win[4][DP_LOADS * 64];
    int srel, jb;
    if (!vo_slot_block(per_slot, n, aff, srel, jb)) return;
    const int slot = slot0 + srel;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int cnt[VO_MAX_LEVELS], total = 0;
    for (int i = 0; i < P.L; ++i) { cnt[i] = sel_cnt[slot * VO_MAX_LEVELS + i]; total += cnt[i]; }
    if (jb == 0 && threadIdx.x == 0) nkp[slot] = total;
    // keypoint g of the quota-ordered list (level-major)
    const int g = jb * 4 + wave;
    int l = 0, base = 0;
    while (l + 1 < P.L && g >= P.qprefix[l + 1]) { base += cnt[l]; ++l; }
    if (g >= P.nfeat || (g - P.qprefix[l]) >= cnt[l]) return;          // wave-uniform; no workgroup barrier below
    const int outi = base + (g - P.qprefix[l]);
    const uint32_t c = sel[(size_t)slot * P.nfeat + g];
    const int x = c & 0xFFF, y = (c >> 12) & 0xFFF;
    const int pitch = P.pitch[l];
    const uint8_t* ctr = blurp + (size_t)slot * P.pyr_stride + P.loff[l] + (size_t)y * pitch + x;
    // the window of the blurred level, row-contiguous (keypoints keep >= 19 rows and 20 columns from the border: ORB's edge threshold is 31)
    uint32_t wreg[DP_LOADS];
#pragma unroll
    for (int i = 0; i < DP_LOADS; ++i) {
        const int e = min(i * 64 + lane, DP_ROWS * DP_DW - 1), r = e / DP_DW, q = e - r * DP_DW;      // (the last load's spare lanes repeat the last dword)
        wreg[i] = ((const U32u*)(ctr + (r - DP_R) * pitch - DP_R + 4 * q))->v;
    }
    // intensity centroid: lanes 0..30 take row v, lanes 32..62 row v+1; lane u is inside the disc on the rows |v| <= umax[|u|].  Every lane
    // loads on every row (lanes outside read the centre pixel and drop it) so the sixteen loads go out back to back, no branch between them
    int m10 = 0, m01 = 0;
    {
        const uint8_t* img = pyr + (size_t)slot * P.pyr_stride + P.loff[l] + (size_t)y * pitch + x;
        const int u = (lane & 31) - 15, half = lane >> 5;
        const int vlim = (lane & 31) < 31 ? (int)((P.umax_pk >> (4 * abs(u))) & 15) : -1;
#pragma unroll
        for (int v0 = -15; v0 <= 15; v0 += 2) {
            const int v = v0 + half;
            const bool in = abs(v) <= vlim;
            const int I = img[in ? v * pitch + u : 0];
            m10 += in ? u * I : 0; m01 += in ? v * I : 0;
        }
    }
    uint8_t* win = (uint8_t*)s_win[wave];
#pragma unroll
    for (int i = 0; i < DP_LOADS; ++i) s_win[wave][i * 64 + lane] = wreg[i];
    m10 = wave_sum_i32(m10); m01 = wave_sum_i32(m01);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double cs = 1.0, sn = 0.0;
    if (m10 != 0 || m01 != 0) {
        const double nrm = sqrt((double)m10 * (double)m10 + (double)m01 * (double)m01);
        cs = (double)m10 / nrm; sn = (double)m01 / nrm;
    }
    uint64_t bits[4];
    const uint8_t* wc = win + DP_R * (DP_DW * 4) + DP_R;                // the keypoint's own pixel inside the window
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int8_t* q = c_pattern[r * 64 + lane];
        const int x1 = __double2int_rn((double)q[0] * cs - (double)q[1] * sn), y1 = __double2int_rn((double)q[0] * sn + (double)q[1] * cs);
        const int x2 = __double2int_rn((double)q[2] * cs - (double)q[3] * sn), y2 = __double2int_rn((double)q[2] * sn + (double)q[3] * cs);
        bits[r] = __ballot(wc[y1 * (DP_DW * 4) + x1] < wc[y2 * (DP_DW * 4) + x2]);
    }
    if (lane == 0) {
        uint64_t* d = (uint64_t*)(desc + ((size_t)slot * P.nfeat + outi) * 32);
        d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
        vo_keypoint kp;
        kp.x = (float)x * P.scale[l];
        kp.y = (float)y * P.scale[l];
        kp.size = 31.f * P.scale[l];
        kp.angle = fast_atan2_deg_dev((float)m01, (float)m10);
        kp.response = (float)((double)sel_key[(size_t)slot * P.nfeat + g] * (1.0 / (25.0 * 7140.0 * 7140.0 * 7140.0 * 7140.0)));
        kp.octave = l; kp.class_id = -1;
        // Frame::GetDepth (reference src/frame.cpp:43-67), bounds-checked
        const SlotDesc sd = slots[slot];
        const int px = __float2int_rn(kp.x), py = __float2int_rn(kp.y);
        const int nx[5] = {0, -1, 0, 1, 0}, ny[5] = {0, 0, -1, 0, 1};
        int dr = 0;
        for (int i = 0; i < 5 && dr == 0; ++i) {
            const int x2 = px + nx[i], y2 = py + ny[i];
            if (x2 >= 0 && y2 >= 0 && x2 < P.W && y2 < P.H) dr = *(const uint16_t*)(sd.depth + (size_t)y2 * sd.depth_stride + 2 * (size_t)x2);
        }
        kp.depth_raw = dr;
        kps[(size_t)slot * P.nfeat + outi] = kp;
    }
}

// ------------------------------------------------------------------------------------------
// per-device function attributes: called from vo_ctx_create with the context's device current (cheap, idempotent)
int vo_orb_upload_constants() {
    HIP_TRY(hipFuncSetAttribute((const void*)k_select, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    return VO_OK;
}

int vo_orb_launch(vo_ctx* c, int slot0, int n) {
    const DevPlan& P = c->plan;
    hipStream_t st = c->stream;
    const int aff = n >= 8 ? 1 : 0;      // frame-to-XCD affinity (see the file header)
    HIP_TRY(hipMemsetAsync(c->d_cand_cnt + (size_t)slot0 * VO_MAX_LEVELS, 0, sizeof(int) * VO_MAX_LEVELS * n, st));
    { ProfScope ps(c, "k_gray");
      const int per = (((P.W + 15) / 16) * P.H + 255) / 256;
      hipLaunchKernelGGL(k_gray, dim3(vo_slot_grid(per, n, aff)), dim3(256), 0, st, P, c->d_slots, c->d_pyr, slot0, n, aff, per); }
    if (c->pyr_gx > 0) {
        ProfScope ps(c, "k_pyramid");
        hipLaunchKernelGGL(k_pyramid, dim3(vo_slot_grid(c->pyr_gx * c->pyr_gy, n, aff)), dim3(256), 0, st, P, c->d_pyr, c->d_tab, c->d_tabs, (const PyrRng*)c->d_pyr_rng, slot0, n, aff, c->pyr_gx, c->pyr_gy);
    } else for (int l = 1; l < P.L; ++l) {
        ProfScope ps(c, "k_resize");
        const int gx = (P.lw[l] + 63) / 64, gy = (P.lh[l] + 3) / 4;
        hipLaunchKernelGGL(k_resize, dim3(vo_slot_grid(gx * gy, n, aff)), dim3(64, 4), 0, st, P, l, c->d_pyr, c->d_tab, c->d_tabs, slot0, n, aff, gx, gy);
    }
    { ProfScope ps(c, "k_fast_nms");
      const int per = 8 * ((P.tile_prefix[P.L] + 7) / 8);
      hipLaunchKernelGGL(k_fast_nms, dim3(vo_slot_grid(per, n, aff)), dim3(64, 4), 0, st, P, c->d_pyr, c->d_cand, c->d_cand_cnt, c->d_status, slot0, n, aff, per); }
    { ProfScope ps(c, "k_select");
      size_t lds = (size_t)P.sel_cap * 12 + 4 * (256 + 8 + 4096);
      hipLaunchKernelGGL(k_select, dim3(vo_slot_grid(P.L, n, aff)), dim3(1024), lds, st, P, c->d_pyr, c->d_cand, c->d_cand_cnt, c->d_sel, c->d_sel_key, c->d_sel_cnt, slot0, n, aff); }
    { ProfScope ps(c, "k_blur");
      const int per = 8 * ((P.btile_prefix[P.L] + 7) / 8);
      hipLaunchKernelGGL(k_blur, dim3(vo_slot_grid(per, n, aff)), dim3(32, 8), 0, st, P, c->d_pyr, c->d_blur, slot0, n, aff, per); }
    { ProfScope ps(c, "k_describe");
      const int per = (P.nfeat + 3) / 4;
      hipLaunchKernelGGL(k_describe, dim3(vo_slot_grid(per, n, aff)), dim3(256), 0, st, P, c->d_slots, c->d_pyr, c->d_blur, c->d_sel, c->d_sel_key, c->d_sel_cnt, c->d_kps, c->d_desc, c->d_nkp, slot0, n, aff, per); }
    HIP_TRY(hipGetLastError());
    return VO_OK;
}
