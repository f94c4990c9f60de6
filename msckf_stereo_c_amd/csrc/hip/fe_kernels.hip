// fe_kernels.hip — hand-written gfx950 kernels of the stereo KLT front-end.
//
//  k_pyr_down3    : cg::pyr_down x 3        (reference call sites image_processor.cpp:239,242): levels 1 .. 3 of an image in one launch
//  k_detect_cells : cg::CornerDetector      (:132,259,657) — per-cell integer Shi-Tomasi maximum (32x32 px tiles)
//  k_lk_points4   : cg::optical_flow_multi_level (:410 temporal, :569 stereo) with the prediction (:321-350);
//  k_pt_geom      : the image-bounds gates (:416-424, :575-583), the stereo initial guess (:542-548),
//                   undistortion and the epipolar gate (:587-617), one thread per point.
//
// Arithmetic contract: every decision-bearing quantity is integer or a fixed
// sequence of IEEE-754 double operations; this file must be compiled with -ffp-contract=off.
// Work shapes: one 16-lane DPP row per tracked point (four points per wavefront, a lane owns one row of the 15x15
// window), one workgroup per 32x32 detector tile, one workgroup per 64x16 pyramid output tile; the
// stream index of the batch is blockIdx.y / blockIdx.z, so a launch covers every VIO stream of a
// context and fills the chip only when many streams are batched.
#include <stdlib.h>
#include <mutex>
#include "fe_device.h"

// ------------------------------------------------------------------------------------------ pyr_down
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { i = (i < 0) ? -i : 2 * (n - 1) - i; }
    return i;
}

// ---- levels 1, 2 and 3 of one image in ONE launch.
// Round 3 ran cg::pyr_down as one launch per level: level k was written to HBM and read back by the level k + 1 launch
// (1.46 x the algorithmic traffic), two of the three launches were tiny, and in the busy device each of the three dependent
// launches queued for compute units on its own (200 us per launch in situ against 55 alone).  Here a workgroup owns a
// 16 x 8 tile of LEVEL 3 and everything above it: it reads the level-0 footprint once (156 x 92 pixels for the 128 x 64 it
// owns; the halo is shared with the neighbouring tiles through the XCD's L2: the tiles of one image get block ids that are
// congruent modulo 8), keeps its level-1 region (76 x 44) and its level-2 region (36 x 20) in LDS, and writes the parts of
// the three levels it owns.  Every level is the separable [1 4 6 4 1] / 256 filter with BORDER_REFLECT_101 of the level
// BELOW it, so the values are those of three separate passes bit for bit: a region holds in-image pixels of its level, the
// two reflected columns / rows on either side of the image that the next level's taps can reach are filled in by copying
// (pd_fill_halo), and the level-0 taps are reflected at the loads.
// Planes: columns are shifted so that the columns a tile OWNS start on a group of four (stores and LDS rows stay dword
// aligned): a level-1 plane column pc is x1 = 4 * x3_0 - 8 + pc (80 columns, 76 used), a level-2 plane column qc is
// x2 = 2 * x3_0 - 4 + qc (40 columns, 36 used); byte planes carry 4 bytes of padding in front of every row because the
// next level's window of a group starts two pixels before a dword boundary.
#define P3_TW 16
#define P3_TH 8
struct Pyr3Job { const uint8_t *src; uint8_t *d1, *d2, *d3; int w0, h0; };

namespace {
constexpr int P3_R2H = 2 * P3_TH + 4, P3_R1H = 2 * P3_R2H + 4, P3_R0H = 2 * P3_R1H + 4;      // 20, 44, 92 rows
constexpr int P3_P2W = 2 * P3_TW + 8, P3_P1W = 2 * (2 * P3_TW + 4) + 8;                        // 40, 80 plane columns
constexpr int P3_P1S = P3_P1W + 8, P3_P2S = P3_P2W + 8;                                        // byte-plane row strides (4 B pad in front, 4 behind)
typedef uint32_t __attribute__((aligned(1))) pd_u32u;

// four neighbouring horizontal [1 4 6 4 1] sums from eleven consecutive pixels b0 .. b10 (three dwords): outputs at b2, b4, b6, b8
__device__ __forceinline__ uint2 pd_h4(uint32_t w0, uint32_t w1, uint32_t w2) {
    const int b0 = w0 & 255u, b1 = (w0 >> 8) & 255u, b2 = (w0 >> 16) & 255u, b3 = w0 >> 24, b4 = w1 & 255u, b5 = (w1 >> 8) & 255u,
              b6 = (w1 >> 16) & 255u, b7 = w1 >> 24, b8 = w2 & 255u, b9 = (w2 >> 8) & 255u, b10 = (w2 >> 16) & 255u;
    const uint32_t h0 = (uint32_t)(b0 + 4 * b1 + 6 * b2 + 4 * b3 + b4), h1 = (uint32_t)(b2 + 4 * b3 + 6 * b4 + 4 * b5 + b6);
    const uint32_t h2 = (uint32_t)(b4 + 4 * b5 + 6 * b6 + 4 * b7 + b8), h3 = (uint32_t)(b6 + 4 * b7 + 6 * b8 + 4 * b9 + b10);
    uint2 o; o.x = h0 | (h1 << 16); o.y = h2 | (h3 << 16);
    return o;
}
// four neighbouring output pixels from five rows of horizontal sums (two columns per register in 16-bit lanes:
// s + 128 <= 16 * 16 * 255 + 128 < 2^16, so the lanes never carry into each other)
__device__ __forceinline__ uint32_t pd_v4(uint2 r0, uint2 r1, uint2 r2, uint2 r3, uint2 r4) {
    const uint32_t lo = (r0.x + r4.x) + 4u * (r1.x + r3.x) + 6u * r2.x + 0x00800080u;
    const uint32_t hi = (r0.y + r4.y) + 4u * (r1.y + r3.y) + 6u * r2.y + 0x00800080u;
    return ((lo >> 8) & 255u) | ((lo >> 24) << 8) | (((hi >> 8) & 255u) << 16) | ((hi >> 24) << 24);
}
// store the (up to four) pixels of a group that lie inside the image row of width w
__device__ __forceinline__ void pd_store4(uint8_t *row, int x, int w, uint32_t px) {
    if (x + 3 < w) *reinterpret_cast<pd_u32u *>(row + x) = px;
    else for (int k = 0; x + k < w; ++k) row[x + k] = (uint8_t)(px >> (8 * k));
}
// BORDER_REFLECT_101 of the next level: the pixels at -2, -1, n, n + 1 of a level are copies of those at 2, 1, n - 2, n - 3.
// plane(row, col) addresses a byte plane whose (0, 0) is the pixel (y_base, x_base); rows x cols plane positions.
__device__ __forceinline__ void pd_fill_halo(uint8_t *plane, int stride, int rows, int cols, int y_base, int x_base, int w, int h, int tid) {
    // (a) the four out-of-image columns, every row the next level can reach; (b) the four out-of-image rows over the in-image columns
    for (int i = tid; i < 4 * rows; i += 256) {
        const int r = i >> 2, q = i & 3;
        const int x = q < 2 ? q - 2 : w + q - 2, y = y_base + r;
        const int c = x - x_base;
        if (c < 0 || c >= cols || y < -2 || y > h + 1) continue;
        const int ys = reflect101(y, h) - y_base, xs = reflect101(x, w) - x_base;
        if (ys < 0 || ys >= rows || xs < 0 || xs >= cols) continue;
        plane[r * stride + 4 + c] = plane[ys * stride + 4 + xs];
    }
    for (int i = tid; i < 4 * cols; i += 256) {
        const int q = i / cols, c = i - q * cols;
        const int y = q < 2 ? q - 2 : h + q - 2, x = x_base + c;
        const int r = y - y_base;
        if (r < 0 || r >= rows || x < 0 || x >= w) continue;
        const int ys = reflect101(y, h) - y_base;
        if (ys < 0 || ys >= rows) continue;
        plane[r * stride + 4 + c] = plane[ys * stride + 4 + c];
    }
}
}  // namespace

__global__ __launch_bounds__(256) void k_pyr_down3(const Pyr3Job *jobs, int n_jobs, int tiles_x, int tiles_y) {
    // block -> (image, tile): the tiles of ONE image get ids congruent modulo 8 (blocks b and b + 8 share an XCD, speed only)
    const int tiles = tiles_x * tiles_y;
    const int bx = blockIdx.x & 7, bq = blockIdx.x >> 3;
    const int ji = bx + 8 * (bq / tiles), tile = bq - (bq / tiles) * tiles;
    if (ji >= n_jobs) return;
    const Pyr3Job job = jobs[ji];
    const int w0 = job.w0, h0 = job.h0, w1 = (w0 + 1) >> 1, h1 = (h0 + 1) >> 1, w2 = (w1 + 1) >> 1, h2 = (h1 + 1) >> 1, w3 = (w2 + 1) >> 1, h3 = (h2 + 1) >> 1;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x3_0 = tx * P3_TW, y3_0 = ty * P3_TH;
    if (x3_0 >= w3 || y3_0 >= h3) return;
    __shared__ __attribute__((aligned(16))) uint16_t s_h0[P3_R0H][P3_P1W];     // horizontal sums of level 0 (rows: level-0 rows, columns: level-1 plane columns)
    __shared__ __attribute__((aligned(16))) uint8_t s_p1[P3_R1H][P3_P1S];      // level-1 region
    __shared__ __attribute__((aligned(16))) uint16_t s_h1[P3_R1H][P3_P2W];
    __shared__ __attribute__((aligned(16))) uint8_t s_p2[P3_R2H][P3_P2S];      // level-2 region
    __shared__ __attribute__((aligned(16))) uint16_t s_h2[P3_R2H][P3_TW];
    const int tid = threadIdx.x;
    const int x1b = 4 * x3_0 - 8, y1b = 4 * y3_0 - 6, y0b = 8 * y3_0 - 14;       // pixel of plane column / row 0
    const int x2b = 2 * x3_0 - 4, y2b = 2 * y3_0 - 2;
    // ---- A: horizontal pass over the level-0 rows this tile needs.  A group is four level-1 columns = eleven source pixels.
    //      Groups whose pixels all lie inside the image row take whole (unaligned) dwords; the groups at the left / right image
    //      border reflect every tap and run in a loop of their own (a wavefront does not pay both paths).
    {
        const int pa = max(0, -y0b), pb = min(P3_R0H, h0 - y0b);                 // plane rows inside the image
        constexpr int NG = P3_P1W / 4;
        // fast groups: x1f >= 1 (first tap 2 x1f - 2 >= 0), x1f + 3 < w1, last byte read 2 x1f + 9 < w0
        int ga = (1 - x1b + 3) >> 2; ga = ga < 0 ? 0 : ga;
        int gb = min(min((w1 - 4 - x1b) >> 2, (w0 - 10 - 2 * x1b) >> 3) + 1, NG);
        if (w1 - 4 - x1b < 0 || w0 - 10 - 2 * x1b < 0) gb = 0;
        if (gb < ga) gb = ga;
        const int ng = gb - ga, n_fast = (pb - pa) * ng;
        const unsigned inv = ng > 0 ? (65536u + (unsigned)ng - 1u) / (unsigned)ng : 0u;      // idx / ng for idx * ng < 2^16
        // every load of the tile is issued before the first sum is formed (a thread has at most MAXIT groups: 24 dwords in
        // flight per lane; with the loads inside the loop a workgroup paid the memory latency once per iteration: 145 us per
        // 192-stream launch alone against ... with them hoisted)
        constexpr int MAXIT = (P3_R0H * NG + 255) / 256;
        uint32_t la[MAXIT], lb[MAXIT], lc[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = tid + 256 * it;
            la[it] = lb[it] = lc[it] = 0u;
            if (idx < n_fast) {
                const int pr = (int)(((unsigned)idx * inv) >> 16), g = ga + idx - pr * ng, p = pa + pr;
                const uint8_t *q = job.src + (size_t)(y0b + p) * w0 + (2 * (x1b + 4 * g) - 2);
                la[it] = *reinterpret_cast<const pd_u32u *>(q); lb[it] = *reinterpret_cast<const pd_u32u *>(q + 4); lc[it] = *reinterpret_cast<const pd_u32u *>(q + 8);
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = tid + 256 * it;
            if (idx < n_fast) {
                const int pr = (int)(((unsigned)idx * inv) >> 16), g = ga + idx - pr * ng, p = pa + pr;
                *reinterpret_cast<uint2 *>(&s_h0[p][4 * g]) = pd_h4(la[it], lb[it], lc[it]);
            }
        }
        const int n_edge = NG - ng, n_slow = (pb - pa) * n_edge;
        for (int idx = tid; idx < n_slow; idx += 256) {
            const int pr = idx / n_edge, e = idx - pr * n_edge, g = e < ga ? e : gb + (e - ga), p = pa + pr;
            const uint8_t *row = job.src + (size_t)(y0b + p) * w0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x1 = x1b + 4 * g + j;
                if (x1 < 0 || x1 >= w1) continue;
                const int c = 2 * x1;
                s_h0[p][4 * g + j] = (uint16_t)(row[reflect101(c - 2, w0)] + 4 * row[reflect101(c - 1, w0)] + 6 * row[reflect101(c, w0)] +
                                                4 * row[reflect101(c + 1, w0)] + row[reflect101(c + 2, w0)]);
            }
        }
    }
    __syncthreads();
    // ---- B: vertical pass -> level 1 (in-image rows of the region), into the plane and, for the pixels this tile owns, to HBM
    {
        const int ra = max(0, -y1b), rb = min(P3_R1H, h1 - y1b);
        constexpr int NG = P3_P1W / 4;
        const int n = (rb - ra) * NG;
        for (int idx = tid; idx < n; idx += 256) {
            const int rr = idx / NG, g = idx - rr * NG, r = ra + rr;
            const int y1 = y1b + r, x1 = x1b + 4 * g;
            if (x1 + 3 < 0 || x1 >= w1) continue;
            uint2 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int p = min(max(reflect101(2 * y1 - 2 + k, h0) - y0b, 0), P3_R0H - 1);
                v[k] = *reinterpret_cast<const uint2 *>(&s_h0[p][4 * g]);
            }
            const uint32_t px = pd_v4(v[0], v[1], v[2], v[3], v[4]);
            *reinterpret_cast<uint32_t *>(&s_p1[r][4 + 4 * g]) = px;
            if (r >= 6 && r < 6 + 4 * P3_TH && g >= 2 && g < 2 + P3_TW && x1 >= 0) pd_store4(job.d1 + (size_t)y1 * w1, x1, w1, px);
        }
    }
    __syncthreads();
    const bool edge1 = x1b < 0 || y1b < 0 || x1b + P3_P1W > w1 || y1b + P3_R1H > h1;      // the region reaches past the level-1 image
    if (edge1) { pd_fill_halo(&s_p1[0][0], P3_P1S, P3_R1H, P3_P1W, y1b, x1b, w1, h1, tid); __syncthreads(); }
    // ---- C: horizontal pass over the level-1 region (rows -2 .. h1 + 1 of the image that the plane holds)
    {
        constexpr int NG = P3_P2W / 4;
        const int ra = max(0, -2 - y1b), rb = min(P3_R1H, h1 + 2 - y1b);
        const int n = (rb - ra) * NG;
        for (int idx = tid; idx < n; idx += 256) {
            const int rr = idx / NG, g = idx - rr * NG, r = ra + rr;
            // the group's eleven pixels start at plane column 8 g - 2: bytes 8 g + 2 .. of the padded row
            const uint32_t *q = reinterpret_cast<const uint32_t *>(&s_p1[r][8 * g]);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
            *reinterpret_cast<uint2 *>(&s_h1[r][4 * g]) = pd_h4(__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d2, d1, 2), __builtin_amdgcn_alignbyte(d3, d2, 2));
        }
    }
    __syncthreads();
    // ---- D: vertical pass -> level 2
    {
        constexpr int NG = P3_P2W / 4;
        const int ra = max(0, -y2b), rb = min(P3_R2H, h2 - y2b);
        const int n = (rb - ra) * NG;
        for (int idx = tid; idx < n; idx += 256) {
            const int rr = idx / NG, g = idx - rr * NG, r = ra + rr;
            const int y2 = y2b + r, x2 = x2b + 4 * g;
            if (x2 + 3 < 0 || x2 >= w2) continue;
            uint2 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = *reinterpret_cast<const uint2 *>(&s_h1[2 * r + k][4 * g]);      // level-1 row 2 y2 - 2 + k = plane row 2 r + k
            const uint32_t px = pd_v4(v[0], v[1], v[2], v[3], v[4]);
            *reinterpret_cast<uint32_t *>(&s_p2[r][4 + 4 * g]) = px;
            if (r >= 2 && r < 2 + 2 * P3_TH && g >= 1 && g < 1 + P3_TW / 2 && x2 >= 0) pd_store4(job.d2 + (size_t)y2 * w2, x2, w2, px);
        }
    }
    __syncthreads();
    const bool edge2 = x2b < 0 || y2b < 0 || x2b + P3_P2W > w2 || y2b + P3_R2H > h2;
    if (edge2) { pd_fill_halo(&s_p2[0][0], P3_P2S, P3_R2H, P3_P2W, y2b, x2b, w2, h2, tid); __syncthreads(); }
    // ---- E: horizontal pass over the level-2 region
    {
        constexpr int NG = P3_TW / 4;
        const int ra = max(0, -2 - y2b), rb = min(P3_R2H, h2 + 2 - y2b);
        const int n = (rb - ra) * NG;
        for (int idx = tid; idx < n; idx += 256) {
            const int rr = idx / NG, g = idx - rr * NG, r = ra + rr;
            // level-3 column x3_0 + 4 g: first tap at level-2 pixel 2 x3_0 + 8 g - 2 = plane column 8 g + 2
            const uint32_t *q = reinterpret_cast<const uint32_t *>(&s_p2[r][4 + 8 * g]);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
            *reinterpret_cast<uint2 *>(&s_h2[r][4 * g]) = pd_h4(__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d2, d1, 2), __builtin_amdgcn_alignbyte(d3, d2, 2));
        }
    }
    __syncthreads();
    // ---- F: vertical pass -> level 3 (every pixel of the tile is owned)
    {
        constexpr int NG = P3_TW / 4;
        const int rb = min(P3_TH, h3 - y3_0);
        const int n = rb * NG;
        for (int idx = tid; idx < n; idx += 256) {
            const int r = idx / NG, g = idx - r * NG;
            const int x3 = x3_0 + 4 * g;
            if (x3 >= w3) continue;
            uint2 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = *reinterpret_cast<const uint2 *>(&s_h2[2 * r + k][4 * g]);
            pd_store4(job.d3 + (size_t)(y3_0 + r) * w3, x3, w3, pd_v4(v[0], v[1], v[2], v[3], v[4]));
        }
    }
}

extern "C" void fe_launch_pyr_down3(const Pyr3Job *jobs_dev, int n_jobs, int max_w0, int max_h0, hipStream_t st) {
    const int w3 = (((max_w0 + 1) / 2 + 1) / 2 + 1) / 2, h3 = (((max_h0 + 1) / 2 + 1) / 2 + 1) / 2;
    const int tiles_x = (w3 + P3_TW - 1) / P3_TW, tiles_y = (h3 + P3_TH - 1) / P3_TH;
    hipLaunchKernelGGL(k_pyr_down3, dim3(8 * ((n_jobs + 7) / 8) * tiles_x * tiles_y), dim3(256), 0, st, jobs_dev, n_jobs, tiles_x, tiles_y);
}

// ------------------------------------------------------------------------------------------ detector
#define DET_BORDER 8
#define DS_LANES 14            // lanes of a 16-lane strip that produce outputs (the two on the right only feed their neighbours)
#define DS_COLS (4 * DS_LANES) // output columns of a strip

// Per-cell maxima are merged as 64-bit keys  gen (8 bits) | score (24 bits) | ~order (32 bits):
//   score  = integer Shi-Tomasi score (< 2^24: two sums of 64 squared differences of bytes),
//   order  = row-major position inside the cell, complemented, so ties go to the first pixel in scan order,
//   gen    = push generation of the context (1..255): a newer push always wins the atomicMax, so the key array is
//            never cleared between frames (only when the generation wraps or the array is reallocated).
__device__ __forceinline__ unsigned long long det_key(unsigned int gen, int score, unsigned int order) {
    return ((unsigned long long)gen << 56) | ((unsigned long long)(unsigned int)score << 32) | (0xFFFFFFFFu - order);
}

// floor(sqrt(v)) for 0 <= v < 2^48, exact: the hardware estimate is corrected with integer compares.
__device__ __forceinline__ unsigned int isqrt48(unsigned long long v, double vd) {
    unsigned int r = (unsigned int)__builtin_amdgcn_sqrt(vd);
    while ((unsigned long long)r * r > v) --r;
    while ((unsigned long long)(r + 1) * (r + 1) <= v) ++r;
    return r;
}

template <int CTRL> __device__ __forceinline__ int det_dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }   // lanes without a source read 0
template <int CTRL> __device__ __forceinline__ unsigned long long det_dpp0_64(unsigned long long v) {
    const unsigned int lo = (unsigned int)det_dpp0<CTRL>((int)(unsigned int)v), hi = (unsigned int)det_dpp0<CTRL>((int)(unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

typedef short det_v2s __attribute__((ext_vector_type(2)));
// a.lo * b.lo + a.hi * b.hi of two pairs of 16-bit lanes (three-operand form with an inline-constant addend: no move)
__device__ __forceinline__ int det_dot2(int a, int b) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(det_v2s, a), __builtin_bit_cast(det_v2s, b), 0, true); }
__device__ __forceinline__ int det_pksub(uint32_t a, uint32_t b) { return __builtin_bit_cast(int, (det_v2s)(__builtin_bit_cast(det_v2s, a) - __builtin_bit_cast(det_v2s, b))); }

// Horizontal 8-sums of one gradient product for the lane's four output columns.  The lane holds the products of its four
// columns as the pair sums P01 = p0 + p1, P23 = p2 + p3 and the single products p0, p2; lane r + 1 holds the next four
// columns, lane r + 2 the four after (row_shl DPP inside the 16-lane strip):
//   H0 = P01 + P23 + (the same of lane r+1),  H2 = H0 - P01 + P01 of lane r+2,  H1 = H0 - p0 + p0'',  H3 = H2 - p2 + p2''.
__device__ __forceinline__ void det_hsum(int P01, int P23, int p0, int p2, int H[4]) {
    const int S = P01 + P23;
    H[0] = S + det_dpp0<0x101>(S);
    H[2] = H[0] + (det_dpp0<0x102>(P01) - P01);
    H[1] = H[0] + (det_dpp0<0x102>(p0) - p0);
    H[3] = H[2] + (det_dpp0<0x102>(p2) - p2);
}

struct DetPix { uint32_t lo, hi; };     // eight pixels of an image row: columns c - 4 .. c - 1 and c .. c + 3 of the lane's first column c

// Gradient products dx*dx, dx*dy, dy*dy (central differences) of the lane's four columns in the row `mid`, and their
// horizontal 8-sums.  The differences are formed two at a time in 16-bit lanes (byte permutes + packed subtract), the
// products and their pair sums are 16-bit dot products: 64 instructions per row instead of 90 with 32-bit arithmetic.
__device__ __forceinline__ void det_hrow(uint32_t up, DetPix mid, uint32_t dn, int Ha[4], int Hb[4], int Hc[4]) {
    const uint32_t rn = (uint32_t)det_dpp0<0x101>((int)mid.hi);          // the pixels of lane r + 1 (its first one is needed)
    // pixels m0 .. m5 = columns c - 1 .. c + 4 as pairs of 16-bit lanes
    const uint32_t m01 = __builtin_amdgcn_perm(mid.hi, mid.lo, 0x0c040c03u);   // (lo.b3, hi.b0)
    const uint32_t m23 = __builtin_amdgcn_perm(0u, mid.hi, 0x0c020c01u);       // (hi.b1, hi.b2)
    const uint32_t m45 = __builtin_amdgcn_perm(rn, mid.hi, 0x0c040c03u);       // (hi.b3, rn.b0)
    const int D01 = det_pksub(m23, m01), D23 = det_pksub(m45, m23);            // dx of columns (0, 1) and (2, 3)
    const int E01 = det_pksub(__builtin_amdgcn_perm(0u, dn, 0x0c010c00u), __builtin_amdgcn_perm(0u, up, 0x0c010c00u));
    const int E23 = det_pksub(__builtin_amdgcn_perm(0u, dn, 0x0c030c02u), __builtin_amdgcn_perm(0u, up, 0x0c030c02u));
    const int D0 = D01 & 0xffff, D2 = D23 & 0xffff, E0 = E01 & 0xffff, E2 = E23 & 0xffff;   // (d, 0): picks the product of the even column
    det_hsum(det_dot2(D01, D01), det_dot2(D23, D23), det_dot2(D0, D01), det_dot2(D2, D23), Ha);
    det_hsum(det_dot2(D01, E01), det_dot2(D23, E23), det_dot2(D0, E01), det_dot2(D2, E23), Hb);
    det_hsum(det_dot2(E01, E01), det_dot2(E23, E23), det_dot2(E0, E01), det_dot2(E2, E23), Hc);
}

// cg::CornerDetector per-cell maxima of cam0 level 0 (image_processor.cpp:259, :657): integer Shi-Tomasi score
// (a + c) - isqrt((a - c)^2 + 4 b^2) of the 8x8 box sums a, b, c of the gradient products, maximum per detector cell with
// ties going to the first pixel in row-major order.
// Round 2 staged a 32x32 tile in LDS and ran three barrier-separated LDS passes over three product planes (24 LDS words
// per pixel, 124 VALU instructions per pixel, 304 us alone for 192 streams).  This version keeps everything in registers:
// a 16-lane DPP row is a STRIP of 64 columns (four per lane, 56 outputs) marched down a SEGMENT of rows.  Per row a lane
// loads eight bytes, forms its twelve products, the horizontal 8-sums come from the two lanes to the right (row_shl), and
// the vertical 8-sums are running sums: the row entering the window is added, the row leaving it is RECOMPUTED from its
// pixels (re-read through L1/L2) and subtracted - cheaper than a ring of eight rows of twelve sums in registers, and there
// is no LDS, no barrier and no inter-wave traffic at all.  The exact score test needs the square root only for a pixel
// that beats the best score its own column has seen in the current cell (disc < (a + c - best)^2 is decided first), so
// after the first rows of a cell almost no pixel takes the expensive path.  At the bottom of a cell row the lanes of a
// strip merge their keys with a segmented DPP max (runs of lanes in the same cell) and the first lane of each run issues
// the one atomicMax.
// Block -> (stream, four jobs): the jobs (strip, segment) of ONE stream get block ids congruent modulo 8, so an image
// travels through the L2 of one XCD (blocks b and b + 8 share an XCD).
__global__ __launch_bounds__(64) void k_detect_cells(const FeStreamDev *streams, int n_streams, int strips, int seg_rows, int waves, unsigned int gen) {
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int si = xcd + 8 * (qb / waves), wi = qb - (qb / waves) * waves;
    if (si >= n_streams) return;
    const FeStreamDev &S = streams[si];
    const int W = S.curr0.w[0], H = S.curr0.h[0];
    const int cw = S.cell_w, ch = S.cell_h, det_cols = S.det_cols, det_floor = S.det_floor;
    unsigned long long *const cell_keys = S.cell_keys;
    const int lane = threadIdx.x, r = lane & 15;
    const int job = 4 * wi + (lane >> 4);
    const int seg = job / strips, strip = job - seg * strips;
    const int xs = DET_BORDER + DS_COLS * strip;                 // first output column of the strip
    const int ys = DET_BORDER + seg_rows * seg;                  // first output row of the segment
    const int y_end = min(ys + seg_rows, H - DET_BORDER);
    const bool job_on = xs < W - DET_BORDER && ys < y_end;
    if (!__any(job_on)) return;
    // the lane's four product columns c0 .. c0 + 3 (outputs x = c0 + 4 + j); a lane right of the image reads clamped
    // addresses: its products only reach outputs that are not valid either
    const int c0 = xs - 4 + 4 * r;
    const int cl = min(c0, W - 4);
    const uint8_t *img = S.curr0.lvl[0];
    typedef uint32_t __attribute__((aligned(1))) u32u;
    auto load = [&](int row) -> DetPix {
        const uint8_t *p = img + (size_t)min(max(row, 0), H - 1) * W + cl;
        DetPix v; v.lo = *(const u32u *)(p - 4); v.hi = *(const u32u *)p; return v;
    };
    int xin[4], cx[4];
    bool col_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = c0 + 4 + j;
        cx[j] = x / cw; xin[j] = x - cx[j] * cw;
        col_ok[j] = job_on && r < DS_LANES && x < W - DET_BORDER;
    }
    int Va[4] = {0, 0, 0, 0}, Vb[4] = {0, 0, 0, 0}, Vc[4] = {0, 0, 0, 0};
    // ---- fill the window: product rows ys - 4 .. ys + 2
    int pr = ys - 4;
    // (the rows are loaded two steps before they are used: e_nx / l_nx are in flight while a step computes)
    DetPix e_mid = load(pr), e_dn = load(pr + 1), e_nx = load(pr + 2);
    uint32_t e_up = load(pr - 1).hi;
#pragma unroll 1
    for (int k = 0; k < 7; ++k) {
        const DetPix nx = load(pr + 3);
        int Ha[4], Hb[4], Hc[4];
        det_hrow(e_up, e_mid, e_dn.hi, Ha, Hb, Hc);
#pragma unroll
        for (int j = 0; j < 4; ++j) { Va[j] += Ha[j]; Vb[j] += Hb[j]; Vc[j] += Hc[j]; }
        e_up = e_mid.hi; e_mid = e_dn; e_dn = e_nx; e_nx = nx; ++pr;
    }
    // ---- steady state: the row pr enters, output row y = pr - 3, the row y - 4 leaves
    uint32_t l_up = load(ys - 5).hi;
    DetPix l_mid = load(ys - 4), l_dn = load(ys - 3), l_nx = load(ys - 2);
    int cy = ys / ch, yin = ys - cy * ch;
    int bs[4] = {0, 0, 0, 0};                                    // best score of the column in the current cell
    unsigned long long bk[4] = {0ULL, 0ULL, 0ULL, 0ULL};
    const int n_rows = seg_rows;                                 // (uniform trip count; rows past y_end are masked)
#pragma unroll 1
    for (int k = 0; k < n_rows; ++k) {
        const int y = ys + k;
        const bool row_ok = y < y_end;
        const DetPix enx = load(pr + 3), lnx = load(y - 1);
        {
            int Ha[4], Hb[4], Hc[4];
            det_hrow(e_up, e_mid, e_dn.hi, Ha, Hb, Hc);
#pragma unroll
            for (int j = 0; j < 4; ++j) { Va[j] += Ha[j]; Vb[j] += Hb[j]; Vc[j] += Hc[j]; }
            e_up = e_mid.hi; e_mid = e_dn; e_dn = e_nx; e_nx = enx; ++pr;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(col_ok[j] && row_ok)) continue;
            const int a = Va[j], b = Vb[j], d = Vc[j];
            // score > T  <=>  isqrt(disc) < X := a + d - T  <=>  disc < X^2 (X > 0): decided exactly (everything is below
            // 2^50) before the square root.  T = the detection floor or the best score of this column in this cell: a
            // later pixel of the column only replaces the best with a strictly larger score (ties go to the first).
            const int X = (a + d) - max(det_floor, bs[j]);
            if (X <= 0) continue;
            const int df = a - d;
            if (abs(df) >= X || 2 * abs(b) >= X) continue;       // each term of disc alone already reaches X^2
            const double discd = (double)df * (double)df + 4.0 * ((double)b * (double)b);   // exact: < 2^48
            if (discd >= (double)X * (double)X) continue;
            const long long disc = (long long)df * df + 4LL * ((long long)b * b);
            const int score = (a + d) - (int)isqrt48((unsigned long long)disc, discd);
            bs[j] = score;
            bk[j] = det_key(gen, score, (unsigned int)(yin * cw + xin[j]));
        }
        {
            int Ha[4], Hb[4], Hc[4];
            det_hrow(l_up, l_mid, l_dn.hi, Ha, Hb, Hc);
#pragma unroll
            for (int j = 0; j < 4; ++j) { Va[j] -= Ha[j]; Vb[j] -= Hb[j]; Vc[j] -= Hc[j]; }
            l_up = l_mid.hi; l_mid = l_dn; l_dn = l_nx; l_nx = lnx;
        }
        // ---- bottom of a cell row (or of the segment): merge the strip's keys per cell, one atomicMax per cell
        ++yin;
        const bool flush = job_on && row_ok && (yin == ch || y + 1 == y_end);
        if (cw < 4) {
            // cells narrower than a lane's four columns (images below 188 pixels with the 47-column detector grid): a lane may
            // touch three cells, so every column sends its own key
            if (flush) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (bk[j]) atomicMax(&cell_keys[cy * det_cols + cx[j]], bk[j]);
                    bs[j] = 0; bk[j] = 0ULL;
                }
            }
        } else if (__any(flush)) {
            // a lane's four columns lie in at most two cells (cells are at least four pixels wide): A = the cell of its
            // first column, B = the cell of its last one when that differs
            unsigned long long ka = 0ULL, kb = 0ULL;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long v = flush ? bk[j] : 0ULL;
                if (cx[j] == cx[0]) ka = v > ka ? v : ka; else kb = v > kb ? v : kb;
            }
            const bool two = cx[3] != cx[0];
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                unsigned long long kk = pass ? kb : ka;
                const int id = flush ? (pass ? (two ? cx[3] : -1 - r) : cx[0]) : -1 - r;     // (negative ids never match a neighbour)
                if (pass && !__any(flush && two)) break;
                // segmented max over runs of lanes with the same cell id: after the steps 1, 2, 4, 8 the first lane of a
                // run holds the maximum of the run
                {
                    const int o = det_dpp0<0x101>(id + 0x40000000) - 0x40000000; const unsigned long long v = det_dpp0_64<0x101>(kk);
                    if (o == id && v > kk) kk = v;
                }
                {
                    const int o = det_dpp0<0x102>(id + 0x40000000) - 0x40000000; const unsigned long long v = det_dpp0_64<0x102>(kk);
                    if (o == id && v > kk) kk = v;
                }
                {
                    const int o = det_dpp0<0x104>(id + 0x40000000) - 0x40000000; const unsigned long long v = det_dpp0_64<0x104>(kk);
                    if (o == id && v > kk) kk = v;
                }
                {
                    const int o = det_dpp0<0x108>(id + 0x40000000) - 0x40000000; const unsigned long long v = det_dpp0_64<0x108>(kk);
                    if (o == id && v > kk) kk = v;
                }
                const int left = det_dpp0<0x111>(id + 0x40000000) - 0x40000000;          // row_shr:1: the lane to the left (none: -2^30)
                if (id >= 0 && left != id && kk) atomicMax(&cell_keys[cy * det_cols + id], kk);
            }
            if (flush) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { bs[j] = 0; bk[j] = 0ULL; }
            }
        }
        if (yin == ch) { yin = 0; ++cy; }
    }
}

extern "C" void fe_launch_detect(const FeStreamDev *streams_dev, int n_streams, int max_w, int max_h, unsigned int gen, hipStream_t st) {
    // rows per segment: short segments give more wavefronts (a launch should fill the device several times over) but every
    // segment pays seven window-filling rows
    const int strips = (max_w - 2 * DET_BORDER + DS_COLS - 1) / DS_COLS, rows = max_h - 2 * DET_BORDER;
    if (strips <= 0 || rows <= 0) return;
    int seg_rows = 32;
    while (seg_rows < 128 && (long long)n_streams * strips * ((rows + seg_rows - 1) / seg_rows) / 4 > 32768) seg_rows *= 2;
    const int segs = (rows + seg_rows - 1) / seg_rows;
    const int waves = (strips * segs + 3) / 4;
    hipLaunchKernelGGL(k_detect_cells, dim3(8 * ((n_streams + 7) / 8) * waves), dim3(64), 0, st, streams_dev, n_streams, strips, seg_rows, waves, gen);
}

// ------------------------------------------------------------------------------------------ point math
// Deterministic atan / tan for the equidistant (fisheye) camera model.  cv::fisheye::distortPoints needs atan(r),
// cv::fisheye::undistortPoints needs tan(theta); libm and the GPU math library do not round them identically, so both
// sides evaluate the SAME sequence of IEEE double operations (+ - * / only, no contraction): argument reduction to a
// small interval and an odd Taylor polynomial.  Accuracy ~2e-16 relative (checked against libm in the tests).

// atan(x), x >= 0:  x > 1 -> pi/2 - atan(1/x);  then atan(x) = atan(c) + atan((x - c) / (1 + x c)) with the nearest
// c in {0, 1/4, 1/2, 3/4, 1} (|z| <= 1/8 -> 11 odd terms leave < 1e-21)
__device__ __forceinline__ double det_atan(double x) {
    const bool inv = x > 1.0;
    if (inv) x = 1.0 / x;
    double c = 0.0, ac = 0.0;
    if (x > 0.875)      { c = 1.0;  ac = 0.78539816339744830962; }
    else if (x > 0.625) { c = 0.75; ac = 0.64350110879328438680; }
    else if (x > 0.375) { c = 0.5;  ac = 0.46364760900080611621; }
    else if (x > 0.125) { c = 0.25; ac = 0.24497866312686415417; }
    const double z = (x - c) / (1.0 + x * c);
    const double z2 = z * z;
    double p = 1.0 / 21.0;
    p = 1.0 / 19.0 - z2 * p;
    p = 1.0 / 17.0 - z2 * p;
    p = 1.0 / 15.0 - z2 * p;
    p = 1.0 / 13.0 - z2 * p;
    p = 1.0 / 11.0 - z2 * p;
    p = 1.0 / 9.0 - z2 * p;
    p = 1.0 / 7.0 - z2 * p;
    p = 1.0 / 5.0 - z2 * p;
    p = 1.0 / 3.0 - z2 * p;
    p = 1.0 - z2 * p;
    const double a = ac + z * p;
    return inv ? 1.5707963267948966192 - a : a;
}
// tan(t), 0 <= t <= pi/2:  t > pi/4 -> 1 / tan(pi/2 - t);  tan = sin / cos with Taylor series on [0, pi/4]
__device__ __forceinline__ double det_tan(double t) {
    const bool inv = t > 0.78539816339744830962;
    if (inv) t = 1.5707963267948966192 - t;
    const double t2 = t * t;
    double s = 1.0 / 121645100408832000.0;              // 1/19!
    s = 1.0 / 355687428096000.0 - t2 * s;                // 1/17!
    s = 1.0 / 1307674368000.0 - t2 * s;                  // 1/15!
    s = 1.0 / 6227020800.0 - t2 * s;                     // 1/13!
    s = 1.0 / 39916800.0 - t2 * s;                       // 1/11!
    s = 1.0 / 362880.0 - t2 * s;                         // 1/9!
    s = 1.0 / 5040.0 - t2 * s;                           // 1/7!
    s = 1.0 / 120.0 - t2 * s;                            // 1/5!
    s = 1.0 / 6.0 - t2 * s;                              // 1/3!
    s = t * (1.0 - t2 * s);
    double c = 1.0 / 6402373705728000.0;                 // 1/18!
    c = 1.0 / 20922789888000.0 - t2 * c;                 // 1/16!
    c = 1.0 / 87178291200.0 - t2 * c;                    // 1/14!
    c = 1.0 / 479001600.0 - t2 * c;                      // 1/12!
    c = 1.0 / 3628800.0 - t2 * c;                        // 1/10!
    c = 1.0 / 40320.0 - t2 * c;                          // 1/8!
    c = 1.0 / 720.0 - t2 * c;                            // 1/6!
    c = 1.0 / 24.0 - t2 * c;                             // 1/4!
    c = 0.5 - t2 * c;                                    // 1/2!
    c = 1.0 - t2 * c;
    return inv ? c / s : s / c;
}


// cv::undistortPoints (radtan: 5 fixed-point iterations; equidistant: cv::fisheye::undistortPoints, Newton on theta,
// <= 10 iterations), then R, then P = {1,1,0,0}.  Same operation sequence as oracle/o_image.cpp undistort_point.
__device__ __forceinline__ void undistort_pt(const CamDev &cam, const double *R, float u, float v, float &xo, float &yo) {
    const double fx = cam.K[0], fy = cam.K[1], cx = cam.K[2], cy = cam.K[3];
    double x = ((double)u - cx) / fx, y = ((double)v - cy) / fy;
    if (cam.model == MSKF_MODEL_EQUIDISTANT) {
        const double *k = cam.D;
        double theta_d = sqrt(x * x + y * y);
        const double half_pi = 1.5707963267948966;
        theta_d = theta_d > half_pi ? half_pi : (theta_d < -half_pi ? -half_pi : theta_d);
        double scale = 0.0;
        if (theta_d > 1e-8) {
            double theta = theta_d;
            for (int j = 0; j < 10; ++j) {
                const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
                const double k0t2 = k[0] * t2, k1t4 = k[1] * t4, k2t6 = k[2] * t6, k3t8 = k[3] * t8;
                const double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) /
                                   (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
                theta = theta - fix;
                if (fabs(fix) < 1e-8) break;
            }
            scale = det_tan(theta) / theta_d;
        }
        x = x * scale; y = y * scale;
    } else {
        const double k1 = cam.D[0], k2 = cam.D[1], p1 = cam.D[2], p2 = cam.D[3];
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2);
            const double deltaX = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
            const double deltaY = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
    }
    if (R) {
        const double X = R[0] * x + R[1] * y + R[2];
        const double Y = R[3] * x + R[4] * y + R[5];
        const double Wd = R[6] * x + R[7] * y + R[8];
        x = X / Wd; y = Y / Wd;
    } else {
        // identity rectification: X = 1*x + 0*y + 0 etc. are exact, W = 1
        x = x / 1.0; y = y / 1.0;
    }
    xo = (float)(x * 1.0 + 0.0);
    yo = (float)(y * 1.0 + 0.0);
}

// cg::project_points with zero rvec / tvec: (x, y) is the ray (x, y, 1); oracle/o_image.cpp distort_point
__device__ __forceinline__ void distort_pt(const CamDev &cam, float xf, float yf, float &uo, float &vo) {
    const double fx = cam.K[0], fy = cam.K[1], cx = cam.K[2], cy = cam.K[3];
    const double x = (double)xf, y = (double)yf;
    double xd, yd;
    if (cam.model == MSKF_MODEL_EQUIDISTANT) {
        const double *k = cam.D;
        const double r = sqrt(x * x + y * y);
        const double theta = det_atan(r);
        const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const double theta_d = theta * (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8);
        const double scale = (r > 1e-8) ? theta_d / r : 1.0;
        xd = x * scale; yd = y * scale;
    } else {
        const double k1 = cam.D[0], k2 = cam.D[1], p1 = cam.D[2], p2 = cam.D[3];
        const double r2 = x * x + y * y, r4 = r2 * r2;
        const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
        const double cdist = 1.0 + k1 * r2 + k2 * r4;
        xd = x * cdist + p1 * a1 + p2 * a2;
        yd = y * cdist + p1 * a3 + p2 * a1;
    }
    uo = (float)(xd * fx + cx);
    vo = (float)(yd * fy + cy);
}

// ------------------------------------------------------------------------------------------ LK
#define LK_HALF 7
#define LK_WIN 15
#define LK_ITERS 30

// ---- DPP butterfly step of the row reductions (l4_row_sum).  All lanes of the row must be active.
template <int CTRL> __device__ __forceinline__ int dpp_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// (double)b * 2^-20, exactly, for |b| < 2^51, in ONE double-precision instruction: b is added (scalar unit) to the bit
// pattern of 1.5 * 2^32, whose mantissa LSB weighs 2^-20; subtracting 1.5 * 2^32 leaves b * 2^-20.  The plain form
// (int64 -> double conversion, then the scaling) is five double-rate instructions per value, and FP64 issues at half
// rate: these uniform conversions were a tenth of the kernel.
__device__ __forceinline__ double lk_scaled_f64(long long b) {
    return __longlong_as_double(b + 0x41F8000000000000LL) - 6442450944.0;
}

// ------------------------------------------------------------------------------------------ LK, four points per wavefront
// k_lk_points4: pyramidal LK (OpenCV calcOpticalFlowPyrLK with OPTFLOW_USE_INITIAL_FLOW semantics, fixed point: every
// decision-bearing quantity is an integer or a fixed sequence of double operations) with a 16-lane row of the
// wavefront per point: lane r of a row owns image row r of the 16 x 16 search footprint, i.e. ALL 15 pixels of window
// row r.  What that buys over one wavefront per point (round 1: 2840 VALU instructions per point, now 1950):
//   * the per-point scalar work of an iteration (weights, 2 x 2 solve, convergence tests) is issued once for four points;
//   * the cross-lane reduction stays inside a DPP row (no cross-row step, no readlane);
//   * a lane walks 16 consecutive bytes, so the bilinear sample is two packed 16-bit dot products
//     (v_dot2_i32_i16 on byte pairs built with v_perm_b32) per row instead of four multiplies per pixel, the row below
//     comes from the neighbouring lane through a DPP row shift fused into the add, and the b1 / b2 sums are packed dot
//     products too (|diff| <= 8160 and |I| <= 4080 fit 16 bits).
// Iterations run in lock step for the four points of a wave; a point that has converged idles until the others have.
typedef short l4_v2s __attribute__((ext_vector_type(2)));
#define L4_DW 6                // staged dwords per row (24 bytes: the 16 a window reads + alignment + slack)
#define L4_TROWS 18            // template source rows: 17 interpolated rows need 18
#define L4_SROWS 20            // staged search rows: the 16 a window reads + 4 of slack
#define L4_MAXOX 8             // window column offset inside the staged rows: 0 .. 8
#define L4_MAXOY (L4_SROWS - 16)

typedef unsigned short l4_v2u __attribute__((ext_vector_type(2)));
// a * b + c on the 24-bit multiplier (full rate; exact while |a|, |b| < 2^23).  The compiler lowers the C expression to
// separate multiplies and a three-operand add, or to the quarter-rate 32 x 32 -> 64 bit multiply-add.
template <int KA, int KC> __device__ __forceinline__ int l4_mad24_k(int x) {        // KA * x + KC, inline constants
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "n"(KA), "v"(x), "n"(KC));
    return d;
}
template <int KA> __device__ __forceinline__ int l4_mad24_kv(int x, int c) {        // KA * x + c
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "n"(KA), "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ int l4_mad24(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ int l4_mad24_vvv(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int l4_dot2(int pair, int w, int acc) {          // running sums: acc is the destination (v_dot2c)
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(l4_v2s, pair), __builtin_bit_cast(l4_v2s, w), acc, false);
}
// Same product with a CONSTANT addend.  The two-operand form above needs the addend in the destination register, i.e. a
// v_mov per call when it is a constant (47 of the 270 instructions of an LK iteration were such moves); with the clamp
// bit set the compiler has to take the three-operand encoding, whose addend is an inline constant or a scalar register.
// Saturation never triggers here (|sums| < 2^23), so the value is the same.
__device__ __forceinline__ int l4_dot2k(int pair, int w, int k) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(l4_v2s, pair), __builtin_bit_cast(l4_v2s, w), k, true);
}
// ((a >> 9), (b >> 9)) as two 16-bit lanes for 0 <= a, b < 2^24: one byte permute takes bits 8..23 of both, one packed
// shift drops the ninth bit (three instructions fewer per pair than shift, shift, pack)
__device__ __forceinline__ int l4_pack_shr9(int a, int b) {
    const l4_v2u v = __builtin_bit_cast(l4_v2u, __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x06050201u));
    return __builtin_bit_cast(int, (l4_v2u)(v >> (unsigned short)1));
}
// DPP row operations (a row = the 16 lanes of one point).  row_shl:n: lane i reads lane i + n, row_shr:n: lane i - n.
#define L4_SHL1 0x101
#define L4_SHR1 0x111
template <int CTRL> __device__ __forceinline__ int l4_dpp0(int v) {        // lanes without a source read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// 16-lane (row) sum of per-lane int32 partials whose total needs more than 32 bits: quad sums in 32 bits (four lanes of
// |v| < 2^28.9 stay below 2^31), the last two butterflies in 64 bits.  Every lane of the row ends up with the sum.
__device__ __forceinline__ long long l4_row_sum(int v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    long long s = (long long)v;
    {
        const int lo = __builtin_amdgcn_update_dpp(0, (int)s, 0x141, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, (int)(s >> 32), 0x141, 0xf, 0xf, true);
        s += (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);     // row_half_mirror
    }
    {
        const int lo = __builtin_amdgcn_update_dpp(0, (int)s, 0x140, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, (int)(s >> 32), 0x140, 0xf, 0xf, true);
        s += (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);     // row_mirror
    }
    return s;
}

// Stage ROWS x 24 bytes at image (x0, y0) as LDS dwords dst[row * 6 + c]; the 16 lanes of a point cooperate (r = lane & 15).
// Pixels outside the image are replicated from the border.
struct __attribute__((packed, aligned(4))) l4_u4 { uint32_t v[4]; };
struct __attribute__((packed, aligned(4))) l4_u2 { uint32_t v[2]; };
template <int ROWS>
__device__ __forceinline__ void l4_stage(const uint8_t *img, int w, int h, int x0, int y0, uint32_t *dst, int r, bool on) {
    const bool fast = x0 >= 0 && y0 >= 0 && x0 + 4 * L4_DW <= w && y0 + ROWS <= h;
    if (fast) {
        // inside the image: lane r takes row r whole (24 bytes = a 16-byte and an 8-byte load at a dword-aligned address),
        // lanes 0 .. ROWS-17 take rows 16 .. ROWS-1 as well: four loads with one address each instead of seven or eight
        // dword loads with an element -> (row, column) division each
        if (on) {
            const uint8_t *p = img + l4_mad24_vvv(y0 + r, w, x0);
            const l4_u4 a = *(const l4_u4 *)p;
            const l4_u2 b = *(const l4_u2 *)(p + 16);
            uint32_t *d = dst + L4_DW * r;
            d[0] = a.v[0]; d[1] = a.v[1]; d[2] = a.v[2]; d[3] = a.v[3]; d[4] = b.v[0]; d[5] = b.v[1];
            if (r < ROWS - 16) {
                const uint8_t *p2 = img + l4_mad24_vvv(y0 + 16 + r, w, x0);
                const l4_u4 a2 = *(const l4_u4 *)p2;
                const l4_u2 b2 = *(const l4_u2 *)(p2 + 16);
                uint32_t *d2 = dst + L4_DW * (16 + r);
                d2[0] = a2.v[0]; d2[1] = a2.v[1]; d2[2] = a2.v[2]; d2[3] = a2.v[3]; d2[4] = b2.v[0]; d2[5] = b2.v[1];
            }
        }
        return;
    }
    // at the border: pixels outside the image are replicated from the edge.  x0 is a multiple of four, so a staged dword lies
    // inside its row, or left of it (four copies of the first pixel), or right of it / across its end (the row's last four
    // pixels shifted down, the last pixel repeated): ONE dword load at a clamped position per element and a byte alignment.
    // (Round 2 loaded four clamped bytes per element: 230 instructions per call, and at the two coarse levels almost every
    // wavefront has a point near the border - a quarter of all LK instructions were this path.)
    constexpr int N = ROWS * L4_DW, PER = (N + 15) / 16;
    typedef uint32_t __attribute__((aligned(1))) l4_u32u;
    uint32_t v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = r + 16 * i;
        v[i] = 0;
        if (on && e < N) {
            const int row = (e * 43) >> 8, c = e - L4_DW * row;       // e / 6 for e < 128
            const uint8_t *rp = img + l4_mad24_vvv(min(max(y0 + row, 0), h - 1), w, 0);
            const int xb = x0 + 4 * c, xc = min(max(xb, 0), w - 4), sh = xb - xc;
            const uint32_t D = *(const l4_u32u *)(rp + xc);
            const uint32_t first = (D & 255u) * 0x01010101u, last = (D >> 24) * 0x01010101u;
            v[i] = sh < 0 ? first : sh >= 4 ? last : __builtin_amdgcn_alignbyte(last, D, (uint32_t)sh);
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int e = r + 16 * i; if (on && e < N) dst[e] = v[i]; }
}

// byte pair (b_k, b_k+1) of a byte string held in aligned dwords, as two 16-bit lanes
template <int K> __device__ __forceinline__ int l4_pair(const uint32_t *a) {
    constexpr int q = K >> 2, m = K & 3;
    if (m == 0) return (int)__builtin_amdgcn_perm(0u, a[q], 0x0c010c00u);
    if (m == 1) return (int)__builtin_amdgcn_perm(0u, a[q], 0x0c020c01u);
    if (m == 2) return (int)__builtin_amdgcn_perm(0u, a[q], 0x0c030c02u);
    return (int)__builtin_amdgcn_perm(a[q + 1], a[q], 0x0c040c03u);
}
template <int N, int K = 0> struct L4Pairs {
    static __device__ __forceinline__ void run(const uint32_t *a, int *pr) { pr[K] = l4_pair<K>(a); L4Pairs<N, K + 1>::run(a, pr); }
};
template <int N> struct L4Pairs<N, N> { static __device__ __forceinline__ void run(const uint32_t *, int *) {} };

__device__ __forceinline__ void l4_weights(float fa, float fb, int &wtop, int &wbot) {
    const int qa = __float2int_rn(fa * 16384.0f), qb = __float2int_rn(fb * 16384.0f);
    // (operands <= 2^14: 24-bit multiply-adds; the 32 x 32 -> 64 bit form the compiler picks otherwise is quarter rate)
    const int w00 = l4_mad24(16384 - qa, 16384 - qb, 8192) >> 14;
    const int w01 = l4_mad24(qa, 16384 - qb, 8192) >> 14;
    const int w10 = l4_mad24(16384 - qa, qb, 8192) >> 14;
    const int w11 = 16384 - w00 - w01 - w10;                    // may be -1: kept signed in its 16-bit lane
    wtop = (w00 & 0xffff) | (w01 << 16);
    wbot = (w10 & 0xffff) | (w11 << 16);
}

// Pyramidal LK of the four points of a wavefront (see above): templates around (ax, ay) in pyramid A, search in pyramid B
// from the initial guess (bx, by), both in level-0 pixels.  Every lane of the wavefront calls it (it stages through LDS and
// uses the workgroup barrier of the one-wave workgroup); `alive` marks the 16-lane rows that carry a point.  Returns the
// tracked position and the OpenCV status (0: lost).
__device__ __forceinline__ void l4_track(const PyrDev &A, const PyrDev &B, bool alive, float ax, float ay, float bx, float by,
                                         uint32_t *sT, uint32_t *sS, int r, float &out_x, float &out_y, int &out_status, unsigned int &dbg_iters) {
    int status = 1;
    float ncx = 0.f, ncy = 0.f;
    const bool winrow = r < LK_WIN;                       // lane 15 only feeds the row below window row 14
    for (int l = MSKF_LEVELS - 1; l >= 0; --l) {
        const uint8_t *imA = A.lvl[l];
        const uint8_t *imB = B.lvl[l];
        const int aw = A.w[l], ah = A.h[l], bw = B.w[l], bh = B.h[l];
        const float sc = __int_as_float((127 - l) << 23);        // 2^-l exactly
        const float pwx = ax * sc - (float)LK_HALF, pwy = ay * sc - (float)LK_HALF;
        if (l == MSKF_LEVELS - 1) { ncx = bx * sc; ncy = by * sc; }
        else { ncx = ncx * 2.0f; ncy = ncy * 2.0f; }
        int ipx = (int)floorf(pwx), ipy = (int)floorf(pwy);
        bool lvl_on = alive && !(ipx < -LK_WIN || ipx >= aw || ipy < -LK_WIN || ipy >= ah);
        if (alive && !lvl_on && l == 0) status = 0;
        if (!lvl_on) { ipx = 0; ipy = 0; }                   // idle slot: any in-range position
        int wtop, wbot;
        l4_weights(lvl_on ? pwx - (float)ipx : 0.f, lvl_on ? pwy - (float)ipy : 0.f, wtop, wbot);
        // ---- stage the template source (18 x 24 B at the window's 4-aligned left edge) and the first search region
        const int ax0 = (ipx - 1) & ~3, ay0 = ipy - 1, oxa = ipx - 1 - ax0;
        float wx = ncx - (float)LK_HALF, wy = ncy - (float)LK_HALF;
        int inx0 = (int)floorf(wx), iny0 = (int)floorf(wy);
        const bool in_b0 = !(inx0 < -LK_WIN || inx0 >= bw || iny0 < -LK_WIN || iny0 >= bh);
        if (!(lvl_on && in_b0)) { inx0 = 0; iny0 = 0; }
        int bx0 = (inx0 - 2) & ~3, by0 = iny0 - 2;
        __syncthreads();
        l4_stage<L4_TROWS>(imA, aw, ah, ax0, ay0, sT, r, true);
        l4_stage<L4_SROWS>(imB, bw, bh, bx0, by0, sS, r, true);
        __syncthreads();
        // ---- interpolated template: lane r holds source row r + 1, so V = T(row r+1) + B(row r+2) is template row r + 1;
        //      rows 0 and 16 are the extra value E of lanes 0 and 15 (source rows 0 / 17 with the top / bottom weights)
        int V[17], E[17];
        {
            uint32_t d[6], a[5];
            int pr[17];
            const uint32_t *rp = sT + (r + 1) * L4_DW;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = rp[i];
#pragma unroll
            for (int i = 0; i < 5; ++i) a[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], oxa);
            L4Pairs<17>::run(a, pr);
            int Bv[17];
#pragma unroll
            for (int c = 0; c < 17; ++c) { V[c] = l4_dot2k(pr[c], wtop, 256); Bv[c] = l4_dot2k(pr[c], wbot, 0); }
            // extra row: source row 0 for lane 0 (top weights, added to its own B), source row 17 for lane 15 (bottom
            // weights, added to its own T)
            const uint32_t *ep = sT + (r == 0 ? 0 : 17) * L4_DW;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = ep[i];
#pragma unroll
            for (int i = 0; i < 5; ++i) a[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], oxa);
            L4Pairs<17>::run(a, pr);
            const int wsel = r == 0 ? wtop : wbot;
#pragma unroll
            for (int c = 0; c < 17; ++c) E[c] = l4_dot2(pr[c], wsel, r == 0 ? Bv[c] + 256 : V[c]) >> 9;
#pragma unroll
            for (int c = 0; c < 17; ++c) V[c] = (V[c] + l4_dpp0<L4_SHL1>(Bv[c])) >> 9;
            // lane 15 is no window row: it hands template row 16 to lane 14 as "the row below"
            if (r == 15) {
#pragma unroll
                for (int c = 0; c < 17; ++c) V[c] = E[c];
            }
        }
        // ---- Scharr gradients of window row r: rows above / below from the neighbouring lanes
        int Pp[8], Ixp[8], Iyp[8];       // packed pairs (column 2m, 2m + 1); column 15 is padding (zero gradient)
        int A11 = 0, A12 = 0, A22 = 0;
        {
            int f[17], e[17];
#pragma unroll
            for (int c = 0; c < 17; ++c) {
                const int up = __builtin_amdgcn_update_dpp(E[c], V[c], L4_SHR1, 0xf, 0xf, false);   // lane 0 keeps its E = template row 0
                const int dn = l4_dpp0<L4_SHL1>(V[c]);
                f[c] = up + dn; e[c] = dn - up;
            }
            // 24-bit multiply-adds (all terms are below 2^19): the 32-bit integer multiply issues at quarter rate
            int gx[16], gy[16], pv[16];
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const int sx = l4_mad24_kv<10>(V[i + 2] - V[i], l4_mad24_k<3, 16>(f[i + 2] - f[i]));
                const int sy = l4_mad24_kv<10>(e[i + 1], l4_mad24_k<3, 16>(e[i] + e[i + 2]));
                gx[i] = sx >> 5;
                gy[i] = sy >> 5;
                pv[i] = V[i + 1];
            }
            gx[15] = 0; gy[15] = 0; pv[15] = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                // lane 15 is no window row: its gradients are zero, so its pixels drop out of every sum
                const int ix = (int)__builtin_amdgcn_perm((uint32_t)gx[2 * m + 1], (uint32_t)gx[2 * m], 0x05040100u);
                const int iy = (int)__builtin_amdgcn_perm((uint32_t)gy[2 * m + 1], (uint32_t)gy[2 * m], 0x05040100u);
                Ixp[m] = winrow ? ix : 0;
                Iyp[m] = winrow ? iy : 0;
                Pp[m] = (int)__builtin_amdgcn_perm((uint32_t)pv[2 * m + 1], (uint32_t)pv[2 * m], 0x05040100u);
                A11 = l4_dot2(Ixp[m], Ixp[m], A11); A12 = l4_dot2(Ixp[m], Iyp[m], A12); A22 = l4_dot2(Iyp[m], Iyp[m], A22);
            }
        }
        const long long A11s = l4_row_sum(A11), A12s = l4_row_sum(A12), A22s = l4_row_sum(A22);
        const double a11 = lk_scaled_f64(A11s), a12 = lk_scaled_f64(A12s), a22 = lk_scaled_f64(A22s);   // (double)A * 2^-20
        double D = a11 * a22 - a12 * a12;
        const double dd = a11 - a22;
        // minEig = numer / (2 * 15 * 15) < 1e-4  <=>  numer < 0x1.70a3d70a3d70bp-5: that constant is the smallest double whose
        // quotient by 450.0 rounds to >= 1e-4 (division is monotonic), so the test is the same without dividing
        const double numer = a22 + a11 - sqrt(dd * dd + 4.0 * a12 * a12);
        bool run = lvl_on && !(numer < 0x1.70a3d70a3d70bp-5 || D < 1.1920928955078125e-07);
        if (lvl_on && !run && l == 0) status = 0;
        const bool solved = run;
        D = 1.0 / D;
        float pdx = 0.f, pdy = 0.f;
        for (int it = 0; it < LK_ITERS; ++it) {
            if (!__any(run)) break;
#ifdef LK_ITER_DBG
            if (run) dbg_iters += 1u << (8 * l);
#endif
            int inx = (int)floorf(wx), iny = (int)floorf(wy);
            if (run && (inx < -LK_WIN || inx >= bw || iny < -LK_WIN || iny >= bh)) {
                if (l == 0) status = 0;
                run = false;
            }
            int ox = inx - bx0, oy = iny - by0;
            const bool need = run && (ox < 0 || ox > L4_MAXOX || oy < 0 || oy > L4_MAXOY);
            if (__any(need)) {
                if (need) { bx0 = (inx - 2) & ~3; by0 = iny - 2; }
                __syncthreads();
                l4_stage<L4_SROWS>(imB, bw, bh, bx0, by0, sS, r, need);
                __syncthreads();
                ox = inx - bx0; oy = iny - by0;
            }
            if (!run) { ox = 0; oy = 0; }                      // idle slot: stay inside the staged region
            int wt, wb;
            l4_weights(wx - (float)inx, wy - (float)iny, wt, wb);
            uint32_t d[5], a[4];
            const uint32_t *rp = sS + l4_mad24_kv<L4_DW>(oy + r, ox >> 2);
#pragma unroll
            for (int i = 0; i < 5; ++i) d[i] = rp[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], ox & 3);
            int pr[15];
            L4Pairs<15>::run(a, pr);
            int sv[16];                                      // 512 x the bilinear sample (< 2^23, positive)
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const int t = l4_dot2k(pr[k], wt, 256), b = l4_dot2k(pr[k], wb, 0);
                sv[k] = t + l4_dpp0<L4_SHL1>(b);
            }
            sv[15] = 0;
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const l4_v2s sp = __builtin_bit_cast(l4_v2s, l4_pack_shr9(sv[2 * m], sv[2 * m + 1]));
                const int df = __builtin_bit_cast(int, (l4_v2s)(sp - __builtin_bit_cast(l4_v2s, Pp[m])));
                b1 = l4_dot2(df, Ixp[m], b1);            // masked pixels carry Ix = Iy = 0
                b2 = l4_dot2(df, Iyp[m], b2);
            }
            const long long b1s = l4_row_sum(b1), b2s = l4_row_sum(b2);
            const double db1 = lk_scaled_f64(b1s), db2 = lk_scaled_f64(b2s);
            const float dx = (float)((a12 * db2 - a22 * db1) * D);
            const float dy = (float)((a12 * db1 - a11 * db2) * D);
            if (run) {
                wx += dx; wy += dy;
                if ((double)dx * (double)dx + (double)dy * (double)dy <= 1e-4) run = false;
                else if (it > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                    wx -= dx * 0.5f; wy -= dy * 0.5f;
                    run = false;
                }
                pdx = dx; pdy = dy;
            }
        }
        if (solved) { ncx = wx + (float)LK_HALF; ncy = wy + (float)LK_HALF; }
    }
    out_x = ncx; out_y = ncy; out_status = status;
}

// One launch per track call (mskf_fe_track*): for the four points of a wavefront, with the point state in registers,
//   temporal LK prev cam0 -> curr cam0 from the predicted position (:321-350, :410)      [do_temporal streams only]
//   -> image-bounds gate (:416-424) -> stereo initial guess undistort(cam0, R01) . distort(cam1) (:542-548)
//   -> stereo LK curr cam0 -> curr cam1 (:569) -> image-bounds gate (:575-583), undistorted points (:601-604, also what
//   publish() sends, :1154-1155) and the epipolar gate (:605-617).
// Round 2 ran this chain as five launches (k_lk_points4 x 2 + k_pt_geom x 3, one thread per point for the geometry) with
// out0 / out1 / status making a round trip through global memory between each; the per-point double-precision geometry is
// a few hundred instructions, issued here once per wavefront for its four points (the sixteen lanes of a row compute the
// same values).  The two tracks run through ONE copy of the LK code (a two-trip loop), so the kernel is no larger.
// Outputs per point: out0 = cam0 point in the current image (tracked, or the input of a stereo-only stream), out1 = matched
// cam1 point, und0 / und1 = their undistorted normalised coordinates, status bit 0 = survived the temporal half, bit 1 =
// stereo match accepted; a point that fails the temporal half gets out1 = und0 = und1 = 0 and status 0.
// Block -> (stream, point group): the blocks b and b + 8 share an XCD (round-robin dispatch, speed only), so the point
// groups of ONE stream are given ids that are congruent modulo 8: a stream's pyramid levels then travel through one L2.
__global__ __launch_bounds__(64) void k_track4(const FeStreamDev *streams, int n_streams, int groups_per_stream) {
    const int x = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int si = x + 8 * (qb / groups_per_stream), gi = qb - (qb / groups_per_stream) * groups_per_stream;
    if (si >= n_streams) return;
    const FeStreamDev &S = streams[si];
    // the point count may live on the device (candidates chosen by fe_book1): the grid is then sized from an estimate and
    // a block takes several point groups if there are more
    const int n_pts = S.n_pts_dev ? *S.n_pts_dev : S.n_pts;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    __shared__ uint32_t s_T[4][L4_TROWS * L4_DW + 2];
    __shared__ uint32_t s_S[4][L4_SROWS * L4_DW + 4];
    uint32_t *sT = s_T[g], *sS = s_S[g];
#pragma nounroll
    for (int gq = gi; 4 * gq < n_pts; gq += groups_per_stream) {
    const int pt = 4 * gq + g;
    const bool in_range = pt < n_pts;
    mskf_point2f pin = {0.f, 0.f};
    if (in_range) pin = S.in_pts[pt];
    unsigned int dbg_iters = 0, dbg_t = 0;
    // state of the point between the tracks
    float c0x = pin.x, c0y = pin.y;           // cam0 point in the current image
    bool ok = in_range;                        // survived the temporal half
    float tx = 0.f, ty = 0.f, gx = 0.f, gy = 0.f;
    bool alive = false;
    int ph = 1;
    if (S.do_temporal) {
        // predictFeatureTracking (:342-347): p2 = H p1, normalise, round to float
        const double *Hm = S.Hpred;
        const double px = (double)pin.x, py = (double)pin.y;
        const double X = Hm[0] * px + Hm[1] * py + Hm[2] * 1.0;
        const double Y = Hm[3] * px + Hm[4] * py + Hm[5] * 1.0;
        const double Z = Hm[6] * px + Hm[7] * py + Hm[8] * 1.0;
        tx = pin.x; ty = pin.y; gx = (float)(X / Z); gy = (float)(Y / Z);
        alive = in_range;
        ph = 0;
    }
    float c1x = 0.f, c1y = 0.f;
    int st1 = 0;
#pragma nounroll
    for (; ph < 2; ++ph) {
        if (ph == 1) {
            // stereo initial guess (:542-548) of the points that are still there
            gx = 0.f; gy = 0.f;
            if (ok) {
                float rx, ry;
                undistort_pt(S.cam0, S.R01, c0x, c0y, rx, ry);
                distort_pt(S.cam1, rx, ry, gx, gy);
            }
            tx = c0x; ty = c0y; alive = ok;
        }
        float nx = 0.f, ny = 0.f;
        int st = 0;
        if (__any(alive)) l4_track(ph ? S.curr0 : S.prev0, ph ? S.curr1 : S.curr0, alive, tx, ty, gx, gy, sT, sS, r, nx, ny, st, dbg_iters);
        if (ph == 0) {
            // temporal result and its image-bounds gate (:416-424)
            const int W = S.curr0.w[0], H = S.curr0.h[0];
            c0x = nx; c0y = ny;
            ok = alive && st != 0 && !(c0y < 0 || c0y > (float)(H - 1) || c0x < 0 || c0x > (float)(W - 1));
            dbg_t = dbg_iters; dbg_iters = 0;
        } else {
            c1x = alive ? nx : 0.f; c1y = alive ? ny : 0.f; st1 = alive ? st : 0;
        }
    }
    // stereo result: image-bounds gate (:575-583), undistorted points (:601-604), epipolar gate (:605-617)
    float u0x = 0.f, u0y = 0.f, u1x = 0.f, u1y = 0.f;
    int sst = st1;
    if (ok) {
        const int W1 = S.curr1.w[0], H1 = S.curr1.h[0];
        if (sst && (c1y < 0 || c1y > (float)(H1 - 1) || c1x < 0 || c1x > (float)(W1 - 1))) sst = 0;
        undistort_pt(S.cam0, nullptr, c0x, c0y, u0x, u0y);
        undistort_pt(S.cam1, nullptr, c1x, c1y, u1x, u1y);
        if (sst) {
            const double *E = S.E;
            const double x0 = (double)u0x, y0 = (double)u0y, x1 = (double)u1x, y1 = (double)u1y;
            const double l0 = (E[0] * x0 + E[1] * y0) + E[2];
            const double l1 = (E[3] * x0 + E[4] * y0) + E[5];
            const double l2 = (E[6] * x0 + E[7] * y0) + E[8];
            const double err = fabs((x1 * l0 + y1 * l1) + l2) / sqrt(l0 * l0 + l1 * l1);
            if (err > S.epi_thresh) sst = 0;
        }
    }
    if (in_range && r == 0) {
        S.out0[pt] = mskf_point2f{c0x, c0y};
        S.out1[pt] = mskf_point2f{c1x, c1y};
        S.und0[pt] = mskf_point2f{u0x, u0y};
#ifdef LK_ITER_DBG
        ((unsigned int *)S.und1)[2 * pt] = dbg_t; ((unsigned int *)S.und1)[2 * pt + 1] = dbg_iters;
#else
        S.und1[pt] = mskf_point2f{u1x, u1y};
        (void)dbg_t;
#endif
        S.status[pt] = (uint8_t)(ok ? (1 | (sst ? 2 : 0)) : 0);
    }
    }
}

// ------------------------------------------------------------------------------------------ bookkeeping (fe_book.h)
#include "fe_book.h"
// One workgroup per VIO stream: which = 0 after the first track call of the frame, 1 after the second.
__global__ __launch_bounds__(256) void k_fe_book(const FeBookDev *books, int which) {
    const FeBookDev &B = books[blockIdx.x];
    extern __shared__ int s_book[];
    FeBookScratch L;
    fe_book_scratch_init(L, s_book, B.cap, B.cand_cap, B.det_cap, B.n_codes, B.det_rows * B.det_cols);
    if (which == 0) fe_book1(B, L); else fe_book2(B, L);
}
// Dynamic LDS the bookkeeping kernel may use: 150 KiB once the attribute is granted (the 4K configuration's lists need more
// than the 64 KiB a kernel gets by default), 64 KiB otherwise.  mskf_stream_create sizes the device books against it.
extern "C" size_t fe_book_lds_budget(void) {
    static const size_t budget = []() -> size_t {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fe_book), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { (void)hipGetLastError(); return 64 * 1024; }
        return 150 * 1024;
    }();
    return budget;
}
extern "C" void fe_launch_book(const FeBookDev *books_dev, int n_streams, int which, size_t scratch_bytes, hipStream_t st) {
    (void)fe_book_lds_budget();
    hipLaunchKernelGGL(k_fe_book, dim3(n_streams), dim3(256), scratch_bytes, st, books_dev, which);
}

// completion mark of the spinning wait (mskf_wait_event): one thread stores a sequence number into pinned host memory
// once everything enqueued before it on the stream (the D2H copies included) is done
__global__ void k_mark(volatile unsigned int *flag, unsigned int seq) {
    *flag = seq;
    __threadfence_system();
}
extern "C" void fe_launch_mark(volatile unsigned int *flag, unsigned int seq, hipStream_t st) {
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, st, flag, seq);
}

// Staging copies as a kernel of the stream itself (pinned host memory <-> device memory, both addressable from a kernel).
// hipMemcpyAsync hands such copies to the SDMA engines: a second kind of queue, shared by every stream of the process, with a
// cross-queue signal either side of each copy.  With sixteen busy streams the copies of all of them convoy behind whichever
// copy is waiting for its own stream's kernel, and now and then (one bench run in ten to twenty, round 4) every stream's next
// copy stood still for seconds up to a minute: every queue of the pipeline was found waiting in front of a staging copy.
// One launch moves up to MSKF_COPY_SEGS segments; 16-byte accesses (the arenas are 64-byte aligned), the odd bytes at the
// end of a segment one by one.
struct MskfCopySegs { void *dst[MSKF_COPY_SEGS]; const void *src[MSKF_COPY_SEGS]; unsigned long long bytes[MSKF_COPY_SEGS]; int blocks_per_seg; };
__global__ __launch_bounds__(256) void k_mskf_copy(MskfCopySegs segs) {
    const int seg = (int)blockIdx.x / segs.blocks_per_seg, b = (int)blockIdx.x - seg * segs.blocks_per_seg;
    const unsigned long long bytes = segs.bytes[seg], n16 = bytes >> 4;
    const uint4 *__restrict__ src = (const uint4 *)segs.src[seg];
    uint4 *__restrict__ dst = (uint4 *)segs.dst[seg];
    const unsigned long long stride = (unsigned long long)segs.blocks_per_seg * 256ULL;
    unsigned long long i = (unsigned long long)b * 256ULL + threadIdx.x;
    for (; i + 3ULL * stride < n16; i += 4ULL * stride) {          // four loads in flight per lane (the source may sit across PCIe)
        const uint4 a0 = src[i], a1 = src[i + stride], a2 = src[i + 2ULL * stride], a3 = src[i + 3ULL * stride];
        dst[i] = a0; dst[i + stride] = a1; dst[i + 2ULL * stride] = a2; dst[i + 3ULL * stride] = a3;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
    if (b == 0 && threadIdx.x < (unsigned)(bytes & 15ULL)) ((uint8_t *)dst)[(n16 << 4) + threadIdx.x] = ((const uint8_t *)src)[(n16 << 4) + threadIdx.x];
}
extern "C" void fe_launch_copy(void *const *dst, const void *const *src, const size_t *bytes, int n_segs, hipStream_t st) {
    MskfCopySegs segs;
    size_t mx = 0;
    for (int i = 0; i < MSKF_COPY_SEGS; ++i) {
        segs.dst[i] = i < n_segs ? dst[i] : nullptr; segs.src[i] = i < n_segs ? src[i] : nullptr; segs.bytes[i] = i < n_segs ? bytes[i] : 0ULL;
        if (i < n_segs && bytes[i] > mx) mx = bytes[i];
    }
    if (n_segs <= 0 || mx == 0) return;
    size_t bps = (mx + 16383) / 16384;                  // 16 KiB per workgroup and sweep
    segs.blocks_per_seg = (int)(bps < 1 ? 1 : bps > 96 ? 96 : bps);
    hipLaunchKernelGGL(k_mskf_copy, dim3(segs.blocks_per_seg * n_segs), dim3(256), 0, st, segs);
}

extern "C" void fe_launch_track(const FeStreamDev *streams_dev, int n_streams, int max_pts, hipStream_t st) {
    if (max_pts <= 0) return;
    const int gps = (max_pts + 3) / 4;
    hipLaunchKernelGGL(k_track4, dim3(8 * ((n_streams + 7) / 8) * gps), dim3(64), 0, st, streams_dev, n_streams, gps);
}
