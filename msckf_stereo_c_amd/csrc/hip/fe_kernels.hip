// fe_kernels.hip — hand-written gfx950 kernels of the stereo KLT front-end.
//
//  k_pyr_down     : cg::pyr_down            (reference call sites image_processor.cpp:239,242)
//  k_detect_cells : cg::CornerDetector      (:132,259,657) — per-cell integer Shi-Tomasi maximum
//  k_lk_points    : cg::optical_flow_multi_level (:410 temporal, :569 stereo) fused with the
//                   prediction (:321-350), the image-bounds gates (:416-424, :575-583), the stereo
//                   initial guess (:542-548), undistortion and the epipolar gate (:587-617).
//
// Arithmetic contract: every decision-bearing quantity is integer or a fixed
// sequence of IEEE-754 double operations; this file must be compiled with -ffp-contract=off.
// Work shapes: one 64-lane wavefront per tracked point (a 15x15 window = 225 pixels, <= 4 per
// lane), one workgroup per detector cell, one workgroup per 64x16 pyramid output tile; the
// stream index of the batch is blockIdx.y / blockIdx.z, so a launch covers every VIO stream of a
// context and fills the chip only when many streams are batched.
#include "fe_device.h"

// ------------------------------------------------------------------------------------------ pyr_down
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { i = (i < 0) ? -i : 2 * (n - 1) - i; }
    return i;
}

#define PD_TW 64
#define PD_TH 16
struct PyrJob { const uint8_t *src; uint8_t *dst; int sw, sh, dw, dh; };

// One workgroup: a 64x16 output tile.  The (2*64+4) x (2*16+4) source footprint is staged in LDS
// once (reflect-101 at the image border), the horizontal [1 4 6 4 1] pass is done into a second
// LDS plane, the vertical pass reads that.
__global__ __launch_bounds__(256) void k_pyr_down(const PyrJob *jobs) {
    const PyrJob job = jobs[blockIdx.z];
    const int ox0 = blockIdx.x * PD_TW, oy0 = blockIdx.y * PD_TH;
    if (ox0 >= job.dw || oy0 >= job.dh) return;
    constexpr int SW = 2 * PD_TW + 4, SH = 2 * PD_TH + 4;
    __shared__ uint8_t s_src[SH][SW + 4];
    __shared__ uint16_t s_h[SH][PD_TW];
    const int tid = threadIdx.x;
    const int sx0 = 2 * ox0 - 2, sy0 = 2 * oy0 - 2;
    for (int i = tid; i < SW * SH; i += 256) {
        const int r = i / SW, c = i - r * SW;
        const int sy = reflect101(sy0 + r, job.sh), sx = reflect101(sx0 + c, job.sw);
        s_src[r][c] = job.src[(size_t)sy * job.sw + sx];
    }
    __syncthreads();
    for (int i = tid; i < SH * PD_TW; i += 256) {
        const int r = i / PD_TW, c = i - r * PD_TW;
        const uint8_t *p = &s_src[r][2 * c];
        s_h[r][c] = (uint16_t)(p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]);
    }
    __syncthreads();
    for (int i = tid; i < PD_TW * PD_TH; i += 256) {
        const int r = i / PD_TW, c = i - r * PD_TW;
        const int ox = ox0 + c, oy = oy0 + r;
        if (ox < job.dw && oy < job.dh) {
            const int s = s_h[2 * r][c] + 4 * s_h[2 * r + 1][c] + 6 * s_h[2 * r + 2][c] + 4 * s_h[2 * r + 3][c] + s_h[2 * r + 4][c];
            job.dst[(size_t)oy * job.dw + ox] = (uint8_t)((s + 128) >> 8);
        }
    }
}

extern "C" void fe_launch_pyr_down(const PyrJob *jobs_dev, int n_jobs, int max_dw, int max_dh, hipStream_t st) {
    dim3 grid((max_dw + PD_TW - 1) / PD_TW, (max_dh + PD_TH - 1) / PD_TH, n_jobs);
    hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, st, jobs_dev);
}

// ------------------------------------------------------------------------------------------ detector
__device__ __forceinline__ long long isqrt64(long long v) {
    long long r = (long long)sqrt((double)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

#define DET_BORDER 8

// One workgroup per detector cell.  The cell (+5 px halo) is staged in LDS in sub-tiles of <= 32x32 pixels;
// gradient products are formed once per pixel, the 8x8 box sums are separable (8-wide horizontal pass,
// then 8-tall vertical pass), and the block reduces to the (max score, first in row-major order) corner.
#define DT 32
__global__ __launch_bounds__(256) void k_detect_cells(const FeStreamDev *streams) {
    const FeStreamDev &S = streams[blockIdx.y];
    const int cell = blockIdx.x;
    if (cell >= S.det_rows * S.det_cols) return;
    const int W = S.curr0.w[0], H = S.curr0.h[0];
    const uint8_t *img = S.curr0.lvl[0];
    const int cw = S.cell_w, ch = S.cell_h;
    const int cy = cell / S.det_cols, cx = cell - cy * S.det_cols;
    const int x0 = cx * cw, y0 = cy * ch;
    __shared__ uint8_t tile[(DT + 10) * (DT + 10)];
    __shared__ int sG[3][(DT + 8) * (DT + 8)];    // dx*dx, dx*dy, dy*dy at tile (r+1, c+1)
    __shared__ int sH[3][(DT + 8) * DT];          // horizontal 8-sums
    __shared__ unsigned long long s_best[4];
    unsigned long long best = 0ULL;
    for (int ty = 0; ty < ch; ty += DT)
        for (int tx = 0; tx < cw; tx += DT) {
            const int tw = min(DT, cw - tx), th = min(DT, ch - ty);
            const int lw = tw + 10, lh = th + 10;
            const int gw = tw + 8, gh = th + 8;
            const int gx0 = x0 + tx - 5, gy0 = y0 + ty - 5;
            __syncthreads();
            for (int i = threadIdx.x; i < lw * lh; i += 256) {
                const int r = i / lw, c = i - r * lw;
                const int gx = min(max(gx0 + c, 0), W - 1), gy = min(max(gy0 + r, 0), H - 1);
                tile[i] = img[(size_t)gy * W + gx];
            }
            __syncthreads();
            for (int i = threadIdx.x; i < gw * gh; i += 256) {
                const int r = i / gw, c = i - r * gw;          // tile position (r+1, c+1)
                const uint8_t *t = tile + (r + 1) * lw + (c + 1);
                const int dx = (int)t[1] - (int)t[-1];
                const int dy = (int)t[lw] - (int)t[-lw];
                sG[0][i] = dx * dx; sG[1][i] = dx * dy; sG[2][i] = dy * dy;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < gh * tw; i += 256) {
                const int r = i / tw, c = i - r * tw;
                const int *g0 = sG[0] + r * gw + c, *g1 = sG[1] + r * gw + c, *g2 = sG[2] + r * gw + c;
                int a = 0, b = 0, d = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) { a += g0[u]; b += g1[u]; d += g2[u]; }
                sH[0][i] = a; sH[1][i] = b; sH[2][i] = d;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < tw * th; i += 256) {
                const int r = i / tw, c = i - r * tw;
                const int x = x0 + tx + c, y = y0 + ty + r;
                if (x < DET_BORDER || y < DET_BORDER || x >= W - DET_BORDER || y >= H - DET_BORDER) continue;
                long long a = 0, b = 0, cc = 0;
#pragma unroll
                for (int v = 0; v < 8; ++v) { a += sH[0][(r + v) * tw + c]; b += sH[1][(r + v) * tw + c]; cc += sH[2][(r + v) * tw + c]; }
                const long long disc = (a - cc) * (a - cc) + 4 * b * b;
                const long long score = (a + cc) - isqrt64(disc);
                if (score > 0) {
                    // row-major scan order inside the cell decides ties: smaller (y, x) wins
                    const unsigned int order = (unsigned int)((y - y0) * cw + (x - x0));
                    const unsigned long long key = ((unsigned long long)score << 32) | (0xFFFFFFFFu - order);
                    best = key > best ? key : best;
                }
            }
        }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = s_best[i] > best ? s_best[i] : best;
        mskf_corner out;
        out.cell = cell;
        if (best == 0ULL) { out.x = 0.f; out.y = 0.f; out.score = 0; }
        else {
            const unsigned int order = 0xFFFFFFFFu - (unsigned int)(best & 0xFFFFFFFFULL);
            out.score = (int)(best >> 32);
            out.y = (float)(y0 + (int)(order / (unsigned)cw));
            out.x = (float)(x0 + (int)(order % (unsigned)cw));
        }
        S.cell_max[cell] = out;
    }
}

extern "C" void fe_launch_detect(const FeStreamDev *streams_dev, int n_streams, int max_cells, hipStream_t st) {
    hipLaunchKernelGGL(k_detect_cells, dim3(max_cells, n_streams), dim3(256), 0, st, streams_dev);
}

// ------------------------------------------------------------------------------------------ point math
// cv::undistortPoints radtan, 5 iterations, then R, then P = {1,1,0,0}
__device__ __forceinline__ void undistort_pt(const CamDev &cam, const double *R, float u, float v, float &xo, float &yo) {
    const double fx = cam.K[0], fy = cam.K[1], cx = cam.K[2], cy = cam.K[3];
    const double k1 = cam.D[0], k2 = cam.D[1], p1 = cam.D[2], p2 = cam.D[3];
    double x = ((double)u - cx) / fx, y = ((double)v - cy) / fy;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2);
        const double deltaX = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        const double deltaY = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
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

__device__ __forceinline__ void distort_pt(const CamDev &cam, float xf, float yf, float &uo, float &vo) {
    const double fx = cam.K[0], fy = cam.K[1], cx = cam.K[2], cy = cam.K[3];
    const double k1 = cam.D[0], k2 = cam.D[1], p1 = cam.D[2], p2 = cam.D[3];
    const double x = (double)xf, y = (double)yf;
    const double r2 = x * x + y * y, r4 = r2 * r2;
    const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
    const double cdist = 1.0 + k1 * r2 + k2 * r4;
    const double xd = x * cdist + p1 * a1 + p2 * a2;
    const double yd = y * cdist + p1 * a3 + p2 * a1;
    uo = (float)(xd * fx + cx);
    vo = (float)(yd * fy + cy);
}

// ------------------------------------------------------------------------------------------ LK
#define LK_HALF 7
#define LK_WIN 15
#define LK_ITERS 30

__device__ __forceinline__ int px_clamped(const uint8_t *img, int w, int h, int x, int y) {
    x = min(max(x, 0), w - 1);
    y = min(max(y, 0), h - 1);
    return img[(size_t)y * w + x];
}

__device__ __forceinline__ void bilinear_weights(float fa, float fb, int &w00, int &w01, int &w10, int &w11) {
    const int qa = __float2int_rn(fa * 16384.0f);
    const int qb = __float2int_rn(fb * 16384.0f);
    w00 = ((16384 - qa) * (16384 - qb) + 8192) >> 14;
    w01 = (qa * (16384 - qb) + 8192) >> 14;
    w10 = ((16384 - qa) * qb + 8192) >> 14;
    w11 = 16384 - w00 - w01 - w10;
}

__device__ __forceinline__ int sample5(const uint8_t *img, int w, int h, int x, int y, int w00, int w01, int w10, int w11) {
    const int s = px_clamped(img, w, h, x, y) * w00 + px_clamped(img, w, h, x + 1, y) * w01 +
                  px_clamped(img, w, h, x, y + 1) * w10 + px_clamped(img, w, h, x + 1, y + 1) * w11;
    return (s + 256) >> 9;
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Pyramidal LK for one point, executed by one full wavefront (all 64 lanes call this with
// identical arguments; control flow is wave-uniform).  s_P is a per-wave LDS scratch of 17*17 ints.
__device__ void lk_point(const PyrDev &A, const PyrDev &B, float ax, float ay, float &bx, float &by, int &status, int *s_P) {
    const int lane = threadIdx.x & 63;
    const double FLT_SCALE = 1.0 / (double)(1 << 20);
    status = 1;
    float ncx = 0.f, ncy = 0.f;
    for (int l = MSKF_LEVELS - 1; l >= 0; --l) {
        const uint8_t *imA = A.lvl[l];
        const uint8_t *imB = B.lvl[l];
        const int aw = A.w[l], ah = A.h[l], bw = B.w[l], bh = B.h[l];
        const float sc = 1.0f / (float)(1 << l);
        const float pwx = ax * sc - (float)LK_HALF, pwy = ay * sc - (float)LK_HALF;
        if (l == MSKF_LEVELS - 1) { ncx = bx * sc; ncy = by * sc; }
        else { ncx = ncx * 2.0f; ncy = ncy * 2.0f; }
        const int ipx = (int)floorf(pwx), ipy = (int)floorf(pwy);
        if (ipx < -LK_WIN || ipx >= aw || ipy < -LK_WIN || ipy >= ah) {
            if (l == 0) status = 0;
            continue;
        }
        int w00, w01, w10, w11;
        bilinear_weights(pwx - (float)ipx, pwy - (float)ipy, w00, w01, w10, w11);
        // 17x17 interpolated template -> LDS
        __syncthreads();
        for (int i = lane; i < 17 * 17; i += 64) {
            const int r = i / 17, c = i - r * 17;
            s_P[i] = sample5(imA, aw, ah, ipx + c - 1, ipy + r - 1, w00, w01, w10, w11);
        }
        __syncthreads();
        // each lane owns pixels lane, lane+64, lane+128, lane+192 (< 225) of the 15x15 window
        int Pv[4], Ix[4], Iy[4], pj[4], pi[4];
        long long A11 = 0, A12 = 0, A22 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k;
            Pv[k] = 0; Ix[k] = 0; Iy[k] = 0; pj[k] = 0; pi[k] = 0;
            if (idx < LK_WIN * LK_WIN) {
                const int j = idx / LK_WIN, i = idx - j * LK_WIN;
                pj[k] = j; pi[k] = i;
                const int *p = s_P + (j + 1) * 17 + (i + 1);
                const int sx = 3 * (p[-17 + 1] - p[-17 - 1]) + 10 * (p[1] - p[-1]) + 3 * (p[17 + 1] - p[17 - 1]);
                const int sy = 3 * (p[17 - 1] - p[-17 - 1]) + 10 * (p[17] - p[-17]) + 3 * (p[17 + 1] - p[-17 + 1]);
                const int gx = (sx + 16) >> 5, gy = (sy + 16) >> 5;
                Pv[k] = p[0]; Ix[k] = gx; Iy[k] = gy;
                A11 += (long long)gx * gx; A12 += (long long)gx * gy; A22 += (long long)gy * gy;
            }
        }
        A11 = wave_sum_i64(A11); A12 = wave_sum_i64(A12); A22 = wave_sum_i64(A22);
        const double a11 = (double)A11 * FLT_SCALE, a12 = (double)A12 * FLT_SCALE, a22 = (double)A22 * FLT_SCALE;
        double D = a11 * a22 - a12 * a12;
        const double dd = a11 - a22;
        const double minEig = (a22 + a11 - sqrt(dd * dd + 4.0 * a12 * a12)) / (2.0 * LK_WIN * LK_WIN);
        if (minEig < 1e-4 || D < 1.1920928955078125e-07) {
            if (l == 0) status = 0;
            continue;
        }
        D = 1.0 / D;
        float wx = ncx - (float)LK_HALF, wy = ncy - (float)LK_HALF;
        float pdx = 0.f, pdy = 0.f;
        for (int it = 0; it < LK_ITERS; ++it) {
            const int inx = (int)floorf(wx), iny = (int)floorf(wy);
            if (inx < -LK_WIN || inx >= bw || iny < -LK_WIN || iny >= bh) {
                if (l == 0) status = 0;
                break;
            }
            bilinear_weights(wx - (float)inx, wy - (float)iny, w00, w01, w10, w11);
            long long b1 = 0, b2 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (lane + 64 * k < LK_WIN * LK_WIN) {
                    const int diff = sample5(imB, bw, bh, inx + pi[k], iny + pj[k], w00, w01, w10, w11) - Pv[k];
                    b1 += (long long)(diff * Ix[k]);
                    b2 += (long long)(diff * Iy[k]);
                }
            }
            b1 = wave_sum_i64(b1); b2 = wave_sum_i64(b2);
            const double db1 = (double)b1 * FLT_SCALE, db2 = (double)b2 * FLT_SCALE;
            const float dx = (float)((a12 * db2 - a22 * db1) * D);
            const float dy = (float)((a12 * db1 - a11 * db2) * D);
            wx += dx; wy += dy;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= 1e-4) break;
            if (it > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                wx -= dx * 0.5f; wy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        ncx = wx + (float)LK_HALF; ncy = wy + (float)LK_HALF;
    }
    bx = ncx; by = ncy;
}

// One wavefront (= one 64-thread workgroup) per point; blockIdx.y = stream of the batch.
__global__ __launch_bounds__(64) void k_lk_points(const FeStreamDev *streams) {
    const FeStreamDev &S = streams[blockIdx.y];
    const int pt = blockIdx.x;
    if (pt >= S.n_pts) return;
    __shared__ int s_P[17 * 17];
    const int W = S.curr0.w[0], H = S.curr0.h[0];
    const mskf_point2f pin = S.in_pts[pt];
    int st_bits = 0;
    float c0x = pin.x, c0y = pin.y;
    bool ok = true;
    if (S.do_temporal) {
        // predictFeatureTracking (:342-347): p2 = H p1, normalise, round to float
        const double *Hm = S.Hpred;
        const double px = (double)pin.x, py = (double)pin.y;
        const double X = Hm[0] * px + Hm[1] * py + Hm[2] * 1.0;
        const double Y = Hm[3] * px + Hm[4] * py + Hm[5] * 1.0;
        const double Z = Hm[6] * px + Hm[7] * py + Hm[8] * 1.0;
        float bx = (float)(X / Z), by = (float)(Y / Z);
        int st;
        lk_point(S.prev0, S.curr0, pin.x, pin.y, bx, by, st, s_P);
        c0x = bx; c0y = by;
        // :416-424
        if (st && (c0y < 0 || c0y > (float)(H - 1) || c0x < 0 || c0x > (float)(W - 1))) st = 0;
        ok = st != 0;
        if (ok) st_bits |= 1;
    } else {
        st_bits |= 1;
    }
    float c1x = 0.f, c1y = 0.f, u0x = 0.f, u0y = 0.f, u1x = 0.f, u1y = 0.f;
    if (ok) {
        // stereo initial guess (:542-548)
        float rx, ry;
        undistort_pt(S.cam0, S.R01, c0x, c0y, rx, ry);
        distort_pt(S.cam1, rx, ry, c1x, c1y);
        int st;
        lk_point(S.curr0, S.curr1, c0x, c0y, c1x, c1y, st, s_P);
        // :575-583
        const int W1 = S.curr1.w[0], H1 = S.curr1.h[0];
        if (st && (c1y < 0 || c1y > (float)(H1 - 1) || c1x < 0 || c1x > (float)(W1 - 1))) st = 0;
        // :601-617 (the undistorted points are also what publish() sends, :1154-1155)
        undistort_pt(S.cam0, nullptr, c0x, c0y, u0x, u0y);
        undistort_pt(S.cam1, nullptr, c1x, c1y, u1x, u1y);
        if (st) {
            const double *E = S.E;
            const double x0 = (double)u0x, y0 = (double)u0y, x1 = (double)u1x, y1 = (double)u1y;
            const double l0 = (E[0] * x0 + E[1] * y0) + E[2];
            const double l1 = (E[3] * x0 + E[4] * y0) + E[5];
            const double l2 = (E[6] * x0 + E[7] * y0) + E[8];
            const double err = fabs((x1 * l0 + y1 * l1) + l2) / sqrt(l0 * l0 + l1 * l1);
            if (err > S.epi_thresh) st = 0;
        }
        if (st) st_bits |= 2;
    }
    if ((threadIdx.x & 63) == 0) {
        S.out0[pt] = mskf_point2f{c0x, c0y};
        S.out1[pt] = mskf_point2f{c1x, c1y};
        S.und0[pt] = mskf_point2f{u0x, u0y};
        S.und1[pt] = mskf_point2f{u1x, u1y};
        S.status[pt] = (uint8_t)st_bits;
    }
}

extern "C" void fe_launch_lk(const FeStreamDev *streams_dev, int n_streams, int max_pts, hipStream_t st) {
    if (max_pts <= 0) return;
    hipLaunchKernelGGL(k_lk_points, dim3(max_pts, n_streams), dim3(64), 0, st, streams_dev);
}
