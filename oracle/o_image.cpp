// oracle/o_image.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).  See o_image.h for provenance.
// Build with -ffp-contract=off: every floating-point expression below is a sequence of single
// IEEE-754 operations in the written order (DESIGN.md §3 "arithmetic contract").
#include "o_image.h"
#include "o_math.h"
#include <cmath>
#include <cfloat>
#include <cstring>

namespace orc {

// ------------------------------------------------------------------ K1 pyr_down
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

// Reference call sites: image_processor.cpp:239,242 (cg::pyr_down).  OpenCV pyrDown for CV_8U:
// separable [1 4 6 4 1], total weight 256, rounding (s+128)>>8, BORDER_REFLECT_101.
void pyr_down(const Img &src, Img &dst) {
    const int W = src.w, H = src.h;
    const int w = (W + 1) / 2, h = (H + 1) / 2;
    dst = Img(w, h);
    static const int k[5] = {1, 4, 6, 4, 1};
    std::vector<int> rowbuf((size_t)w);
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int j = -2; j <= 2; ++j) {
                const int sy = reflect101(2 * y + j, H);
                int hs = 0;
                for (int i = -2; i <= 2; ++i) hs += k[i + 2] * src.d[(size_t)sy * W + reflect101(2 * x + i, W)];
                s += k[j + 2] * hs;
            }
            dst.d[(size_t)y * w + x] = (uint8_t)((s + 128) >> 8);
        }
    }
}

// image_processor.cpp:229-244: level 0 is a copy, levels 1..3 by pyr_down (loop bound literal 4, Q6)
void build_pyramid(const Img &l0, std::vector<Img> &pyr) {
    pyr.clear();
    pyr.push_back(l0);
    for (int i = 1; i < LK_LEVELS; ++i) {
        Img t;
        pyr_down(pyr[i - 1], t);
        pyr.push_back(t);
    }
}

// ------------------------------------------------------------------ K2/K3 LK
// Fixed-point contract (mirrors OpenCV lkpyramid.cpp's W_BITS = 14 scheme):
//  * sub-pixel offsets quantised to 14 bits, bilinear weights integer, w11 = 2^14 - w00 - w01 - w10
//  * interpolated samples keep 5 fractional bits: (sum + 256) >> 9
//  * spatial derivative = Scharr (3,10,3)x(-1,0,1) of the interpolated 17x17 template, (s + 16) >> 5
//  * A11,A12,A22,b1,b2 accumulate in int64 (order independent), then * 2^-20 in double
//  * 2x2 solve in double, written order, result rounded to float; points are float like cv::Point2f
static inline void bilinear_weights(float f_a, float f_b, int &w00, int &w01, int &w10, int &w11) {
    const int qa = (int)lrintf(f_a * 16384.0f);
    const int qb = (int)lrintf(f_b * 16384.0f);
    w00 = ((16384 - qa) * (16384 - qb) + 8192) >> 14;
    w01 = (qa * (16384 - qb) + 8192) >> 14;
    w10 = ((16384 - qa) * qb + 8192) >> 14;
    w11 = 16384 - w00 - w01 - w10;
}

static inline int sample5(const Img &im, int x, int y, int w00, int w01, int w10, int w11) {
    const int s = im.at(x, y) * w00 + im.at(x + 1, y) * w01 + im.at(x, y + 1) * w10 + im.at(x + 1, y + 1) * w11;
    return (s + 256) >> 9;
}

// work counters for sizing the GPU kernel: [0] points, [1] levels solved, [2] iterations
long long g_lk_stats[3] = {0, 0, 0};

void lk_track_point(const std::vector<Img> &pyrA, const std::vector<Img> &pyrB,
                    float ax, float ay, float &bx, float &by, uint8_t &status) {
    const double FLT_SCALE = 1.0 / (1 << 20);
    const double MIN_EIG = 1e-4;
    const double EPS2 = 0.01 * 0.01;
    status = 1;
    ++g_lk_stats[0];
    float ncx = 0.f, ncy = 0.f;  // next point, centre coordinates, current level
    for (int l = LK_LEVELS - 1; l >= 0; --l) {
        const Img &A = pyrA[l];
        const Img &B = pyrB[l];
        const float sc = 1.0f / (float)(1 << l);
        const float pwx = ax * sc - (float)LK_HALF_WIN;  // window top-left, prev
        const float pwy = ay * sc - (float)LK_HALF_WIN;
        if (l == LK_LEVELS - 1) { ncx = bx * sc; ncy = by * sc; }
        else { ncx = ncx * 2.0f; ncy = ncy * 2.0f; }

        const int ipx = (int)floorf(pwx), ipy = (int)floorf(pwy);
        if (ipx < -LK_WIN || ipx >= A.w || ipy < -LK_WIN || ipy >= A.h) {
            if (l == 0) status = 0;
            continue;
        }
        int w00, w01, w10, w11;
        bilinear_weights(pwx - (float)ipx, pwy - (float)ipy, w00, w01, w10, w11);

        // 17x17 interpolated template, index [j+1][i+1] for i,j in [-1,15]
        int P[17][17];
        for (int j = -1; j <= 15; ++j)
            for (int i = -1; i <= 15; ++i)
                P[j + 1][i + 1] = sample5(A, ipx + i, ipy + j, w00, w01, w10, w11);
        int Ix[15][15], Iy[15][15];
        int64_t A11 = 0, A12 = 0, A22 = 0;
        for (int j = 0; j < 15; ++j)
            for (int i = 0; i < 15; ++i) {
                const int (*p)[17] = P;
                const int r = j + 1, c = i + 1;
                const int sx = 3 * (p[r - 1][c + 1] - p[r - 1][c - 1]) + 10 * (p[r][c + 1] - p[r][c - 1]) +
                               3 * (p[r + 1][c + 1] - p[r + 1][c - 1]);
                const int sy = 3 * (p[r + 1][c - 1] - p[r - 1][c - 1]) + 10 * (p[r + 1][c] - p[r - 1][c]) +
                               3 * (p[r + 1][c + 1] - p[r - 1][c + 1]);
                const int gx = (sx + 16) >> 5, gy = (sy + 16) >> 5;
                Ix[j][i] = gx; Iy[j][i] = gy;
                A11 += (int64_t)gx * gx; A12 += (int64_t)gx * gy; A22 += (int64_t)gy * gy;
            }
        const double a11 = (double)A11 * FLT_SCALE, a12 = (double)A12 * FLT_SCALE, a22 = (double)A22 * FLT_SCALE;
        double D = a11 * a22 - a12 * a12;
        const double dd = a11 - a22;
        const double minEig = (a22 + a11 - std::sqrt(dd * dd + 4.0 * a12 * a12)) / (2.0 * LK_WIN * LK_WIN);
        if (minEig < MIN_EIG || D < (double)FLT_EPSILON) {
            if (l == 0) status = 0;
            continue;
        }
        D = 1.0 / D;
        ++g_lk_stats[1];

        float wx = ncx - (float)LK_HALF_WIN, wy = ncy - (float)LK_HALF_WIN;
        float pdx = 0.f, pdy = 0.f;
        for (int it = 0; it < LK_MAX_ITER; ++it) {
            const int inx = (int)floorf(wx), iny = (int)floorf(wy);
            if (inx < -LK_WIN || inx >= B.w || iny < -LK_WIN || iny >= B.h) {
                if (l == 0) status = 0;
                break;
            }
            bilinear_weights(wx - (float)inx, wy - (float)iny, w00, w01, w10, w11);
            ++g_lk_stats[2];
            int64_t b1 = 0, b2 = 0;
            for (int j = 0; j < 15; ++j)
                for (int i = 0; i < 15; ++i) {
                    const int diff = sample5(B, inx + i, iny + j, w00, w01, w10, w11) - P[j + 1][i + 1];
                    b1 += (int64_t)diff * Ix[j][i];
                    b2 += (int64_t)diff * Iy[j][i];
                }
            const double db1 = (double)b1 * FLT_SCALE, db2 = (double)b2 * FLT_SCALE;
            const float dx = (float)((a12 * db2 - a22 * db1) * D);
            const float dy = (float)((a12 * db1 - a11 * db2) * D);
            wx += dx; wy += dy;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= EPS2) break;
            if (it > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                wx -= dx * 0.5f; wy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        ncx = wx + (float)LK_HALF_WIN; ncy = wy + (float)LK_HALF_WIN;
    }
    bx = ncx; by = ncy;
}

// call sites: image_processor.cpp:410 (temporal) and :569 (stereo), window 15, 30 iterations
void lk_track(const std::vector<Img> &pyrA, const std::vector<Img> &pyrB,
              const std::vector<mskf_point2f> &ptsA, std::vector<mskf_point2f> &ptsB,
              std::vector<uint8_t> &status) {
    status.assign(ptsA.size(), 0);
    if (ptsB.size() != ptsA.size()) ptsB = ptsA;
    for (size_t i = 0; i < ptsA.size(); ++i)
        lk_track_point(pyrA, pyrB, ptsA[i].x, ptsA[i].y, ptsB[i].x, ptsB[i].y, status[i]);
}

// ------------------------------------------------------------------ K4 detector
// Integer Shi-Tomasi score on an 8x8 box of central-difference gradients:
//   a = S dx^2, c = S dy^2, b = S dx dy;  score = (a + c) - isqrt((a-c)^2 + 4 b^2)   (== 2*lambda_min, floor)
// response = score / 256.0 (box area 64, gradient scale 2 -> SVO/vikit "shiTomasiScore" units).
static inline int64_t isqrt64(int64_t v) {
    int64_t r = (int64_t)std::sqrt((double)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

enum { DET_BORDER = 8 };

int32_t shi_tomasi_score(const Img &img, int x, int y) {
    if (x < DET_BORDER || y < DET_BORDER || x >= img.w - DET_BORDER || y >= img.h - DET_BORDER) return 0;
    int64_t a = 0, b = 0, c = 0;
    for (int v = y - 4; v < y + 4; ++v)
        for (int u = x - 4; u < x + 4; ++u) {
            const int dx = (int)img.d[(size_t)v * img.w + u + 1] - (int)img.d[(size_t)v * img.w + u - 1];
            const int dy = (int)img.d[(size_t)(v + 1) * img.w + u] - (int)img.d[(size_t)(v - 1) * img.w + u];
            a += dx * dx; b += dx * dy; c += dy * dy;
        }
    const int64_t disc = (a - c) * (a - c) + 4 * b * b;
    return (int32_t)((a + c) - isqrt64(disc));
}

void CornerDetector::set_image_size(int w, int h) {
    cell_h = (h + rows - 1) / rows;
    cell_w = (w + cols - 1) / cols;
}

// image_processor.cpp:647 detector_.set_grid_position(cg::Point2f(x, y))
void CornerDetector::set_grid_position(float x, float y) {
    if (cell_w == 0) return;
    int r = (int)(y / (float)cell_h), c = (int)(x / (float)cell_w);
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    c = c < 0 ? 0 : (c >= cols ? cols - 1 : c);
    occupancy[(size_t)r * cols + c] = 1;
}

void CornerDetector::cell_maxima(const Img &img, std::vector<mskf_corner> &out) const {
    out.assign((size_t)rows * cols, mskf_corner{0.f, 0.f, 0, 0});
    for (size_t i = 0; i < out.size(); ++i) out[i].cell = (int)i;
    for (int y = 0; y < img.h; ++y)
        for (int x = 0; x < img.w; ++x) {
            const int32_t s = shi_tomasi_score(img, x, y);
            const int cell = (y / cell_h) * cols + (x / cell_w);
            if (s > out[cell].score) { out[cell].score = s; out[cell].x = (float)x; out[cell].y = (float)y; }
        }
}

// image_processor.cpp:259,657 detector_.detect_features(img, pts, responses)
void CornerDetector::detect_features(const Img &img, std::vector<mskf_point2f> &pts, std::vector<double> &responses) {
    set_image_size(img.w, img.h);
    std::vector<mskf_corner> mx;
    cell_maxima(img, mx);
    pts.clear(); responses.clear();
    for (size_t i = 0; i < mx.size(); ++i) {
        if (mx[i].score > thr * 256 && !occupancy[i]) {
            pts.push_back(mskf_point2f{mx[i].x, mx[i].y});
            responses.push_back((double)mx[i].score / 256.0);
        }
    }
    std::fill(occupancy.begin(), occupancy.end(), 0);
}

// ------------------------------------------------------------------ K5 point math
// cv::undistortPoints (radtan, 5 fixed-point iterations) then R then P; double internals, float in/out.
void undistort_point(const CamModel &cam, const double R[9], const double Pn[4],
                     float u, float v, float &xo, float &yo) {
    const double fx = cam.K[0], fy = cam.K[1], cx = cam.K[2], cy = cam.K[3];
    double x = ((double)u - cx) / fx, y = ((double)v - cy) / fy;
    if (cam.model == MSKF_MODEL_EQUIDISTANT) {
        // cv::fisheye::undistortPoints
        const double *k = cam.D;
        double theta_d = std::sqrt(x * x + y * y);
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
                if (std::fabs(fix) < 1e-8) break;
            }
            scale = det_tan(theta) / theta_d;     // o_math.h: same operation sequence as the device
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
    const double X = R[0] * x + R[1] * y + R[2];
    const double Y = R[3] * x + R[4] * y + R[5];
    const double W = R[6] * x + R[7] * y + R[8];
    const double xr = X / W, yr = Y / W;
    xo = (float)(xr * Pn[0] + Pn[2]);
    yo = (float)(yr * Pn[1] + Pn[3]);
}

// cg::project_points(pts, rvec=0, tvec=0, K, D) (image_processor.cpp:837): treats (x,y) as the ray (x,y,1)
void distort_point(const CamModel &cam, float xf, float yf, float &uo, float &vo) {
    const double fx = cam.K[0], fy = cam.K[1], cx = cam.K[2], cy = cam.K[3];
    const double x = (double)xf, y = (double)yf;
    double xd, yd;
    if (cam.model == MSKF_MODEL_EQUIDISTANT) {
        const double *k = cam.D;
        const double r = std::sqrt(x * x + y * y);
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

}  // namespace orc
