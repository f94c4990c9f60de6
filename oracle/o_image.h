// oracle/o_image.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into the product.
//
// CPU statement of the pixel arithmetic the reference takes from the absent vikit_cg
// (cg::pyr_down, cg::optical_flow_multi_level, cg::CornerDetector, cg::undistort_points,
// cg::project_points; call sites image_processor.cpp:132,239,242,259,410,569,647,657,810,837).
// vikit_cg is un-vendored and un-pinned (README.md:9-13, msckf_core/CMakeLists.txt:59,64), so the
// arithmetic below is the published OpenCV algorithm each commented-out cv:: call next to those
// call sites names (image_processor.cpp:217-227, 399-408, 559-567, 809), restated with a fully
// specified fixed-point contract so a GPU can reproduce it bit for bit.
// parity unpinned: the reference holds no golden vectors for any of these functions.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "../include/mskf_types.h"

namespace orc {

struct Img {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Img() {}
    Img(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_, 0) {}
    inline int at(int x, int y) const {  // replicate border
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        return d[(size_t)y * w + x];
    }
};

enum { LK_LEVELS = 4, LK_HALF_WIN = 7, LK_WIN = 15, LK_MAX_ITER = 30 };

// K1: 5-tap [1 4 6 4 1]/16 separable Gaussian, BORDER_REFLECT_101, (sum+128)>>8, dst = ((w+1)/2, (h+1)/2)
void pyr_down(const Img &src, Img &dst);
void build_pyramid(const Img &l0, std::vector<Img> &pyr);  // 4 levels, image_processor.cpp:229-244

// K2/K3: pyramidal LK, OPTFLOW_USE_INITIAL_FLOW semantics, window 15, <=30 iterations, eps 0.01
void lk_track(const std::vector<Img> &pyrA, const std::vector<Img> &pyrB,
              const std::vector<mskf_point2f> &ptsA, std::vector<mskf_point2f> &ptsB,
              std::vector<uint8_t> &status);
void lk_track_point(const std::vector<Img> &pyrA, const std::vector<Img> &pyrB,
                    float ax, float ay, float &bx, float &by, uint8_t &status);

// K4: grid corner detector. One best integer Shi-Tomasi corner per det cell.
struct CornerDetector {
    int rows = 30, cols = 47, thr = 10;
    int cell_w = 0, cell_h = 0;
    std::vector<uint8_t> occupancy;
    CornerDetector() {}
    CornerDetector(int r, int c, int t) : rows(r), cols(c), thr(t), occupancy((size_t)r * c, 0) {}
    void set_image_size(int w, int h);
    void set_grid_position(float x, float y);
    // all per-cell maxima (no threshold / occupancy), cell order
    void cell_maxima(const Img &img, std::vector<mskf_corner> &out) const;
    // thresholded + occupancy-filtered detections in cell order; clears occupancy afterwards
    void detect_features(const Img &img, std::vector<mskf_point2f> &pts, std::vector<double> &responses);
};
int32_t shi_tomasi_score(const Img &img, int x, int y);

// K5: radtan undistort / distort-project (OpenCV undistortPoints / projectPoints with rvec=tvec=0)
struct CamModel {
    double K[4];   // fx fy cx cy
    double D[4];   // k1 k2 p1 p2
    int model;
};
void undistort_point(const CamModel &cam, const double R[9], const double Pnew[4],
                     float u, float v, float &xo, float &yo);
void distort_point(const CamModel &cam, float x, float y, float &uo, float &vo);

extern long long g_lk_stats[3];  // [points, levels solved, iterations] since the last reset (test sizing aid)
}  // namespace orc
