// fe_book_ransac.cpp — the device source of the 2-point RANSAC (fb_two_point_ransac, msckf_stereo_c_amd/csrc/hip/fe_book.h)
// executed on the CPU behind a C entry point, for tests/test_ransac.py: the same header the bookkeeping kernel is built from
// (phases of independent items; on the host they run one after the other), fed undistorted point pairs, compared there with the
// CPU oracle's twoPointRansac marker for marker.
//
// build: g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC -I<repo> tests/cpp/fe_book_ransac.cpp -o libfe_book_ransac.so
#include <cstring>
#include <vector>
#include "msckf_stereo_c_amd/csrc/hip/fe_book.h"

extern "C" void fb_ransac_run(int n, const mskf_point2f *prev_und, const mskf_point2f *curr_und, const double *R_p_c, double fx, double fy,
                              double inlier_error, int iters, unsigned long long *draws, int32_t *markers) {
    const int cap = n + 8;
    FeBookDev B;
    std::memset(&B, 0, sizeof(B));
    FeBookState st;
    std::memset(&st, 0, sizeof(st));
    st.ransac_draws = *draws;
    B.st = &st;
    B.cap = cap;
    B.ransac = 1; B.ransac_iters = iters; B.ransac_thr = inlier_error;
    B.ransac_npu[0] = B.ransac_npu[1] = 2.0 / (fx + fy);
    std::memcpy(B.R_p_c[0], R_p_c, sizeof(B.R_p_c[0]));
    std::vector<mskf_point2f> prev(prev_und, prev_und + n), curr(curr_und, curr_und + n);
    prev.resize(cap); curr.resize(cap);
    B.prev.und0 = prev.data(); B.t_und0 = curr.data();
    std::vector<double> rs_pair(4 * (size_t)cap), rs_scalar(48);
    std::vector<float> rs_pt(4 * (size_t)cap);
    B.rs_pair = rs_pair.data(); B.rs_pt = rs_pt.data(); B.rs_scalar = rs_scalar.data();
    std::vector<int> scratch(fe_book_scratch_ints(cap, cap, cap, 4, 16), 0x5a5a5a5a);      // (nothing may depend on what the scratch held before)
    FeBookScratch L;
    fe_book_scratch_init(L, scratch.data(), cap, cap, cap, 4, 16);
    std::vector<int> list(cap), mark(cap, 1);
    for (int k = 0; k < n; ++k) list[k] = k;
    fb_two_point_ransac(B, L, 0, n, list.data(), mark.data());
    for (int k = 0; k < n; ++k) markers[k] = mark[k];
    *draws = st.ransac_draws;
}
