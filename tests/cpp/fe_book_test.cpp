// fe_book_test.cpp — CPU check of the front-end's device bookkeeping (msckf_stereo_c_amd/csrc/hip/fe_book.h).
//
// fe_book.h is written so that the SAME source runs on the host (phases of independent items, no atomics, no cross-lane
// operations): here fe_book1 / fe_book2 are executed on random frames — random previous grids, random track results
// (including points on the image border and in the partial grid rows / columns of quirk Q7), random detector keys with
// many score ties and stale generations, random outcomes of the candidates' stereo match — and compared, frame after
// frame with the state carried over, with a plain restatement of the reference's own flow built on std::map and
// std::stable_sort (image_processor.cpp:416-513 trackFeatures tail, :622-756 addNewFeatures, :758-768 pruneGridFeatures;
// the same structure as oracle/o_frontend.cpp).  Everything is integer / float-exact, so the comparison is bitwise.
//
// build: g++ -O2 -std=c++17 -I<repo> tests/cpp/fe_book_test.cpp -o fe_book_test      run: ./fe_book_test [trials]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>
#include "msckf_stereo_c_amd/csrc/hip/fe_book.h"

typedef unsigned long long u64;
struct Feat { u64 id = 0; float response = 0.f; int lifetime = 0; mskf_point2f cam0{0, 0}, cam1{0, 0}, und0{0, 0}, und1{0, 0}; };
typedef std::map<int, std::vector<Feat>> Grid;

struct Cfg { int W, H, grid_row, grid_col, grid_min, grid_max, det_rows, det_cols, thr, q4; int ransac; double K[2][4], R[2][9], ransac_thr; };

// the product's host implementation of twoPointRansac (csrc/host/image_processor.cpp; == the oracle's, tests/test_ransac.py)
extern "C" void mskfh_two_point_ransac(int n, const mskf_point2f *pts1_und, const mskf_point2f *pts2_und, const double *R_p_c, const double *intrinsics,
                                       double inlier_error, double success_probability, unsigned long long *draws, int32_t *markers);
struct Info { int before = 0, after_tracking = 0, after_matching = 0, after_ransac = 0; };

static unsigned hash2(float x, float y, unsigned salt) {
    unsigned a, b;
    std::memcpy(&a, &x, 4); std::memcpy(&b, &y, 4);
    unsigned h = a * 2654435761u ^ (b + salt) * 40503u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return h;
}
// the (made-up, deterministic) outcome of the stereo match of a candidate point
static void cand_result(mskf_point2f p, unsigned salt, mskf_point2f &o1, mskf_point2f &u0, mskf_point2f &u1, uint8_t &status) {
    const unsigned h = hash2(p.x, p.y, salt);
    status = (uint8_t)(1 | ((h % 10) < 7 ? 2 : 0));
    o1 = mskf_point2f{p.x - 3.25f - (float)(h & 7), p.y + 0.5f};
    u0 = mskf_point2f{p.x * 0.002f - 0.7f, p.y * 0.002f - 0.4f};
    u1 = mskf_point2f{o1.x * 0.002f - 0.7f, o1.y * 0.002f - 0.4f};
}

struct Frame {      // the inputs of one frame
    std::vector<mskf_point2f> t_out0, t_out1, t_und0, t_und1;
    std::vector<uint8_t> t_status;
    std::vector<u64> keys;
    unsigned gen, salt;
};

// ---------------------------------------------------------------------------------------------- reference flow
static void ref_frame(const Cfg &c, const Frame &f, const Grid &prev, Grid &curr, Info &info, u64 &next_id, u64 &ransac_draws,
                      std::vector<mskf_point2f> &cand_sent, std::vector<int> &cand_sent_index) {
    const int grid_height = c.H / c.grid_row, grid_width = c.W / c.grid_col;
    const int det_ch = (c.H + c.det_rows - 1) / c.det_rows, det_cw = (c.W + c.det_cols - 1) / c.det_cols;
    const int n_cells = c.grid_row * c.grid_col;
    curr.clear();
    // trackFeatures (:352-513) after the tracks
    std::vector<Feat> flat;
    for (const auto &it : prev) for (const auto &pf : it.second) flat.push_back(pf);
    info.before = (int)flat.size();
    if (!flat.empty()) {
        info.after_tracking = info.after_matching = info.after_ransac = 0;
        // :482-500 (Q5 cleared): the matched cam0 and cam1 pairs through twoPointRansac, a feature must be an inlier of both
        std::vector<int> keep(flat.size(), 1);
        if (c.ransac) {
            std::vector<size_t> idx;
            std::vector<mskf_point2f> p0, c0, p1, c1;
            for (size_t i = 0; i < flat.size(); ++i) {
                if ((f.t_status[i] & 3) != 3) continue;
                idx.push_back(i);
                p0.push_back(flat[i].und0); p1.push_back(flat[i].und1); c0.push_back(f.t_und0[i]); c1.push_back(f.t_und1[i]);
            }
            std::vector<int32_t> in0(idx.size() + 1), in1(idx.size() + 1);
            mskfh_two_point_ransac((int)idx.size(), p0.data(), c0.data(), c.R[0], c.K[0], c.ransac_thr, 0.99, &ransac_draws, in0.data());
            mskfh_two_point_ransac((int)idx.size(), p1.data(), c1.data(), c.R[1], c.K[1], c.ransac_thr, 0.99, &ransac_draws, in1.data());
            for (size_t k = 0; k < idx.size(); ++k) keep[idx[k]] = in0[k] != 0 && in1[k] != 0;
        }
        for (size_t i = 0; i < flat.size(); ++i) {
            if (!(f.t_status[i] & 1)) continue;
            ++info.after_tracking;
            if (!(f.t_status[i] & 2)) continue;
            ++info.after_matching;
            if (!keep[i]) continue;
            const int row = static_cast<int>(f.t_out0[i].y / grid_height), col = static_cast<int>(f.t_out0[i].x / grid_width);
            const int code = row * c.grid_col + col;
            Feat g = flat[i];
            g.lifetime = flat[i].lifetime + 1;
            g.response = 0.f;
            g.cam0 = f.t_out0[i]; g.cam1 = f.t_out1[i]; g.und0 = f.t_und0[i]; g.und1 = f.t_und1[i];
            curr[code].push_back(g);
            ++info.after_ransac;
        }
    }
    // addNewFeatures (:622-756)
    std::vector<uint8_t> occ((size_t)c.det_rows * c.det_cols, 0);
    for (const auto &it : curr)
        for (const auto &ft : it.second) {
            const int y = static_cast<int>(ft.cam0.y), x = static_cast<int>(ft.cam0.x);
            int r = (int)((float)y / (float)det_ch), cc = (int)((float)x / (float)det_cw);
            r = r < 0 ? 0 : (r >= c.det_rows ? c.det_rows - 1 : r);
            cc = cc < 0 ? 0 : (cc >= c.det_cols ? c.det_cols - 1 : cc);
            occ[(size_t)r * c.det_cols + cc] = 1;
        }
    std::vector<mskf_point2f> new_features;
    std::vector<double> new_features_responses;
    for (int k = 0; k < c.det_rows * c.det_cols; ++k) {
        const u64 key = f.keys[k];
        if ((unsigned)(key >> 56) != f.gen) continue;
        const int score = (int)((key >> 32) & 0xFFFFFFULL);
        if (score <= c.thr || occ[k]) continue;
        const unsigned order = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFULL);
        const int cy = k / c.det_cols, cx = k - cy * c.det_cols;
        new_features.push_back(mskf_point2f{(float)(cx * det_cw + (int)(order % (unsigned)det_cw)), (float)(cy * det_ch + (int)(order / (unsigned)det_cw))});
        new_features_responses.push_back((double)score / 256.0);
    }
    std::vector<std::vector<std::pair<mskf_point2f, double>>> sieve((size_t)n_cells);
    for (size_t i = 0; i < new_features.size(); ++i) {
        const int row = static_cast<int>(new_features[i].y / grid_height), col = static_cast<int>(new_features[i].x / grid_width);
        const size_t code = (size_t)(row * c.grid_col + col);
        if (code >= sieve.size()) continue;
        sieve[code].push_back(std::make_pair(new_features[i], new_features_responses[i]));
    }
    std::vector<mskf_point2f> cand;
    std::vector<double> sieved_responses;
    std::vector<int> cand_code;
    for (size_t code = 0; code < sieve.size(); ++code) {
        auto &item = sieve[code];
        if ((int)item.size() > c.grid_max) {
            std::stable_sort(item.begin(), item.end(), [](const std::pair<mskf_point2f, double> &a, const std::pair<mskf_point2f, double> &b) { return a.second > b.second; });
            item.erase(item.begin() + c.grid_max, item.end());
        }
        for (const auto &p : item) { cand.push_back(p.first); sieved_responses.push_back(p.second); cand_code.push_back((int)code); }
    }
    // what the device sends to the second track call: the candidates of the cells with a vacancy, with their position in this list
    cand_sent.clear(); cand_sent_index.clear();
    for (size_t i = 0; i < cand.size(); ++i) {
        const int have = curr.count(cand_code[i]) ? (int)curr[cand_code[i]].size() : 0;
        if (have < c.grid_min) { cand_sent.push_back(cand[i]); cand_sent_index.push_back((int)i); }
    }
    std::map<int, std::vector<Feat>> grid_new;
    for (int code = 0; code < n_cells; ++code) grid_new[code] = std::vector<Feat>();
    for (size_t i = 0; i < cand.size(); ++i) {
        mskf_point2f o1, u0, u1; uint8_t st;
        cand_result(cand[i], f.salt, o1, u0, u1, st);
        if (!(st & 2)) continue;
        Feat nf;
        nf.response = (float)(c.q4 ? new_features_responses[i] : sieved_responses[i]);      // Q4 (:698)
        nf.cam0 = cand[i]; nf.cam1 = o1; nf.und0 = u0; nf.und1 = u1;
        const int row = static_cast<int>(cand[i].y / grid_height), col = static_cast<int>(cand[i].x / grid_width);
        grid_new[row * c.grid_col + col].push_back(nf);
    }
    for (auto &it : grid_new) std::stable_sort(it.second.begin(), it.second.end(), [](const Feat &a, const Feat &b) { return a.response > b.response; });
    for (int code = 0; code < n_cells; ++code) {
        std::vector<Feat> &here = curr[code];
        std::vector<Feat> &fresh = grid_new[code];
        if ((int)here.size() >= c.grid_min) continue;
        const int vacancy = c.grid_min - (int)here.size();
        for (int k = 0; k < vacancy && k < (int)fresh.size(); ++k) {
            here.push_back(fresh[k]);
            here.back().id = next_id++;
            here.back().lifetime = 1;
        }
    }
    // pruneGridFeatures (:758-768)
    for (auto &it : curr) {
        auto &g = it.second;
        if ((int)g.size() <= c.grid_max) continue;
        std::stable_sort(g.begin(), g.end(), [](const Feat &a, const Feat &b) { return a.lifetime > b.lifetime; });
        g.erase(g.begin() + c.grid_max, g.end());
    }
}

// ---------------------------------------------------------------------------------------------- device-logic side
struct HostGrid {
    std::vector<u64> id; std::vector<int> lifetime, code; std::vector<float> response; std::vector<mskf_point2f> cam0, cam1, und0, und1;
    void resize(int n) { id.resize(n); lifetime.resize(n); code.resize(n); response.resize(n); cam0.resize(n); cam1.resize(n); und0.resize(n); und1.resize(n); }
    FeGridArr arr() { return FeGridArr{id.data(), lifetime.data(), code.data(), response.data(), cam0.data(), cam1.data(), und0.data(), und1.data()}; }
};

static bool same_pt(mskf_point2f a, mskf_point2f b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

int main(int argc, char **argv) {
    const int trials = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937 rng(12345);
    auto U = [&](int lo, int hi) { return (int)(rng() % (unsigned)(hi - lo + 1)) + lo; };
    long frames_checked = 0, feats_checked = 0, pruned_cells = 0, cand_total = 0, ransac_rejected = 0, draws_total = 0;
    for (int trial = 0; trial < trials; ++trial) {
        Cfg c;
        const int sizes[][2] = {{752, 480}, {376, 240}, {333, 251}, {1280, 720}, {640, 400}};
        const int si = U(0, 4);
        c.W = sizes[si][0]; c.H = sizes[si][1];
        c.grid_row = U(2, 10); c.grid_col = U(2, 12);
        c.grid_min = U(1, 6); c.grid_max = c.grid_min + U(0, 3);
        c.det_rows = 30; c.det_cols = 47;
        c.thr = 10 * 256; c.q4 = U(0, 1);
        // every other trial runs the 2-point RANSAC between the tracks: cameras with EuRoC-like focal lengths, a small rotation
        c.ransac = trial & 1; c.ransac_thr = 3.0;
        for (int cam = 0; cam < 2; ++cam) {
            c.K[cam][0] = 458.654 - 1.2 * cam; c.K[cam][1] = 457.296 - 0.8 * cam; c.K[cam][2] = 367.215; c.K[cam][3] = 248.375;
            const double wx = 1e-3 * U(-5, 5), wy = 1e-3 * U(-5, 5), wz = 1e-3 * U(-5, 5);
            const double Rm[9] = {1.0, -wz, wy, wz, 1.0, -wx, -wy, wx, 1.0};
            std::memcpy(c.R[cam], Rm, sizeof(Rm));
        }
        const int grid_h = c.H / c.grid_row, grid_w = c.W / c.grid_col;
        const int det_ch = (c.H + c.det_rows - 1) / c.det_rows, det_cw = (c.W + c.det_cols - 1) / c.det_cols;
        const int n_cells = c.grid_row * c.grid_col;
        const int n_codes = std::max(((c.H - 1) / grid_h) * c.grid_col + (c.W - 1) / grid_w + 1, n_cells);
        const int det_cells = c.det_rows * c.det_cols;
        const int cap = n_codes * c.grid_max + 8, cand_cap = n_cells * c.grid_max + 8, det_cap = det_cells;
        // device-side state
        HostGrid g[3];
        for (auto &x : g) x.resize(cap);
        FeBookState st;
        std::memset(&st, 0, sizeof(st));
        std::vector<int> scratch(fe_book_scratch_ints(cap, cand_cap, det_cap, n_codes, det_cells));
        std::vector<mskf_point2f> det_pt(det_cap), cand_pt(cand_cap), c_out0(cand_cap), c_out1(cand_cap), c_und0(cand_cap), c_und1(cand_cap);
        std::vector<int> det_score(det_cap), cand_index(cand_cap), cand_score(cand_cap), cand_off(n_cells + 1), cand_cnt(n_cells + 1), cell_count(n_codes + 1);
        std::vector<uint8_t> c_status(cand_cap);
        std::vector<int> x_info(16); std::vector<u64> x_id(cap); std::vector<int> x_life(cap);
        std::vector<mskf_point2f> x_c0(cap), x_c1(cap), x_u0(cap), x_u1(cap);
        int ip = 0;     // index of the "prev" grid among g[0], g[1]; g[2] is the tracked list
        // reference-side state
        Grid prev, curr;
        Info info;
        u64 next_id = 0, ref_draws = 0;
        std::vector<double> rs_pair(4 * (size_t)cap), rs_scalar(48);
        std::vector<float> rs_pt(4 * (size_t)cap);
        const int n_frames = U(3, 8);
        for (int fr = 0; fr < n_frames; ++fr) {
            Frame f;
            f.gen = (unsigned)(fr % 255) + 1; f.salt = rng();
            std::vector<Feat> flat;
            for (const auto &it : prev) for (const auto &pf : it.second) flat.push_back(pf);
            const int n = (int)flat.size();
            if (n != st.n_prev) { std::printf("FAIL trial %d frame %d: n_prev %d vs %d\n", trial, fr, st.n_prev, n); return 1; }
            const int loss = U(0, 100);       // percent of the features this frame loses (some frames lose everything)
            // RANSAC trials: the undistorted points move by a common flow (0: pure rotation, the degenerate branch) plus noise, a
            // few of them wildly; otherwise they are unrelated to the previous frame's
            const float flow_x = c.ransac ? 0.0011f * (float)U(-4, 4) : 0.f, flow_y = c.ransac ? 0.0009f * (float)U(-4, 4) : 0.f;
            f.t_out0.resize(n); f.t_out1.resize(n); f.t_und0.resize(n); f.t_und1.resize(n); f.t_status.resize(n);
            for (int i = 0; i < n; ++i) {
                const int r = U(0, 99);
                f.t_status[i] = (uint8_t)(r < loss / 2 ? 0 : (r < loss ? 1 : 3));
                const int kind = U(0, 19);
                float x = (float)U(0, c.W - 2) + (float)U(0, 1023) / 1024.f, y = (float)U(0, c.H - 2) + (float)U(0, 1023) / 1024.f;
                if (kind == 0) x = (float)(c.W - 1);
                if (kind == 1) y = (float)(c.H - 1);
                if (kind == 2) { x = 0.f; y = 0.f; }
                if (kind == 3) x = (float)(c.grid_col * grid_w) + 0.25f < (float)(c.W - 1) ? (float)(c.grid_col * grid_w) + 0.25f : x;   // Q7: column == grid_col
                f.t_out0[i] = mskf_point2f{x, y};
                f.t_out1[i] = mskf_point2f{x - 5.5f, y + 0.125f};
                f.t_und0[i] = mskf_point2f{x * 0.001f, y * 0.001f};
                f.t_und1[i] = mskf_point2f{x * 0.001f - 0.01f, y * 0.001f};
                if (c.ransac) {
                    const float wild = U(0, 9) == 0 ? 0.02f * (float)U(-3, 3) : 0.f;
                    const float zx = flat[i].und0.x * 0.1f * (float)U(0, 3) * flow_x, zy = flat[i].und0.y * 0.1f * (float)U(0, 3) * flow_y;    // depth-like spread along the flow
                    f.t_und0[i] = mskf_point2f{flat[i].und0.x + flow_x + zx + 1e-5f * (float)U(-20, 20) + wild, flat[i].und0.y + flow_y + zy + 1e-5f * (float)U(-20, 20)};
                    f.t_und1[i] = mskf_point2f{flat[i].und1.x + flow_x + zx + 1e-5f * (float)U(-20, 20), flat[i].und1.y + flow_y + zy + 1e-5f * (float)U(-20, 20) - wild};
                }
                if (f.t_status[i] != 3) { f.t_out1[i] = mskf_point2f{0, 0}; }
            }
            f.keys.assign(det_cells, 0ULL);
            const int density = U(0, 100);
            for (int k = 0; k < det_cells; ++k) {
                if (U(0, 99) >= density) continue;
                const int cy = k / c.det_cols, cx = k - cy * c.det_cols;
                const int x0 = cx * det_cw, y0 = cy * det_ch;
                if (x0 >= c.W || y0 >= c.H) continue;
                const int ox = U(0, std::min(det_cw, c.W - x0) - 1), oy = U(0, std::min(det_ch, c.H - y0) - 1);
                const unsigned order = (unsigned)(oy * det_cw + ox);
                const int score = U(0, 3) == 0 ? c.thr + U(-2, 2) : c.thr + 256 * U(1, 6);       // few distinct values: ties everywhere
                const unsigned gen = U(0, 9) == 0 ? ((f.gen + 7) % 255) + 1 : f.gen;           // some keys are stale
                f.keys[k] = ((u64)gen << 56) | ((u64)(unsigned)score << 32) | (0xFFFFFFFFu - order);
            }
            // ---- reference
            std::vector<mskf_point2f> ref_cand; std::vector<int> ref_cand_index;
            ref_frame(c, f, prev, curr, info, next_id, ref_draws, ref_cand, ref_cand_index);
            // ---- device logic
            FeBookDev B;
            std::memset(&B, 0, sizeof(B));
            B.grid_row = c.grid_row; B.grid_col = c.grid_col; B.grid_min = c.grid_min; B.grid_max = c.grid_max; B.n_codes = n_codes; B.n_cells = n_cells;
            B.grid_w = grid_w; B.grid_h = grid_h; B.det_rows = c.det_rows; B.det_cols = c.det_cols; B.det_cw = det_cw; B.det_ch = det_ch;
            B.thr_score = c.thr; B.q4 = c.q4; B.cap = cap; B.cand_cap = cand_cap; B.det_cap = det_cap; B.gen = f.gen; B.st = &st;
            B.prev = g[ip].arr(); B.curr = g[ip ^ 1].arr(); B.tracked = g[2].arr();
            B.t_out0 = f.t_out0.data(); B.t_out1 = f.t_out1.data(); B.t_und0 = f.t_und0.data(); B.t_und1 = f.t_und1.data(); B.t_status = f.t_status.data();
            B.cell_keys = f.keys.data(); B.det_pt = det_pt.data(); B.det_score = det_score.data();
            B.cand_pt = cand_pt.data(); B.cand_index = cand_index.data(); B.cand_score = cand_score.data(); B.cand_off = cand_off.data(); B.cand_cnt = cand_cnt.data();
            B.c_out0 = c_out0.data(); B.c_out1 = c_out1.data(); B.c_und0 = c_und0.data(); B.c_und1 = c_und1.data(); B.c_status = c_status.data();
            B.cell_count = cell_count.data();
            B.ransac = c.ransac; B.ransac_iters = 7; B.ransac_thr = c.ransac_thr;
            for (int cam = 0; cam < 2; ++cam) { B.ransac_npu[cam] = 2.0 / (c.K[cam][0] + c.K[cam][1]); std::memcpy(B.R_p_c[cam], c.R[cam], sizeof(B.R_p_c[cam])); }
            B.rs_pair = rs_pair.data(); B.rs_pt = rs_pt.data(); B.rs_scalar = rs_scalar.data();
            B.x_info = x_info.data(); B.x_id = x_id.data(); B.x_lifetime = x_life.data(); B.x_cam0 = x_c0.data(); B.x_cam1 = x_c1.data(); B.x_und0 = x_u0.data(); B.x_und1 = x_u1.data();
            FeBookScratch L;
            fe_book_scratch_init(L, scratch.data(), cap, cand_cap, det_cap, n_codes, det_cells);
            std::fill(scratch.begin(), scratch.end(), 0x5a5a5a5a);      // (nothing may depend on what the scratch held before)
            fe_book1(B, L);
            // candidates: the reference's list restricted to the cells with a vacancy, same order, same positions
            if (st.n_cand != (int)ref_cand.size()) { std::printf("FAIL trial %d frame %d: %d candidates vs %d\n", trial, fr, st.n_cand, (int)ref_cand.size()); return 1; }
            for (int i = 0; i < st.n_cand; ++i) {
                if (!same_pt(cand_pt[i], ref_cand[i]) || cand_index[i] != ref_cand_index[i]) { std::printf("FAIL trial %d frame %d: candidate %d differs: pt (%g,%g) vs (%g,%g), index %d vs %d, score %d\n", trial, fr, i, cand_pt[i].x, cand_pt[i].y, ref_cand[i].x, ref_cand[i].y, cand_index[i], ref_cand_index[i], cand_score[i]); return 1; }
                mskf_point2f o1, u0, u1; uint8_t s;
                cand_result(cand_pt[i], f.salt, o1, u0, u1, s);
                c_out0[i] = cand_pt[i]; c_out1[i] = o1; c_und0[i] = u0; c_und1[i] = u1; c_status[i] = s;
            }
            cand_total += st.n_cand;
            std::fill(scratch.begin(), scratch.end(), 0x3c3c3c3c);
            fe_book2(B, L);
            // ---- compare the published grid, the id counter and the tracking info
            std::vector<Feat> want; std::vector<int> want_code;
            for (const auto &it : curr) {
                for (const auto &ft : it.second) { want.push_back(ft); want_code.push_back(it.first); }
            }
            if (st.overflow) { std::printf("FAIL trial %d frame %d: overflow flag\n", trial, fr); return 1; }
            if (st.n_curr != (int)want.size() || x_info[0] != st.n_curr) { std::printf("FAIL trial %d frame %d: %d features vs %d\n", trial, fr, st.n_curr, (int)want.size()); return 1; }
            for (int i = 0; i < st.n_curr; ++i) {
                const HostGrid &G = g[ip ^ 1];
                const bool ok = G.id[i] == want[i].id && G.lifetime[i] == want[i].lifetime && G.code[i] == want_code[i] &&
                                std::memcmp(&G.response[i], &want[i].response, 4) == 0 && same_pt(G.cam0[i], want[i].cam0) && same_pt(G.cam1[i], want[i].cam1) &&
                                same_pt(G.und0[i], want[i].und0) && same_pt(G.und1[i], want[i].und1) &&
                                x_id[i] == want[i].id && x_life[i] == want[i].lifetime && same_pt(x_c0[i], want[i].cam0) && same_pt(x_c1[i], want[i].cam1) &&
                                same_pt(x_u0[i], want[i].und0) && same_pt(x_u1[i], want[i].und1);
                if (!ok) {
                    std::printf("FAIL trial %d frame %d: feature %d differs (id %llu vs %llu, life %d vs %d, code %d vs %d)\n", trial, fr, i, G.id[i], want[i].id,
                                G.lifetime[i], want[i].lifetime, G.code[i], want_code[i]);
                    return 1;
                }
            }
            if (st.ransac_draws != ref_draws) { std::printf("FAIL trial %d frame %d: RANSAC draw counter %llu vs %llu\n", trial, fr, st.ransac_draws, ref_draws); return 1; }
            ransac_rejected += info.after_matching - info.after_ransac;
            if (fr == n_frames - 1) draws_total += (long)ref_draws;
            if (st.next_id != next_id) { std::printf("FAIL trial %d frame %d: next id %llu vs %llu\n", trial, fr, st.next_id, next_id); return 1; }
            if (st.before_tracking != info.before || st.after_tracking != info.after_tracking || st.after_matching != info.after_matching ||
                st.after_ransac != info.after_ransac) { std::printf("FAIL trial %d frame %d: tracking info\n", trial, fr); return 1; }
            for (const auto &it : curr) pruned_cells += (int)it.second.size() == c.grid_max ? 1 : 0;
            feats_checked += st.n_curr; ++frames_checked;
            prev = curr;
            ip ^= 1;
        }
    }
    std::printf("fe_book: %ld frames, %ld features, %ld candidates, %ld full cells, %ld RANSAC rejections (%ld numbers drawn, counter checked every frame): device logic == reference flow\n",
                frames_checked, feats_checked, cand_total, pruned_cells, ransac_rejected, draws_total);
    return 0;
}
