// oracle/o_frontend.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).  See o_frontend.h.
#include "o_frontend.h"
#include <algorithm>
#include <cmath>
#include <cstdio>

namespace orc {

// image_processor.cpp:32-124 (ctor + loadParameters) and :126-137 (initialize)
ImageProcessor::ImageProcessor(const mskf_calib &calib, const mskf_fe_cfg &cfg)
    : feature_msg_ptr_(new CameraMeasurement), calib_(calib), cfg_(cfg),
      detector_(cfg.det_rows, cfg.det_cols, cfg.fast_threshold),
      prev_features_ptr(new GridFeatures()), curr_features_ptr(new GridFeatures()) {
    for (int i = 0; i < 4; ++i) {
        cam0_.K[i] = calib.cam0_intrinsics[i]; cam0_.D[i] = calib.cam0_distortion[i];
        cam1_.K[i] = calib.cam1_intrinsics[i]; cam1_.D[i] = calib.cam1_distortion[i];
    }
    cam0_.model = calib.cam0_model; cam1_.model = calib.cam1_model;
    // :63-72
    SE3 m4_cam0_imu = SE3::from16(calib.T_cam0_imu);
    R_cam0_imu = m4_cam0_imu.R.t();
    t_cam0_imu = -(R_cam0_imu * m4_cam0_imu.t);
    SE3 m4_cam1_cam0 = SE3::from16(calib.T_cam1_cam0);
    SE3 T_cam1_imu = m4_cam1_cam0 * m4_cam0_imu;
    R_cam1_imu = T_cam1_imu.R.t();
    t_cam1_imu = -(R_cam1_imu * T_cam1_imu.t);
}

// :205-211
void ImageProcessor::imuCallback(const mskf_imu_sample &msg) {
    if (is_first_img) return;
    imu_msg_buffer.push_back(msg);
}

// :139-203
void ImageProcessor::stereoCallback(const Img &cam0, const Img &cam1, double t0, double t1) {
    cam0_curr_time = t0;
    (void)t1;
    if (cfg_.compat_flags & MSKF_COMPAT_Q2_PREV_ALIAS) {
        // Q2: from the second frame on prev and curr image objects are the same (:192), so the
        // "previous" timestamp read during this callback is already the current one.
        if (!is_first_img) cam0_prev_time = t0;
    }
    cam0_curr_img = cam0;
    cam1_curr_img = cam1;
    if (grid_height == 0) {  // Q7 function-local statics: first image wins
        grid_height = cam0.h / cfg_.grid_row;
        grid_width = cam0.w / cfg_.grid_col;
    }
    createImagePyramids();
    if (is_first_img) {
        // first call: curr grid has no pre-created keys (ctor leaves it empty, :40)
        initializeFirstFrame();
        is_first_img = false;
    } else {
        trackFeatures();
        addNewFeatures();
        pruneGridFeatures();
    }
    // debug dump (the reference exports the same maps when is_draw, :163-184)
    last_dump = FrameDump();
    for (const auto &g : *curr_features_ptr)
        for (const auto &f : g.second) {
            last_dump.ids.push_back(f.id);
            last_dump.lifetime.push_back(f.lifetime);
            last_dump.cam0.push_back(f.cam0_point);
            last_dump.cam1.push_back(f.cam1_point);
        }
    publish();
    last_dump.info = mskf_tracking_info{t0, before_tracking, after_tracking, after_matching, after_ransac};
    // :192-200
    if (!(cfg_.compat_flags & MSKF_COMPAT_Q2_PREV_ALIAS)) cam0_prev_time = t0;
    prev_features_ptr = curr_features_ptr;
    std::swap(prev_cam0_pyramid_, curr_cam0_pyramid_);
    curr_features_ptr.reset(new GridFeatures());
    for (int code = 0; code < cfg_.grid_row * cfg_.grid_col; ++code) (*curr_features_ptr)[code] = std::vector<FeatureMetaData>(0);
}

// :213-245
void ImageProcessor::createImagePyramids() {
    build_pyramid(cam0_curr_img, curr_cam0_pyramid_);
    build_pyramid(cam1_curr_img, curr_cam1_pyramid_);
}

static bool featureCompareByResponse(const FeatureMetaData &a, const FeatureMetaData &b) { return a.response > b.response; }
static bool featureCompareByLifetime(const FeatureMetaData &a, const FeatureMetaData &b) { return a.lifetime > b.lifetime; }

// :247-319
void ImageProcessor::initializeFirstFrame() {
    const Img &img = cam0_curr_img;
    std::vector<mskf_point2f> new_features;
    std::vector<double> new_features_responses;
    detector_.detect_features(img, new_features, new_features_responses);

    std::vector<mskf_point2f> cam0_points = new_features;
    std::vector<mskf_point2f> cam1_points;
    std::vector<uint8_t> inlier_markers;
    stereoMatch(cam0_points, cam1_points, inlier_markers);

    std::vector<mskf_point2f> cam0_inliers, cam1_inliers;
    std::vector<float> response_inliers;
    for (size_t i = 0; i < inlier_markers.size(); ++i) {
        if (inlier_markers[i] == 0) continue;
        cam0_inliers.push_back(cam0_points[i]);
        cam1_inliers.push_back(cam1_points[i]);
        response_inliers.push_back((float)new_features_responses[i]);
    }
    GridFeatures grid_new_features;
    for (int code = 0; code < cfg_.grid_row * cfg_.grid_col; ++code) grid_new_features[code] = std::vector<FeatureMetaData>(0);
    for (size_t i = 0; i < cam0_inliers.size(); ++i) {
        int row = static_cast<int>(cam0_inliers[i].y / grid_height);
        int col = static_cast<int>(cam0_inliers[i].x / grid_width);
        int code = row * cfg_.grid_col + col;
        FeatureMetaData nf;
        nf.response = response_inliers[i];
        nf.cam0_point = cam0_inliers[i];
        nf.cam1_point = cam1_inliers[i];
        grid_new_features[code].push_back(nf);
    }
    for (auto &item : grid_new_features)  // Q19: stable sort is the defined behaviour
        std::stable_sort(item.second.begin(), item.second.end(), featureCompareByResponse);
    for (int code = 0; code < cfg_.grid_row * cfg_.grid_col; ++code) {
        std::vector<FeatureMetaData> &features_this_grid = (*curr_features_ptr)[code];
        std::vector<FeatureMetaData> &new_features_this_grid = grid_new_features[code];
        for (int k = 0; k < cfg_.grid_min_feature_num && k < (int)new_features_this_grid.size(); ++k) {
            features_this_grid.push_back(new_features_this_grid[k]);
            features_this_grid.back().id = next_feature_id++;
            features_this_grid.back().lifetime = 1;
        }
    }
}

// :321-350   H = K R K^-1
void ImageProcessor::predictFeatureTracking(const std::vector<mskf_point2f> &in, const M3 &R_p_c, const double intr[4],
                                            std::vector<mskf_point2f> &out) {
    if (in.empty()) { out.clear(); return; }
    out.resize(in.size());
    M3 K; K(0, 0) = intr[0]; K(0, 2) = intr[2]; K(1, 1) = intr[1]; K(1, 2) = intr[3]; K(2, 2) = 1.0;
    M3 Ki; Ki(0, 0) = 1.0 / intr[0]; Ki(0, 2) = -intr[2] / intr[0]; Ki(1, 1) = 1.0 / intr[1]; Ki(1, 2) = -intr[3] / intr[1]; Ki(2, 2) = 1.0;
    M3 H = K * R_p_c * Ki;
    for (size_t i = 0; i < in.size(); ++i) {
        V3 p1((double)in[i].x, (double)in[i].y, 1.0);
        V3 p2 = H * p1;
        out[i].x = (float)(p2[0] / p2[2]);
        out[i].y = (float)(p2[1] / p2[2]);
    }
}

// :352-532
void ImageProcessor::trackFeatures() {
    M3 cam0_R_p_c, cam1_R_p_c;
    integrateImuData(cam0_R_p_c, cam1_R_p_c);

    std::vector<unsigned long long> prev_ids;
    std::vector<int> prev_lifetime;
    std::vector<mskf_point2f> prev_cam0_points, prev_cam1_points;
    for (const auto &item : *prev_features_ptr)
        for (const auto &pf : item.second) {
            prev_ids.push_back(pf.id);
            prev_lifetime.push_back(pf.lifetime);
            prev_cam0_points.push_back(pf.cam0_point);
            prev_cam1_points.push_back(pf.cam1_point);
        }
    before_tracking = (int)prev_cam0_points.size();
    if (prev_ids.empty()) return;

    std::vector<mskf_point2f> curr_cam0_points;
    std::vector<uint8_t> track_inliers;
    predictFeatureTracking(prev_cam0_points, cam0_R_p_c, cam0_.K, curr_cam0_points);
    lk_track(prev_cam0_pyramid_, curr_cam0_pyramid_, prev_cam0_points, curr_cam0_points, track_inliers);

    const int rows = cam0_curr_img.h, cols = cam0_curr_img.w;
    for (size_t i = 0; i < curr_cam0_points.size(); ++i) {
        if (track_inliers[i] == 0) continue;
        if (curr_cam0_points[i].y < 0 || curr_cam0_points[i].y > rows - 1 ||
            curr_cam0_points[i].x < 0 || curr_cam0_points[i].x > cols - 1)
            track_inliers[i] = 0;
    }
    std::vector<unsigned long long> prev_tracked_ids;
    std::vector<int> prev_tracked_lifetime;
    std::vector<mskf_point2f> prev_tracked_cam0, prev_tracked_cam1, curr_tracked_cam0;
    for (size_t i = 0; i < track_inliers.size(); ++i) {
        if (!track_inliers[i]) continue;
        prev_tracked_ids.push_back(prev_ids[i]);
        prev_tracked_lifetime.push_back(prev_lifetime[i]);
        prev_tracked_cam0.push_back(prev_cam0_points[i]);
        prev_tracked_cam1.push_back(prev_cam1_points[i]);
        curr_tracked_cam0.push_back(curr_cam0_points[i]);
    }
    after_tracking = (int)curr_tracked_cam0.size();

    std::vector<mskf_point2f> curr_cam1_points;
    std::vector<uint8_t> match_inliers;
    stereoMatch(curr_tracked_cam0, curr_cam1_points, match_inliers);

    std::vector<unsigned long long> prev_matched_ids;
    std::vector<int> prev_matched_lifetime;
    std::vector<mskf_point2f> curr_matched_cam0, curr_matched_cam1;
    for (size_t i = 0; i < match_inliers.size(); ++i) {
        if (!match_inliers[i]) continue;
        prev_matched_ids.push_back(prev_tracked_ids[i]);
        prev_matched_lifetime.push_back(prev_tracked_lifetime[i]);
        curr_matched_cam0.push_back(curr_tracked_cam0[i]);
        curr_matched_cam1.push_back(curr_cam1_points[i]);
    }
    after_matching = (int)curr_matched_cam0.size();

    // Q5: both twoPointRansac calls are commented out (:482-500); with the switch cleared they run as written there
    std::vector<int> cam0_ransac_inliers, cam1_ransac_inliers;
    const bool ransac = !(cfg_.compat_flags & MSKF_COMPAT_Q5_NO_RANSAC);
    if (ransac) {
        std::vector<mskf_point2f> prev_matched_cam0, prev_matched_cam1;
        for (size_t i = 0; i < match_inliers.size(); ++i) {
            if (!match_inliers[i]) continue;
            prev_matched_cam0.push_back(prev_tracked_cam0[i]);
            prev_matched_cam1.push_back(prev_tracked_cam1[i]);
        }
        twoPointRansac(prev_matched_cam0, curr_matched_cam0, cam0_R_p_c, cam0_, cfg_.ransac_threshold, 0.99, cam0_ransac_inliers);
        twoPointRansac(prev_matched_cam1, curr_matched_cam1, cam1_R_p_c, cam1_, cfg_.ransac_threshold, 0.99, cam1_ransac_inliers);
    }
    after_ransac = 0;
    for (size_t i = 0; i < curr_matched_cam0.size(); ++i) {
        if (ransac && (cam0_ransac_inliers[i] == 0 || cam1_ransac_inliers[i] == 0)) continue;
        int row = static_cast<int>(curr_matched_cam0[i].y / grid_height);
        int col = static_cast<int>(curr_matched_cam0[i].x / grid_width);
        int code = row * cfg_.grid_col + col;  // Q7: col may equal grid_col
        (*curr_features_ptr)[code].push_back(FeatureMetaData());
        FeatureMetaData &g = (*curr_features_ptr)[code].back();
        g.id = prev_matched_ids[i];
        g.lifetime = ++prev_matched_lifetime[i];
        g.cam0_point = curr_matched_cam0[i];
        g.cam1_point = curr_matched_cam1[i];
        ++after_ransac;
    }
}

// Counter-based replacement of cg::uniform_integer(lo, hi) (vikit_cg, absent): splitmix64 of a per-processor draw
// counter, reduced to [lo, hi].  The product's host mirror uses the same definition (image_processor.cpp there).
int ImageProcessor::uniformInteger(int lo, int hi) {
    unsigned long long z = 0x5EED5EED5EED5EEDULL + 0x9E3779B97F4A7C15ULL * (++ransac_draws);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return lo + (int)(z % (unsigned long long)(hi - lo + 1));
}

// :888-908, float arithmetic as written (Point2f members are float)
void ImageProcessor::rescalePoints(std::vector<mskf_point2f> &pts1, std::vector<mskf_point2f> &pts2, float &scaling_factor) {
    scaling_factor = 0.0f;
    for (size_t i = 0; i < pts1.size(); ++i) {
        scaling_factor += std::sqrt(pts1[i].x * pts1[i].x + pts1[i].y * pts1[i].y);
        scaling_factor += std::sqrt(pts2[i].x * pts2[i].x + pts2[i].y * pts2[i].y);
    }
    scaling_factor = (float)(pts1.size() + pts2.size()) / scaling_factor * std::sqrt(2.0f);
    for (size_t i = 0; i < pts1.size(); ++i) {
        pts1[i].x *= scaling_factor; pts1[i].y *= scaling_factor;
        pts2[i].x *= scaling_factor; pts2[i].y *= scaling_factor;
    }
}

namespace {
// x of [a b] x = rhs with columns a, b of two entries: inverse by adjugate / determinant, then the product
inline void solve2(const double a[2], const double b[2], const double rhs[2], double x[2]) {
    const double det = a[0] * b[1] - b[0] * a[1];
    const double i00 = b[1] / det, i01 = -b[0] / det, i10 = -a[1] / det, i11 = a[0] / det;
    x[0] = i00 * rhs[0] + i01 * rhs[1];
    x[1] = i10 * rhs[0] + i11 * rhs[1];
}
}  // namespace

// :911-1135.  Float where the reference holds cg::Point2f, double elsewhere; sums run in index order.
void ImageProcessor::twoPointRansac(const std::vector<mskf_point2f> &pts1, const std::vector<mskf_point2f> &pts2, const M3 &R_p_c,
                                    const CamModel &cam, double inlier_error, double success_probability,
                                    std::vector<int> &inlier_markers) {
    const size_t n = pts1.size();
    double norm_pixel_unit = 2.0 / (cam.K[0] + cam.K[1]);
    const int iter_num = static_cast<int>(std::ceil(std::log(1 - success_probability) / std::log(1 - 0.7 * 0.7)));
    inlier_markers.assign(n, 1);
    if (n == 0) return;
    std::vector<mskf_point2f> p1(n), p2(n);
    undistortPoints(pts1, cam, p1);
    undistortPoints(pts2, cam, p2);
    for (auto &pt : p1) {   // :938-944, no perspective division
        const V3 h = R_p_c * V3((double)pt.x, (double)pt.y, 1.0);
        pt.x = (float)h[0]; pt.y = (float)h[1];
    }
    float scaling_factor = 0.0f;
    rescalePoints(p1, p2, scaling_factor);
    norm_pixel_unit *= scaling_factor;
    std::vector<mskf_point2f> diff(n);
    for (size_t i = 0; i < n; ++i) diff[i] = mskf_point2f{p1[i].x - p2[i].x, p1[i].y - p2[i].y};
    std::vector<double> dist(n);
    double mean_pt_distance = 0.0;
    int raw_inlier_cntr = 0;
    for (size_t i = 0; i < n; ++i) {
        dist[i] = std::sqrt((double)(diff[i].x * diff[i].x + diff[i].y * diff[i].y));
        if (dist[i] > 50.0 * norm_pixel_unit) inlier_markers[i] = 0;
        else { mean_pt_distance += dist[i]; ++raw_inlier_cntr; }
    }
    mean_pt_distance /= raw_inlier_cntr;
    if (raw_inlier_cntr < 3) { inlier_markers.assign(n, 0); return; }
    if (mean_pt_distance < norm_pixel_unit) {   // degenerate motion, :992-1001
        for (size_t i = 0; i < n; ++i) {
            if (inlier_markers[i] == 0) continue;
            if (dist[i] > inlier_error * norm_pixel_unit) inlier_markers[i] = 0;
        }
        return;
    }
    std::vector<double> ct(3 * n);   // coeff_t rows: tx, ty, tz
    for (size_t i = 0; i < n; ++i) {
        ct[3 * i + 0] = (double)diff[i].y;
        ct[3 * i + 1] = (double)(-diff[i].x);
        ct[3 * i + 2] = (double)(p1[i].x * p2[i].y - p1[i].y * p2[i].x);
    }
    std::vector<int> raw_inlier_idx;
    for (size_t i = 0; i < n; ++i) if (inlier_markers[i] != 0) raw_inlier_idx.push_back((int)i);
    std::vector<int> best_inlier_set;
    const int m = (int)raw_inlier_idx.size();
    for (int iter_idx = 0; iter_idx < iter_num; ++iter_idx) {
        const int select_idx1 = uniformInteger(0, m - 1);
        const int select_idx_diff = uniformInteger(1, m - 1);
        const int select_idx2 = select_idx1 + select_idx_diff < m ? select_idx1 + select_idx_diff : select_idx1 + select_idx_diff - m;
        const int pair1 = raw_inlier_idx[select_idx1], pair2 = raw_inlier_idx[select_idx2];
        const double c[3][2] = {{ct[3 * pair1 + 0], ct[3 * pair2 + 0]}, {ct[3 * pair1 + 1], ct[3 * pair2 + 1]}, {ct[3 * pair1 + 2], ct[3 * pair2 + 2]}};
        const double l1[3] = {std::fabs(c[0][0]) + std::fabs(c[0][1]), std::fabs(c[1][0]) + std::fabs(c[1][1]), std::fabs(c[2][0]) + std::fabs(c[2][1])};
        int base = 0;
        for (int k = 1; k < 3; ++k) if (l1[k] < l1[base]) base = k;      // std::min_element: first minimum
        const int ia = base == 0 ? 1 : 0, ib = base == 2 ? 1 : 2;       // the two other columns, ascending
        double model[3], sol[2];
        {
            const double rhs[2] = {-c[base][0], -c[base][1]};
            solve2(c[ia], c[ib], rhs, sol);
            model[base] = 1.0; model[ia] = sol[0]; model[ib] = sol[1];
        }
        std::vector<int> inlier_set;
        for (size_t i = 0; i < n; ++i) {
            if (inlier_markers[i] == 0) continue;
            const double e = (ct[3 * i] * model[0] + ct[3 * i + 1] * model[1]) + ct[3 * i + 2] * model[2];
            if (std::fabs(e) < inlier_error * norm_pixel_unit) inlier_set.push_back((int)i);
        }
        if (inlier_set.size() < 0.2 * n) continue;
        // least-squares refit over the inliers: ((A^T A)^-1 A^T) (-c_base), evaluated left to right
        double saa = 0, sab = 0, sbb = 0;
        for (int idx : inlier_set) {
            const double a = ct[3 * idx + ia], b = ct[3 * idx + ib];
            saa += a * a; sab += a * b; sbb += b * b;
        }
        const double det = saa * sbb - sab * sab;
        const double i00 = sbb / det, i01 = -sab / det, i10 = -sab / det, i11 = saa / det;
        double s0 = 0, s1 = 0;
        for (int idx : inlier_set) {
            const double a = ct[3 * idx + ia], b = ct[3 * idx + ib], r = -ct[3 * idx + base];
            s0 += (i00 * a + i01 * b) * r;
            s1 += (i10 * a + i11 * b) * r;
        }
        double better[3];
        better[base] = 1.0; better[ia] = s0; better[ib] = s1;
        double this_error = 0.0;
        for (int idx : inlier_set)
            this_error += std::fabs((ct[3 * idx] * better[0] + ct[3 * idx + 1] * better[1]) + ct[3 * idx + 2] * better[2]);
        this_error /= inlier_set.size();
        (void)this_error;   // only compared through the set size in the reference (:1118-1121)
        if (inlier_set.size() > best_inlier_set.size()) best_inlier_set = inlier_set;
    }
    inlier_markers.assign(n, 0);
    for (int idx : best_inlier_set) inlier_markers[idx] = 1;
}

void ImageProcessor::undistortPoints(const std::vector<mskf_point2f> &in, const CamModel &cam,
                                     std::vector<mskf_point2f> &out, const M3 &R) {
    if (in.empty()) return;
    static const double Pdef[4] = {1, 1, 0, 0};  // image_processor.h:261
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) undistort_point(cam, R.m, Pdef, in[i].x, in[i].y, out[i].x, out[i].y);
}

void ImageProcessor::distortPoints(const std::vector<mskf_point2f> &in, const CamModel &cam, std::vector<mskf_point2f> &out) {
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) distort_point(cam, in[i].x, in[i].y, out[i].x, out[i].y);
}

// :534-620
void ImageProcessor::stereoMatch(const std::vector<mskf_point2f> &cam0_points, std::vector<mskf_point2f> &cam1_points,
                                 std::vector<uint8_t> &inlier_markers) {
    if (cam0_points.empty()) return;
    const M3 R_cam0_cam1 = R_cam1_imu.t() * R_cam0_imu;
    if (cam1_points.empty()) {
        std::vector<mskf_point2f> und;
        undistortPoints(cam0_points, cam0_, und, R_cam0_cam1);
        distortPoints(und, cam1_, cam1_points);
    }
    lk_track(curr_cam0_pyramid_, curr_cam1_pyramid_, cam0_points, cam1_points, inlier_markers);

    const int rows = cam1_curr_img.h, cols = cam1_curr_img.w;
    for (size_t i = 0; i < cam1_points.size(); ++i) {
        if (inlier_markers[i] == 0) continue;
        if (cam1_points[i].y < 0 || cam1_points[i].y > rows - 1 || cam1_points[i].x < 0 || cam1_points[i].x > cols - 1)
            inlier_markers[i] = 0;
    }
    const V3 t_cam0_cam1 = R_cam1_imu.t() * (t_cam0_imu - t_cam1_imu);
    const M3 E = skew(t_cam0_cam1) * R_cam0_cam1;

    std::vector<mskf_point2f> u0, u1;
    undistortPoints(cam0_points, cam0_, u0);
    undistortPoints(cam1_points, cam1_, u1);
    const double norm_pixel_unit = 4.0 / (cam0_.K[0] + cam0_.K[1] + cam1_.K[0] + cam1_.K[1]);
    for (size_t i = 0; i < u0.size(); ++i) {
        if (inlier_markers[i] == 0) continue;
        const double x0 = (double)u0[i].x, y0 = (double)u0[i].y, x1 = (double)u1[i].x, y1 = (double)u1[i].y;
        const double l0 = (E(0, 0) * x0 + E(0, 1) * y0) + E(0, 2);
        const double l1 = (E(1, 0) * x0 + E(1, 1) * y0) + E(1, 2);
        const double l2 = (E(2, 0) * x0 + E(2, 1) * y0) + E(2, 2);
        const double error = std::fabs((x1 * l0 + y1 * l1) + l2) / std::sqrt(l0 * l0 + l1 * l1);
        if (error > cfg_.stereo_threshold * norm_pixel_unit) inlier_markers[i] = 0;
    }
}

// :622-756
void ImageProcessor::addNewFeatures() {
    const Img &curr_img = cam0_curr_img;
    detector_.set_image_size(curr_img.w, curr_img.h);
    for (const auto &features : *curr_features_ptr)
        for (const auto &feature : features.second) {
            const int y = static_cast<int>(feature.cam0_point.y);
            const int x = static_cast<int>(feature.cam0_point.x);
            detector_.set_grid_position((float)x, (float)y);
        }
    std::vector<mskf_point2f> new_features;
    std::vector<double> new_features_responses;
    detector_.detect_features(curr_img, new_features, new_features_responses);

    std::vector<std::vector<std::pair<mskf_point2f, double>>> sieve((size_t)cfg_.grid_row * cfg_.grid_col);
    for (size_t i = 0; i < new_features.size(); ++i) {
        int row = static_cast<int>(new_features[i].y / grid_height);
        int col = static_cast<int>(new_features[i].x / grid_width);
        size_t code = (size_t)(row * cfg_.grid_col + col);
        if (code >= sieve.size()) continue;  // reference indexes out of bounds here (Q7, x >= grid_col*grid_width); defined: dropped
        sieve[code].push_back(std::make_pair(new_features[i], new_features_responses[i]));
    }
    new_features.clear();
    for (auto &item : sieve) {
        if ((int)item.size() > cfg_.grid_max_feature_num) {
            std::stable_sort(item.begin(), item.end(),
                             [](const std::pair<mskf_point2f, double> &a, const std::pair<mskf_point2f, double> &b) { return a.second > b.second; });
            item.erase(item.begin() + cfg_.grid_max_feature_num, item.end());
        }
        for (const auto &p : item) new_features.push_back(p.first);
    }
    int detected_new_features = (int)new_features.size();

    std::vector<mskf_point2f> cam0_points = new_features;
    std::vector<mskf_point2f> cam1_points;
    std::vector<uint8_t> inlier_markers;
    stereoMatch(cam0_points, cam1_points, inlier_markers);

    std::vector<mskf_point2f> cam0_inliers, cam1_inliers;
    std::vector<float> response_inliers;
    const bool q4 = (cfg_.compat_flags & MSKF_COMPAT_Q4_RESPONSE_INDEX) != 0;
    // responses re-gathered in sieve order for the non-compat path
    std::vector<double> sieved_responses;
    for (auto &item : sieve) for (const auto &p : item) sieved_responses.push_back(p.second);
    for (size_t i = 0; i < inlier_markers.size(); ++i) {
        if (inlier_markers[i] == 0) continue;
        cam0_inliers.push_back(cam0_points[i]);
        cam1_inliers.push_back(cam1_points[i]);
        response_inliers.push_back((float)(q4 ? new_features_responses[i] : sieved_responses[i]));  // Q4 (:698)
    }
    int matched_new_features = (int)cam0_inliers.size();
    (void)detected_new_features; (void)matched_new_features;  // "seems unsynced" print only (:703-706)

    GridFeatures grid_new_features;
    for (int code = 0; code < cfg_.grid_row * cfg_.grid_col; ++code) grid_new_features[code] = std::vector<FeatureMetaData>(0);
    for (size_t i = 0; i < cam0_inliers.size(); ++i) {
        int row = static_cast<int>(cam0_inliers[i].y / grid_height);
        int col = static_cast<int>(cam0_inliers[i].x / grid_width);
        int code = row * cfg_.grid_col + col;
        FeatureMetaData nf;
        nf.response = response_inliers[i];
        nf.cam0_point = cam0_inliers[i];
        nf.cam1_point = cam1_inliers[i];
        grid_new_features[code].push_back(nf);
    }
    for (auto &item : grid_new_features)
        std::stable_sort(item.second.begin(), item.second.end(), featureCompareByResponse);
    for (int code = 0; code < cfg_.grid_row * cfg_.grid_col; ++code) {
        std::vector<FeatureMetaData> &features_this_grid = (*curr_features_ptr)[code];
        std::vector<FeatureMetaData> &new_features_this_grid = grid_new_features[code];
        if ((int)features_this_grid.size() >= cfg_.grid_min_feature_num) continue;
        int vacancy_num = cfg_.grid_min_feature_num - (int)features_this_grid.size();
        for (int k = 0; k < vacancy_num && k < (int)new_features_this_grid.size(); ++k) {
            features_this_grid.push_back(new_features_this_grid[k]);
            features_this_grid.back().id = next_feature_id++;
            features_this_grid.back().lifetime = 1;
        }
    }
}

// :758-768
void ImageProcessor::pruneGridFeatures() {
    for (auto &item : *curr_features_ptr) {
        auto &grid_features = item.second;
        if ((int)grid_features.size() <= cfg_.grid_max_feature_num) continue;
        std::stable_sort(grid_features.begin(), grid_features.end(), featureCompareByLifetime);
        grid_features.erase(grid_features.begin() + cfg_.grid_max_feature_num, grid_features.end());
    }
}

// :850-889
void ImageProcessor::integrateImuData(M3 &cam0_R_p_c, M3 &cam1_R_p_c) {
    size_t begin = 0;
    while (begin < imu_msg_buffer.size()) {
        if (imu_msg_buffer[begin].time_stamp - cam0_prev_time < -0.01) ++begin;
        else break;
    }
    size_t end = begin;
    while (end < imu_msg_buffer.size()) {
        if (imu_msg_buffer[end].time_stamp - cam0_curr_time < 0.005) ++end;
        else break;
    }
    V3 mean_ang_vel;
    for (size_t i = begin; i < end; ++i)
        mean_ang_vel = mean_ang_vel + V3(imu_msg_buffer[i].angular_velocity[0], imu_msg_buffer[i].angular_velocity[1], imu_msg_buffer[i].angular_velocity[2]);
    if (end > begin) mean_ang_vel = mean_ang_vel * (double)(1.0f / (float)(end - begin));
    V3 cam0_mean = R_cam0_imu.t() * mean_ang_vel;
    V3 cam1_mean = R_cam1_imu.t() * mean_ang_vel;
    double dtime = cam0_curr_time - cam0_prev_time;  // 0 under Q2
    cam0_R_p_c = rodrigues(cam0_mean * dtime).t();
    cam1_R_p_c = rodrigues(cam1_mean * dtime).t();
    imu_msg_buffer.erase(imu_msg_buffer.begin(), imu_msg_buffer.begin() + end);
}

// :1137-1182
void ImageProcessor::publish() {
    feature_msg_ptr_->time_stamp = cam0_curr_time;
    std::vector<unsigned long long> curr_ids;
    std::vector<mskf_point2f> c0, c1;
    for (const auto &g : *curr_features_ptr)
        for (const auto &f : g.second) {
            curr_ids.push_back(f.id);
            c0.push_back(f.cam0_point);
            c1.push_back(f.cam1_point);
        }
    std::vector<mskf_point2f> u0, u1;
    undistortPoints(c0, cam0_, u0);
    undistortPoints(c1, cam1_, u1);
    if (!(cfg_.compat_flags & MSKF_COMPAT_Q1_MSG_ACCUMULATE)) feature_msg_ptr_->features.clear();
    for (size_t i = 0; i < curr_ids.size(); ++i) {
        feature_msg_ptr_->features.push_back(mskf_feature_meas{0, 0, 0, 0, 0, 0});  // Q1
        feature_msg_ptr_->features[i].id = (uint32_t)curr_ids[i];
        feature_msg_ptr_->features[i].u0 = u0[i].x;
        feature_msg_ptr_->features[i].v0 = u0[i].y;
        feature_msg_ptr_->features[i].u1 = u1[i].x;
        feature_msg_ptr_->features[i].v1 = u1[i].y;
    }
}

}  // namespace orc
