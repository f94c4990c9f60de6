// image_processor.cpp — host mirror of cg::ImageProcessor; see image_processor.h.
// Reference: msckf_core/src/image_processor.cpp (line numbers cited per function).
#include "image_processor.h"
#include "host_prof.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>

namespace cg {

// config_io.h:13-81 conversions + image_processor.cpp:54-72 / msckf_vio.cpp:115-125 keys
mskf_calib calib_from_yaml(const YAML::Node &n) {
    mskf_calib c;
    std::memset(&c, 0, sizeof(c));
    auto model = [](const std::string &s) { return s == "equidistant" ? MSKF_MODEL_EQUIDISTANT : MSKF_MODEL_RADTAN; };
    auto copy = [](const std::vector<double> &v, double *dst, size_t cnt, const char *what) {
        if (v.size() != cnt) throw YAML::Exception(std::string("yaml: wrong length for ") + what);
        for (size_t i = 0; i < cnt; ++i) dst[i] = v[i];
    };
    c.cam0_model = model(n["cam0"]["distortion_model"].as<std::string>());
    c.cam1_model = model(n["cam1"]["distortion_model"].as<std::string>());
    copy(n["cam0"]["intrinsics"].as<std::vector<double>>(), c.cam0_intrinsics, 4, "cam0.intrinsics");
    copy(n["cam1"]["intrinsics"].as<std::vector<double>>(), c.cam1_intrinsics, 4, "cam1.intrinsics");
    copy(n["cam0"]["distortion_coeffs"].as<std::vector<double>>(), c.cam0_distortion, 4, "cam0.distortion_coeffs");
    copy(n["cam1"]["distortion_coeffs"].as<std::vector<double>>(), c.cam1_distortion, 4, "cam1.distortion_coeffs");
    copy(n["cam0"]["T_cam_imu"].as<std::vector<double>>(), c.T_cam0_imu, 16, "cam0.T_cam_imu");
    copy(n["cam1"]["T_cn_cnm1"].as<std::vector<double>>(), c.T_cam1_cam0, 16, "cam1.T_cn_cnm1");
    copy(n["T_imu_body"].as<std::vector<double>>(), c.T_imu_body, 16, "T_imu_body");
    // resolution is not read by the reference (image size comes from the first image); keep it as a hint
    if (n["cam0"]["resolution"].IsDefined()) {
        std::vector<double> r = n["cam0"]["resolution"].as<std::vector<double>>();
        if (r.size() == 2) { c.width = (int)r[0]; c.height = (int)r[1]; }
    }
    return c;
}

mskf_fe_cfg fe_cfg_from_yaml(const YAML::Node &y) {
    mskf_fe_cfg c;
    std::memset(&c, 0, sizeof(c));
    c.grid_row = y["grid_row"].as<int>();
    c.grid_col = y["grid_col"].as<int>();
    c.grid_min_feature_num = y["grid_min_feature_num"].as<int>();
    c.grid_max_feature_num = y["grid_max_feature_num"].as<int>();
    c.pyramid_levels = y["pyramid_levels"].as<int>();
    c.patch_size = y["patch_size"].as<int>();
    c.fast_threshold = y["fast_threshold"].as<int>();
    c.ransac_threshold = y["ransac_threshold"].as<int>();   // parsed as<int>() in the reference (:83-84, Q6)
    c.stereo_threshold = y["stereo_threshold"].as<int>();
    c.max_iteration = y["max_iteration"].as<int>();
    c.track_precision = y["track_precision"].as<double>();
    c.det_rows = 30; c.det_cols = 47;                       // CornerDetector(30, 47, thr), :132
    c.compat_flags = MSKF_COMPAT_REFERENCE;
    return c;
}

ImageProcessor::ImageProcessor(YAML::Node cfg_cam_imu)
    : feature_msg_ptr_(new CameraMeasurement), cfg_cam_imu_(cfg_cam_imu), have_yaml_(true) {
    std::memset(&calib_, 0, sizeof(calib_));
    std::memset(&cfg_, 0, sizeof(cfg_));
    std::memset(&processor_config, 0, sizeof(processor_config));
}

ImageProcessor::ImageProcessor(const mskf_calib &calib, const mskf_fe_cfg &cfg)
    : feature_msg_ptr_(new CameraMeasurement), calib_(calib), cfg_(cfg) {
    std::memset(&processor_config, 0, sizeof(processor_config));
}

ImageProcessor::~ImageProcessor() {
    if (debug_.is_open()) debug_.close();
    if (own_stream_ && stream_) mskf_stream_destroy(stream_);
    if (own_ctx_) mskf_ctx_destroy(own_ctx_);
}

void ImageProcessor::fail(const char *what, int rc) {
    error_ = std::string(what) + ": " + mskf_last_error();
    std::fprintf(stderr, "ImageProcessor: %s failed (%d): %s\n", what, rc, mskf_last_error());
}

// image_processor.cpp:52-124
bool ImageProcessor::loadParameters() {
    if (have_yaml_) {
        calib_ = calib_from_yaml(cfg_cam_imu_);
        YAML::Node cfg_imgproc = YAML::LoadFile("../config/app_imgproc.yaml");   // Q16
        cfg_ = fe_cfg_from_yaml(cfg_imgproc);
    }
    processor_config.grid_row = cfg_.grid_row;
    processor_config.grid_col = cfg_.grid_col;
    processor_config.grid_min_feature_num = cfg_.grid_min_feature_num;
    processor_config.grid_max_feature_num = cfg_.grid_max_feature_num;
    processor_config.pyramid_levels = cfg_.pyramid_levels;
    processor_config.patch_size = cfg_.patch_size;
    processor_config.fast_threshold = cfg_.fast_threshold;
    processor_config.max_iteration = cfg_.max_iteration;
    processor_config.track_precision = cfg_.track_precision;
    processor_config.ransac_threshold = cfg_.ransac_threshold;
    processor_config.stereo_threshold = cfg_.stereo_threshold;
    // :63-72
    hm::Rigid m4_cam0_imu = hm::Rigid::from_rowmajor16(calib_.T_cam0_imu);
    R_cam0_imu = m4_cam0_imu.R.transpose();
    t_cam0_imu = -(R_cam0_imu * m4_cam0_imu.t);
    hm::Rigid m4_cam1_cam0 = hm::Rigid::from_rowmajor16(calib_.T_cam1_cam0);
    hm::Rigid T_cam1_imu = m4_cam1_cam0 * m4_cam0_imu;
    R_cam1_imu = T_cam1_imu.R.transpose();
    t_cam1_imu = -(R_cam1_imu * T_cam1_imu.t);
    return true;
}

// :126-137
bool ImageProcessor::initialize() {
    if (!loadParameters()) return false;
    occupancy_.assign((size_t)cfg_.det_rows * cfg_.det_cols, 0);
    if (have_yaml_) debug_.open("debug_imageprocessor.txt");
    return true;
}

// :205-211
void ImageProcessor::imuCallback(const cg::ImuConstPtr &msg) {
    if (is_first_img) return;
    imu_msg_buffer.push_back(*msg);
}

void ImageProcessor::phaseBegin(double time_stamp, int width, int height) {
    if (width <= 0 || height <= 0) { width = calib_.width; height = calib_.height; }
    cam0_curr_time = time_stamp;
    if ((cfg_.compat_flags & MSKF_COMPAT_Q2_PREV_ALIAS) && !is_first_img) cam0_prev_time = time_stamp;  // Q2 (:192)
    img_w = width; img_h = height;
    if (grid_height == 0) {  // Q7: function-local statics, first image wins (:250-251)
        grid_height = height / cfg_.grid_row;
        grid_width = width / cfg_.grid_col;
        det_cell_h = (height + cfg_.det_rows - 1) / cfg_.det_rows;
        det_cell_w = (width + cfg_.det_cols - 1) / cfg_.det_cols;
        // largest grid code a point inside the image can get (Q7: rows / columns past the nominal grid exist when the
        // image size is not a multiple of the grid)
        n_codes_ = std::max(((height - 1) / grid_height) * cfg_.grid_col + (width - 1) / grid_width + 1, cfg_.grid_row * cfg_.grid_col);
    }
}

// :139-203
void ImageProcessor::stereoCallback(const cg::Image &cam0_img, const cg::Image &cam1_img, bool is_draw) {
    const int w = cam0_img.image.cols(), h = cam0_img.image.rows();
    if (!stream_) {
        // stand-alone use (no System): own context + stream sized from the first image
        calib_.width = w; calib_.height = h;
        mskf_ekf_cfg e;
        std::memset(&e, 0, sizeof(e));
        e.max_cam_state_size = 20; e.max_stack_rows = 1500; e.noise_feature = 0.035;
        int rc = mskf_ctx_create(0, &own_ctx_);
        if (rc == MSKF_OK) rc = mskf_stream_create(own_ctx_, &calib_, &cfg_, &e, &stream_);
        if (rc != MSKF_OK) { fail("mskf_stream_create", rc); return; }
        own_stream_ = true;
        mskf_fe_set_detect_floor(stream_, cfg_.fast_threshold * 256);
    }
    if (canDeviceFrame()) {
        // every frame after the first: one call, the bookkeeping between the tracks runs on the device
        mskf_fe_frame_args fa;
        if (!frameBegin(cam0_img.time_stamp, fa)) return;
        mskf_stream *ss[1] = {stream_};
        const uint8_t *a[1] = {cam0_img.image.data()}, *b[1] = {cam1_img.image.data()};
        int frc = mskf_fe_frame_batch_begin(mskf_stream_ctx(stream_), 1, ss, a, b, 0, &fa);
        if (frc == MSKF_OK) frc = mskf_fe_frame_batch_end(mskf_stream_ctx(stream_));
        if (frc != MSKF_OK) { fail("mskf_fe_frame_batch", frc); return; }
        frameEnd(fa, is_draw);
        return;
    }
    phaseBegin(cam0_img.time_stamp, w, h);
    int rc = mskf_fe_push_stereo(stream_, cam0_img.image.data(), cam1_img.image.data(), w, h, w, cam0_img.time_stamp);
    if (rc != MSKF_OK) { fail("mskf_fe_push_stereo", rc); return; }
    mskf_fe_track_args a1, a2;
    phasePrepare1(a1);
    rc = mskf_fe_track(stream_, &a1);
    if (rc != MSKF_OK) { fail("mskf_fe_track", rc); return; }
    phaseAfter1(a2);
    if (a2.n > 0) {
        rc = mskf_fe_track(stream_, &a2);
        if (rc != MSKF_OK) { fail("mskf_fe_track", rc); return; }
    }
    phaseAfter2(is_draw);
}

static void fill_args(mskf_fe_track_args &a, int n, int do_temporal, std::vector<mskf_point2f> &in, std::vector<mskf_point2f> &o0,
                      std::vector<mskf_point2f> &o1, std::vector<mskf_point2f> &u0, std::vector<mskf_point2f> &u1,
                      std::vector<uint8_t> &st) {
    std::memset(&a, 0, sizeof(a));
    o0.resize(n); o1.resize(n); u0.resize(n); u1.resize(n); st.assign(n, 0);
    a.n = n; a.do_temporal = do_temporal;
    a.in_pts = in.data(); a.out0 = o0.data(); a.out1 = o1.data(); a.und0 = u0.data(); a.und1 = u1.data(); a.status = st.data();
    a.Hpred[0] = a.Hpred[4] = a.Hpred[8] = 1.0;
}

// CornerDetector::detect_features on the device's per-cell maxima: threshold + occupancy, cell order
void ImageProcessor::detectFeatures(std::vector<Point2f> &pts, std::vector<double> &responses) {
    const int cells = cfg_.det_rows * cfg_.det_cols;
    cell_max_.resize(cells);
    int n = 0;
    // only the cells whose maximum beats the detector threshold come back (in cell order)
    int rc = mskf_fe_get_cell_candidates(stream_, cfg_.fast_threshold * 256, cell_max_.data(), cells, &n);
    pts.clear(); responses.clear();
    if (rc != MSKF_OK) { fail("mskf_fe_get_cell_candidates", rc); return; }
    for (int i = 0; i < n; ++i) {
        if (!occupancy_[cell_max_[i].cell]) {
            pts.push_back(Point2f(cell_max_[i].x, cell_max_[i].y));
            responses.push_back((double)cell_max_[i].score / 256.0);
        }
    }
    std::fill(occupancy_.begin(), occupancy_.end(), 0);
}

void ImageProcessor::setGridPosition(float x, float y) {
    int r = (int)(y / (float)det_cell_h), c = (int)(x / (float)det_cell_w);
    r = r < 0 ? 0 : (r >= cfg_.det_rows ? cfg_.det_rows - 1 : r);
    c = c < 0 ? 0 : (c >= cfg_.det_cols ? cfg_.det_cols - 1 : c);
    occupancy_[(size_t)r * cfg_.det_cols + c] = 1;
}

// :321-340  H = K R K^-1
void ImageProcessor::computeHpred(const hm::Mat3 &R_p_c, double H[9]) const {
    const double *intr = calib_.cam0_intrinsics;
    hm::Mat3 K; K(0, 0) = intr[0]; K(0, 2) = intr[2]; K(1, 1) = intr[1]; K(1, 2) = intr[3]; K(2, 2) = 1.0;
    hm::Mat3 Ki; Ki(0, 0) = 1.0 / intr[0]; Ki(0, 2) = -intr[2] / intr[0]; Ki(1, 1) = 1.0 / intr[1]; Ki(1, 2) = -intr[3] / intr[1]; Ki(2, 2) = 1.0;
    hm::Mat3 Hm = K * R_p_c * Ki;
    std::memcpy(H, Hm.m, sizeof(double) * 9);
}

static hm::Mat3 rodrigues(const hm::Vec3 &r) {   // cv::Rodrigues, vector -> matrix
    const double th = hm::norm(r);
    if (th < 2.220446049250313e-16) return hm::Mat3::identity();
    hm::Vec3 k = r / th;
    const double c = std::cos(th), s = std::sin(th), c1 = 1 - c;
    hm::Mat3 R = c * hm::Mat3::identity() + s * hm::skew(k);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) += c1 * k[i] * k[j];
    return R;
}

// :850-889
void ImageProcessor::integrateImuData(hm::Mat3 &cam0_R_p_c, hm::Mat3 &cam1_R_p_c) {
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
    hm::Vec3 mean_ang_vel;
    for (size_t i = begin; i < end; ++i) mean_ang_vel = mean_ang_vel + imu_msg_buffer[i].angular_velocity;
    if (end > begin) mean_ang_vel = mean_ang_vel * (double)(1.0f / (float)(end - begin));
    hm::Vec3 cam0_mean = R_cam0_imu.transpose() * mean_ang_vel;
    hm::Vec3 cam1_mean = R_cam1_imu.transpose() * mean_ang_vel;
    const double dtime = cam0_curr_time - cam0_prev_time;   // 0 under Q2
    cam0_R_p_c = rodrigues(cam0_mean * dtime).transpose();
    cam1_R_p_c = rodrigues(cam1_mean * dtime).transpose();
    imu_msg_buffer.erase(imu_msg_buffer.begin(), imu_msg_buffer.begin() + end);
}

void ImageProcessor::phasePrepare1(mskf_fe_track_args &args) {
    if (is_first_img) {
        // initializeFirstFrame head (:247-268): detect, stereo-match all detections
        std::vector<Point2f> det;
        detectFeatures(det, cand_responses_det_);
        in_pts_.resize(det.size());
        cand_index_.resize(det.size());
        for (size_t i = 0; i < det.size(); ++i) { in_pts_[i] = mskf_point2f{det[i].x, det[i].y}; cand_index_[i] = (int)i; }
        fill_args(args, (int)in_pts_.size(), 0, in_pts_, out0_, out1_, und0_, und1_, status_);
        stage_ = 1;
        return;
    }
    // trackFeatures head (:352-410): the previous grid in iteration order is exactly what prev_ holds
    hostprof::Scope hp(hostprof::FE_PREPARE);
    hm::Mat3 cam0_R_p_c, cam1_R_p_c;
    integrateImuData(cam0_R_p_c, cam1_R_p_c);
    cam0_R_p_c_ = cam0_R_p_c; cam1_R_p_c_ = cam1_R_p_c;
    fill_args(args, (int)prev_.size(), 1, prev_.cam0, out0_, out1_, und0_, und1_, status_);
    computeHpred(cam0_R_p_c, args.Hpred);
    stage_ = 2;
}

static bool cmpResponse(const float &a, const float &b) { return a > b; }

// std::stable_sort allocates a merge buffer on every call; the per-cell lists it is used on hold a handful of
// entries, for which a stable insertion sort gives the same order without touching the allocator
template <class It, class Cmp>
static void small_stable_sort(It b, It e, Cmp cmp) {
    if (e - b > 32) { std::stable_sort(b, e, cmp); return; }
    for (It i = b; i != e; ++i)
        for (It j = i; j != b && cmp(*j, *(j - 1)); --j) std::iter_swap(j, j - 1);
}

// Matched candidates (status bit 1) grouped by grid code, the best `vacancy` of every cell by response (stable: Q19) get
// the next ids in code order (:700-750; first frame :270-316 with every cell empty).  Candidates arrive in ascending code
// order already (the sieve emits them cell by cell, detections of the first frame are binned here).
void ImageProcessor::selectNewFeatures(bool first_frame) {
    const bool q4 = (cfg_.compat_flags & MSKF_COMPAT_Q4_RESPONSE_INDEX) != 0;
    const int n_cells = cfg_.grid_row * cfg_.grid_col;
    new_.clear();
    // (code, response, index) of the matched candidates
    struct Cand { int code; float response; int idx; };
    std::vector<Cand> cands;
    cands.reserve(status_.size());
    for (size_t i = 0; i < status_.size(); ++i) {
        if (!(status_[i] & 2)) continue;
        const double resp = first_frame ? cand_responses_det_[i] : (q4 ? cand_responses_det_[cand_index_[i]] : cand_responses_sieved_[i]);   // Q4 (:698)
        cands.push_back(Cand{gridCode(out0_[i]), (float)resp, (int)i});
    }
    // ascending code, then descending response; stable, so equal responses keep their candidate order
    std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.code != b.code ? a.code < b.code : a.response > b.response; });
    for (size_t k = 0; k < cands.size();) {
        const int code = cands[k].code;
        size_t e = k;
        while (e < cands.size() && cands[e].code == code) ++e;
        if (code >= 0 && code < n_cells) {        // only the nominal cells are filled (:736)
            const int have = first_frame ? 0 : cell_count_[code];
            const int vacancy = cfg_.grid_min_feature_num - have;
            for (int q = 0; q < vacancy && k + q < e; ++q) {
                const int i = cands[k + q].idx;
                new_.push(next_feature_id++, 1, code, cands[k + q].response, out0_[i], out1_[i], und0_[i], und1_[i]);
            }
        }
        k = e;
    }
}

void ImageProcessor::initializeFirstFrameTail() {
    tracked_.clear();
    cell_count_.assign(n_codes_, 0);
    selectNewFeatures(true);
}

// :416-513
void ImageProcessor::trackFeaturesTail() {
    tracked_.clear();
    cell_count_.assign(n_codes_, 0);
    before_tracking = (int)prev_.size();
    if (prev_.size() == 0) return;   // :383
    after_tracking = 0; after_matching = 0; after_ransac = 0;
    // Q5: the reference has both twoPointRansac calls commented out (:482-500); with the switch cleared they run on the
    // matched cam0 and cam1 temporal pairs and a feature must be an inlier of both
    const bool ransac = !(cfg_.compat_flags & MSKF_COMPAT_Q5_NO_RANSAC);
    std::vector<int> keep;
    if (ransac) {
        std::vector<Point2f> p0, c0v, p1, c1v;
        std::vector<size_t> idx;
        for (size_t i = 0; i < status_.size(); ++i) {
            if ((status_[i] & 3) != 3) continue;
            idx.push_back(i);
            p0.push_back(Point2f(prev_.und0[i].x, prev_.und0[i].y)); p1.push_back(Point2f(prev_.und1[i].x, prev_.und1[i].y));
            c0v.push_back(Point2f(und0_[i].x, und0_[i].y)); c1v.push_back(Point2f(und1_[i].x, und1_[i].y));
        }
        std::vector<int> in0, in1;
        twoPointRansac(p0, c0v, cam0_R_p_c_, calib_.cam0_intrinsics, processor_config.ransac_threshold, 0.99, in0);
        twoPointRansac(p1, c1v, cam1_R_p_c_, calib_.cam1_intrinsics, processor_config.ransac_threshold, 0.99, in1);
        keep.assign(status_.size(), 0);
        for (size_t k = 0; k < idx.size(); ++k) keep[idx[k]] = in0[k] != 0 && in1[k] != 0;
    }
    for (size_t i = 0; i < status_.size(); ++i) {
        if (!(status_[i] & 1)) continue;
        ++after_tracking;
        if (!(status_[i] & 2)) continue;
        ++after_matching;
        if (ransac && !keep[i]) continue;
        int code = gridCode(out0_[i]);   // Q7: col may equal grid_col
        if (code < 0) code = 0;
        if (code >= n_codes_) { n_codes_ = code + 1; cell_count_.resize(n_codes_, 0); }
        tracked_.push(prev_.id[i], prev_.lifetime[i] + 1, code, 0.f, out0_[i], out1_[i], und0_[i], und1_[i]);
        ++cell_count_[code];
        ++after_ransac;
    }
}

// Counter-based stand-in for cg::uniform_integer(lo, hi): splitmix64 of the draw counter, reduced to [lo, hi]
static int uniformInteger(unsigned long long &ransac_draws, int lo, int hi) {
    uint64_t z = 0x5EED5EED5EED5EEDULL + 0x9E3779B97F4A7C15ULL * (++ransac_draws);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return lo + (int)(z % (uint64_t)(hi - lo + 1));
}

// :888-908 (single precision, as cg::Point2f arithmetic)
static void rescalePoints(std::vector<Point2f> &pts1, std::vector<Point2f> &pts2, float &scaling_factor) {
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

void ImageProcessor::twoPointRansac(const std::vector<Point2f> &pts1_undistorted, const std::vector<Point2f> &pts2_undistorted,
                                    const hm::Mat3 &R_p_c, const double intrinsics[4], double inlier_error,
                                    double success_probability, std::vector<int> &inlier_markers) {
    two_point_ransac(pts1_undistorted, pts2_undistorted, R_p_c, intrinsics, inlier_error, success_probability, ransac_draws, inlier_markers);
}

// :911-1135
void two_point_ransac(const std::vector<Point2f> &pts1_undistorted, const std::vector<Point2f> &pts2_undistorted,
                      const hm::Mat3 &R_p_c, const double intrinsics[4], double inlier_error,
                      double success_probability, unsigned long long &ransac_draws, std::vector<int> &inlier_markers) {
    const size_t n = pts1_undistorted.size();
    double norm_pixel_unit = 2.0 / (intrinsics[0] + intrinsics[1]);
    const int iter_num = static_cast<int>(std::ceil(std::log(1 - success_probability) / std::log(1 - 0.7 * 0.7)));
    inlier_markers.assign(n, 1);     // :925-926
    if (n == 0) return;
    std::vector<Point2f> prev = pts1_undistorted, curr = pts2_undistorted;
    // compensate the previous points with the relative rotation (:936-944)
    for (Point2f &pt : prev) {
        const hm::Vec3 rotated = R_p_c * hm::Vec3((double)pt.x, (double)pt.y, 1.0);
        pt = Point2f((float)rotated[0], (float)rotated[1]);
    }
    float scale = 0.0f;
    rescalePoints(prev, curr, scale);
    norm_pixel_unit *= scale;
    // per pair: difference, its length, and the epipolar coefficients of (tx, ty, tz) (:949-1012)
    struct Pair { double len, tx, ty, tz; };
    std::vector<Pair> pairs(n);
    double length_sum = 0.0;
    int candidates = 0;
    for (size_t i = 0; i < n; ++i) {
        const float dx = prev[i].x - curr[i].x, dy = prev[i].y - curr[i].y;
        Pair &q = pairs[i];
        q.len = std::sqrt((double)(dx * dx + dy * dy));
        q.tx = (double)dy;
        q.ty = (double)(-dx);
        q.tz = (double)(prev[i].x * curr[i].y - prev[i].y * curr[i].x);
        if (q.len > 50.0 * norm_pixel_unit) inlier_markers[i] = 0;      // :961-967
        else { length_sum += q.len; ++candidates; }
    }
    const double mean_length = length_sum / candidates;
    if (candidates < 3) { inlier_markers.assign(n, 0); return; }       // :974-977
    const double tol = inlier_error * norm_pixel_unit;
    if (mean_length < norm_pixel_unit) {                                // degenerate (pure rotation) case, :985-1001
        for (size_t i = 0; i < n; ++i)
            if (inlier_markers[i] && pairs[i].len > tol) inlier_markers[i] = 0;
        return;
    }
    std::vector<int> pool;
    for (size_t i = 0; i < n; ++i) if (inlier_markers[i]) pool.push_back((int)i);
    const int m = (int)pool.size();
    auto coeff = [&](int i, int k) { return k == 0 ? pairs[i].tx : (k == 1 ? pairs[i].ty : pairs[i].tz); };
    std::vector<int> best, support;
    for (int it = 0; it < iter_num; ++it) {
        // two distinct pairs (:1025-1033)
        const int first = uniformInteger(ransac_draws, 0, m - 1);
        const int step = uniformInteger(ransac_draws, 1, m - 1);
        const int second = first + step < m ? first + step : first + step - m;
        const int i1 = pool[first], i2 = pool[second];
        // the coefficient column with the smallest L1 norm is fixed to 1, the 2x2 system gives the other two (:1036-1065)
        double l1[3];
        for (int k = 0; k < 3; ++k) l1[k] = std::fabs(coeff(i1, k)) + std::fabs(coeff(i2, k));
        int fixed = 0;
        for (int k = 1; k < 3; ++k) if (l1[k] < l1[fixed]) fixed = k;
        const int ka = fixed == 0 ? 1 : 0, kb = fixed == 2 ? 1 : 2;
        double t[3];
        {
            const double a0 = coeff(i1, ka), a1 = coeff(i2, ka), b0 = coeff(i1, kb), b1 = coeff(i2, kb);
            const double r0 = -coeff(i1, fixed), r1 = -coeff(i2, fixed);
            const double det = a0 * b1 - b0 * a1;
            const double v00 = b1 / det, v01 = -b0 / det, v10 = -a1 / det, v11 = a0 / det;
            t[fixed] = 1.0;
            t[ka] = v00 * r0 + v01 * r1;
            t[kb] = v10 * r0 + v11 * r1;
        }
        support.clear();
        for (size_t i = 0; i < n; ++i) {
            if (!inlier_markers[i]) continue;
            const double err = (pairs[i].tx * t[0] + pairs[i].ty * t[1]) + pairs[i].tz * t[2];
            if (std::fabs(err) < tol) support.push_back((int)i);
        }
        if (support.size() < 0.2 * n) continue;                         // :1078-1079
        // refit on the support set: ((A^T A)^-1 A^T)(-c), left to right (:1082-1112)
        double saa = 0, sab = 0, sbb = 0;
        for (int i : support) { const double a = coeff(i, ka), b = coeff(i, kb); saa += a * a; sab += a * b; sbb += b * b; }
        const double det = saa * sbb - sab * sab;
        const double v00 = sbb / det, v01 = -sab / det, v10 = -sab / det, v11 = saa / det;
        double ua = 0, ub = 0;
        for (int i : support) {
            const double a = coeff(i, ka), b = coeff(i, kb), r = -coeff(i, fixed);
            ua += (v00 * a + v01 * b) * r;
            ub += (v10 * a + v11 * b) * r;
        }
        (void)ua; (void)ub;       // the refitted model only feeds best_error, which nothing reads (:1114-1121)
        if (support.size() > best.size()) best = support;
    }
    inlier_markers.assign(n, 0);
    for (int i : best) inlier_markers[i] = 1;
}

// :622-688
void ImageProcessor::addNewFeaturesHead() {
    {
        hostprof::Scope hp(hostprof::FE_OCCUPANCY);
        for (size_t k = 0; k < tracked_.size(); ++k) {
            const int y = static_cast<int>(tracked_.cam0[k].y);
            const int x = static_cast<int>(tracked_.cam0[k].x);
            setGridPosition((float)x, (float)y);
        }
    }
    std::vector<Point2f> new_features;
    { hostprof::Scope hp(hostprof::FE_DETECT); detectFeatures(new_features, cand_responses_det_); }
    hostprof::Scope hp_sieve(hostprof::FE_SIEVE);
    // The reference sieves every detection into its grid cell, keeps the grid_max best of a cell (:664-675), stereo-matches
    // all of them (:677-688) and then uses only those of cells with a vacancy (:736-750).  A candidate's cell is its sieve
    // cell (the matched cam0 point IS the candidate) and the vacancies are known here, after tracking: candidates of
    // full cells cannot influence any output, so only their NUMBER is kept.  It still counts: Q4 indexes the
    // detection-order responses with the candidate's position in the reference's full candidate list, so every
    // candidate that is sent to the device carries that position.
    const int n_cells = cfg_.grid_row * cfg_.grid_col;
    sieve_.resize((size_t)n_cells);
    sieve_count_.assign((size_t)n_cells, 0);
    for (auto &cell : sieve_) cell.clear();
    for (size_t i = 0; i < new_features.size(); ++i) {
        const mskf_point2f p{new_features[i].x, new_features[i].y};
        const int code = gridCode(p);
        if (code < 0 || code >= n_cells) continue;   // the reference indexes out of bounds here (Q7); defined: dropped
        ++sieve_count_[code];
        if (cell_count_[code] < cfg_.grid_min_feature_num) sieve_[code].push_back(std::make_pair(p, cand_responses_det_[i]));
    }
    in_pts_.clear(); cand_responses_sieved_.clear(); cand_index_.clear();
    int flat = 0;
    for (int code = 0; code < n_cells; ++code) {
        const int kept = std::min(sieve_count_[code], cfg_.grid_max_feature_num);
        auto &item = sieve_[code];
        if (!item.empty()) {
            if ((int)item.size() > cfg_.grid_max_feature_num)
                small_stable_sort(item.begin(), item.end(),
                                  [](const std::pair<mskf_point2f, double> &a, const std::pair<mskf_point2f, double> &b) { return a.second > b.second; });
            for (int k = 0; k < kept; ++k) {
                in_pts_.push_back(item[k].first);
                cand_responses_sieved_.push_back(item[k].second);
                cand_index_.push_back(flat + k);
            }
        }
        flat += kept;
    }
}

// :690-750
void ImageProcessor::addNewFeaturesTail() { selectNewFeatures(false); }

// This frame's grid in the reference's iteration order: ascending code; inside a cell the tracked features in track
// order, then the new ones in rank order; cells over grid_max keep their grid_max longest-lived features
// (pruneGridFeatures, :758-768; stable).  tracked_ is in track order, new_ in code order.
void ImageProcessor::assembleGrid() {
    curr_.clear();
    cell_start_.assign((size_t)n_codes_ + 1, 0);
    for (size_t k = 0; k < tracked_.size(); ++k) ++cell_start_[tracked_.code[k] + 1];
    for (int c = 0; c < n_codes_; ++c) cell_start_[c + 1] += cell_start_[c];
    order_.resize(tracked_.size());
    {
        std::vector<int> &fill = sieve_count_;      // scratch
        fill.assign((size_t)n_codes_, 0);
        for (size_t k = 0; k < tracked_.size(); ++k) { const int c = tracked_.code[k]; order_[cell_start_[c] + fill[c]++] = (int)k; }
    }
    size_t nn = 0;
    struct Slot { int lifetime; int from_new; int idx; };
    std::vector<Slot> cell;
    for (int c = 0; c < n_codes_; ++c) {
        cell.clear();
        for (int q = cell_start_[c]; q < cell_start_[c + 1]; ++q) cell.push_back(Slot{tracked_.lifetime[order_[q]], 0, order_[q]});
        while (nn < new_.size() && new_.code[nn] == c) { cell.push_back(Slot{new_.lifetime[nn], 1, (int)nn}); ++nn; }
        if ((int)cell.size() > cfg_.grid_max_feature_num) {
            small_stable_sort(cell.begin(), cell.end(), [](const Slot &a, const Slot &b) { return a.lifetime > b.lifetime; });
            cell.resize((size_t)cfg_.grid_max_feature_num);
        }
        for (const Slot &sl : cell) curr_.push_from(sl.from_new ? new_ : tracked_, (size_t)sl.idx);
    }
}

void ImageProcessor::phaseAfter1(mskf_fe_track_args &args2) {
    std::memset(&args2, 0, sizeof(args2));
    if (stage_ == 1) {
        initializeFirstFrameTail();
        is_first_img = false;
        stage_ = 4;
        return;
    }
    { hostprof::Scope hp(hostprof::FE_TRACK_TAIL); trackFeaturesTail(); }
    addNewFeaturesHead();
    fill_args(args2, (int)in_pts_.size(), 0, in_pts_, out0_, out1_, und0_, und1_, status_);
    stage_ = 3;
}

// process-wide switch: -1 = not decided (MSKF_FE_BOOKS=host in the environment decides), 0 = device, 1 = host
static std::atomic<int> g_fe_books_host{-1};
void ImageProcessor::setFeBooksOnHost(int on) { g_fe_books_host.store(on < 0 ? -1 : (on ? 1 : 0)); }
bool ImageProcessor::feBooksOnHost() {
    int v = g_fe_books_host.load(std::memory_order_relaxed);
    if (v < 0) { const char *e = std::getenv("MSKF_FE_BOOKS"); v = (e && e[0] == 'h') ? 1 : 0; g_fe_books_host.store(v); }
    return v != 0;
}

bool ImageProcessor::canDeviceFrame() const {
    if (feBooksOnHost() || !stream_ || is_first_img) return false;
    return mskf_fe_grid_capacity(stream_) > 0;       // (the RANSAC of :482-500, when switched on, runs inside the device frame too)
}

bool ImageProcessor::frameBegin(double time_stamp, mskf_fe_frame_args &a) {
    phaseBegin(time_stamp, 0, 0);
    hostprof::Scope hp(hostprof::FE_PREPARE);
    std::memset(&a, 0, sizeof(a));
    if (!device_grid_valid_) {
        // the last frame ran on the host (the first frame of the stream): hand the device its grid, id counter and the
        // tracking counters that survive frames without features
        const int32_t counters[3] = {after_tracking, after_matching, after_ransac};
        const int rc = mskf_fe_set_grid(stream_, (int)prev_.size(), (const uint64_t *)prev_.id.data(), prev_.lifetime.data(), prev_.cam0.data(), prev_.cam1.data(),
                                        prev_.und0.data(), prev_.und1.data(), (uint64_t)next_feature_id, counters, (uint64_t)ransac_draws);
        if (rc != MSKF_OK) { fail("mskf_fe_set_grid", rc); return false; }
        device_grid_valid_ = true;
    }
    hm::Mat3 cam0_R_p_c, cam1_R_p_c;
    integrateImuData(cam0_R_p_c, cam1_R_p_c);                // :360 (the window is consumed even when nothing is tracked: the buffer must not grow)
    cam0_R_p_c_ = cam0_R_p_c; cam1_R_p_c_ = cam1_R_p_c;
    computeHpred(cam0_R_p_c, a.Hpred);
    std::memcpy(a.R_p_c[0], cam0_R_p_c.m, sizeof(a.R_p_c[0]));   // twoPointRansac's rotation compensation (:936-944), when it is on
    std::memcpy(a.R_p_c[1], cam1_R_p_c.m, sizeof(a.R_p_c[1]));
    const int cap = mskf_fe_grid_capacity(stream_);
    curr_.id.resize(cap); curr_.lifetime.resize(cap); curr_.code.resize(cap); curr_.response.resize(cap);
    curr_.cam0.resize(cap); curr_.cam1.resize(cap); curr_.und0.resize(cap); curr_.und1.resize(cap);
    a.capacity = cap;
    static_assert(sizeof(FeatureIDType) == sizeof(uint64_t), "id type");
    a.id = (uint64_t *)curr_.id.data(); a.lifetime = curr_.lifetime.data();
    a.cam0 = curr_.cam0.data(); a.cam1 = curr_.cam1.data(); a.und0 = curr_.und0.data(); a.und1 = curr_.und1.data();
    stage_ = 5;
    return true;
}

void ImageProcessor::frameEnd(const mskf_fe_frame_args &a, bool is_draw) {
    const size_t n = (size_t)a.n;
    curr_.id.resize(n); curr_.lifetime.resize(n); curr_.code.resize(n); curr_.response.resize(n);
    curr_.cam0.resize(n); curr_.cam1.resize(n); curr_.und0.resize(n); curr_.und1.resize(n);
    for (size_t k = 0; k < n; ++k) curr_.code[k] = gridCode(curr_.cam0[k]);      // (the host path's arrays stay complete: a later host frame may read them)
    before_tracking = a.before_tracking; after_tracking = a.after_tracking; after_matching = a.after_matching; after_ransac = a.after_ransac;
    next_feature_id = (FeatureIDType)a.next_feature_id;
    ransac_draws = (unsigned long long)a.ransac_draws;
    ++device_frames_;
    stage_ = 0;
    if (is_draw) {   // :163-184
        prev_ids_.assign(prev_.id.begin(), prev_.id.end());
        prev_cam0_points_.clear(); prev_cam1_points_.clear(); curr_cam0_points_.clear(); curr_cam1_points_.clear();
        for (size_t k = 0; k < prev_.size(); ++k) { prev_cam0_points_[prev_.id[k]] = Point2f(prev_.cam0[k].x, prev_.cam0[k].y); prev_cam1_points_[prev_.id[k]] = Point2f(prev_.cam1[k].x, prev_.cam1[k].y); }
        for (size_t k = 0; k < curr_.size(); ++k) { curr_cam0_points_[curr_.id[k]] = Point2f(curr_.cam0[k].x, curr_.cam0[k].y); curr_cam1_points_[curr_.id[k]] = Point2f(curr_.cam1[k].x, curr_.cam1[k].y); }
    }
    hostprof::Scope hp_pub(hostprof::FE_PUBLISH);
    publish();
    if (!(cfg_.compat_flags & MSKF_COMPAT_Q2_PREV_ALIAS)) cam0_prev_time = cam0_curr_time;
    std::swap(prev_, curr_);     // (the pyramid rotation of :194 is part of the device call)
}

void ImageProcessor::phaseAfter2(bool is_draw) {
    device_grid_valid_ = false;      // a host-side frame: the device's grid (if any) is stale from here on
    if (stage_ == 3) {
        { hostprof::Scope hp(hostprof::FE_NEW_TAIL); addNewFeaturesTail(); }
        { hostprof::Scope hp(hostprof::FE_PRUNE); assembleGrid(); }
    } else if (stage_ == 4) {
        assembleGrid();      // first frame: the new features are the grid (no pruning in initializeFirstFrame, grid_min <= grid_max)
    } else {
        curr_.clear();
    }
    stage_ = 0;
    if (is_draw) {   // :163-184
        prev_ids_.assign(prev_.id.begin(), prev_.id.end());
        prev_cam0_points_.clear(); prev_cam1_points_.clear(); curr_cam0_points_.clear(); curr_cam1_points_.clear();
        for (size_t k = 0; k < prev_.size(); ++k) { prev_cam0_points_[prev_.id[k]] = Point2f(prev_.cam0[k].x, prev_.cam0[k].y); prev_cam1_points_[prev_.id[k]] = Point2f(prev_.cam1[k].x, prev_.cam1[k].y); }
        for (size_t k = 0; k < curr_.size(); ++k) { curr_cam0_points_[curr_.id[k]] = Point2f(curr_.cam0[k].x, curr_.cam0[k].y); curr_cam1_points_[curr_.id[k]] = Point2f(curr_.cam1[k].x, curr_.cam1[k].y); }
    }
    hostprof::Scope hp_pub(hostprof::FE_PUBLISH);
    publish();
    // :192-200
    if (!(cfg_.compat_flags & MSKF_COMPAT_Q2_PREV_ALIAS)) cam0_prev_time = cam0_curr_time;
    std::swap(prev_, curr_);     // prev <- the grid just published
    mskf_fe_swap(stream_);
}

// :1137-1182
void ImageProcessor::publish() {
    feature_msg_ptr_->time_stamp = cam0_curr_time;
    std::vector<FeatureMeasurement> &msg = feature_msg_ptr_->features;
    if (!(cfg_.compat_flags & MSKF_COMPAT_Q1_MSG_ACCUMULATE)) msg.clear();
    const size_t n = curr_.size();
    // Q1: the reference push_backs one value-initialised record per feature and then writes records 0 .. n-1: the
    // message grows by n every frame and is never cleared
    if (compact_tail_) {
        if (!(cfg_.compat_flags & MSKF_COMPAT_Q1_MSG_ACCUMULATE)) logical_size_ = 0;
        logical_size_ += n;
        const size_t head = std::max(max_published_, n);            // live + stale records
        msg.resize(std::min(logical_size_, head + 1), FeatureMeasurement{0, 0, 0, 0, 0});
    } else
    msg.resize(msg.size() + n, FeatureMeasurement{0, 0, 0, 0, 0});
    for (size_t i = 0; i < n; ++i) {
        FeatureMeasurement &m = msg[i];
        m.id = (unsigned int)curr_.id[i];
        m.u0 = curr_.und0[i].x; m.v0 = curr_.und0[i].y; m.u1 = curr_.und1[i].x; m.v1 = curr_.und1[i].y;
    }
    if (n > max_published_) max_published_ = n;
    if (!(cfg_.compat_flags & MSKF_COMPAT_Q1_MSG_ACCUMULATE)) max_published_ = n;
    last_tracking_info = TrackingInfo{cam0_curr_time, before_tracking, after_tracking, after_matching, after_ransac};
    if (debug_.is_open())
        debug_ << std::fixed << std::setprecision(9) << cam0_curr_time << ": " << before_tracking << ", " << after_tracking << ", "
               << after_matching << ", " << after_ransac << std::endl;
}

void ImageProcessor::dumpCurrent(std::vector<FeatureIDType> &ids, std::vector<int> &lifetime, std::vector<Point2f> &cam0,
                                 std::vector<Point2f> &cam1) const {
    // after phaseAfter2 the published grid is prev_
    ids.assign(prev_.id.begin(), prev_.id.end());
    lifetime.assign(prev_.lifetime.begin(), prev_.lifetime.end());
    cam0.clear(); cam1.clear();
    for (size_t k = 0; k < prev_.size(); ++k) { cam0.push_back(Point2f(prev_.cam0[k].x, prev_.cam0[k].y)); cam1.push_back(Point2f(prev_.cam1[k].x, prev_.cam1[k].y)); }
}

}  // namespace cg
