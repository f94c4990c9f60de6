// image_processor.h — host mirror of cg::ImageProcessor (reference msckf_core/include/image_processor.h:27-367).
//
// Same public surface (ctor from the camchain YAML node, initialize(), stereoCallback(), imuCallback(),
// feature_msg_ptr_, the five id/point containers, processor_config).  All pixel work goes through the
// C-ABI (include/mskf_hip.h): pyramids, detector maxima, pyramidal LK + stereo matching + gates run
// as HIP kernels; this class keeps the reference's bookkeeping (grid buckets, sorting, ids, message).
// stereoCallback() is exactly phasePush -> phasePrepare1 -> track -> phaseAfter1 -> track -> phaseAfter2;
// BatchRunner drives the same phases for many streams with one batched device call per phase.
#pragma once
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../../include/mskf_hip.h"
#include "cg_types.h"
#include "yaml_lite.h"

namespace cg {

mskf_calib calib_from_yaml(const YAML::Node &cfg_cam_imu);
mskf_fe_cfg fe_cfg_from_yaml(const YAML::Node &cfg_imgproc);

// twoPointRansac of image_processor.cpp:911-1135 on points already undistorted to normalised coordinates; `ransac_draws`
// is the state of the counter-based draw generator (see ImageProcessor::twoPointRansac)
void two_point_ransac(const std::vector<Point2f> &pts1_undistorted, const std::vector<Point2f> &pts2_undistorted,
                      const hm::Mat3 &R_p_c, const double intrinsics[4], double inlier_error, double success_probability,
                      unsigned long long &ransac_draws, std::vector<int> &inlier_markers);

class ImageProcessor {
  public:
    // reference constructor (image_processor.cpp:32-42); config paths as in the reference (Q16)
    explicit ImageProcessor(YAML::Node cfg_cam_imu);
    // explicit-configuration constructor used by System / BatchRunner
    ImageProcessor(const mskf_calib &calib, const mskf_fe_cfg &cfg);
    ImageProcessor(const ImageProcessor &) = delete;
    ImageProcessor operator=(const ImageProcessor &) = delete;
    ~ImageProcessor();

    bool initialize();
    void stereoCallback(const cg::Image &cam0_img, const cg::Image &cam1_img, bool is_draw = false);
    void imuCallback(const cg::ImuConstPtr &msg);

    std::shared_ptr<CameraMeasurement> feature_msg_ptr_;

    typedef unsigned long long int FeatureIDType;
    std::vector<FeatureIDType> prev_ids_;
    std::map<FeatureIDType, cg::Point2f> prev_cam0_points_;
    std::map<FeatureIDType, cg::Point2f> prev_cam1_points_;
    std::map<FeatureIDType, cg::Point2f> curr_cam0_points_;
    std::map<FeatureIDType, cg::Point2f> curr_cam1_points_;

    struct ProcessorConfig {
        int grid_row;
        int grid_col;
        int grid_min_feature_num;
        int grid_max_feature_num;
        int pyramid_levels;
        int patch_size;
        int fast_threshold;
        int max_iteration;
        double track_precision;
        double ransac_threshold;
        double stereo_threshold;
    };
    ProcessorConfig processor_config;

    typedef std::shared_ptr<ImageProcessor> Ptr;
    typedef std::shared_ptr<const ImageProcessor> ConstPtr;

    // ---- device attachment + phased interface (BatchRunner)
    void attach(mskf_stream *s) { stream_ = s; }
    mskf_stream *stream() const { return stream_; }
    void phaseBegin(double time_stamp, int width, int height);      // timestamps, Q2 aliasing, grid size (Q7)
    void phasePrepare1(mskf_fe_track_args &args);                   // first frame: detections; else prev features
    void phaseAfter1(mskf_fe_track_args &args);                     // consume results; prepare new-feature candidates
    void phaseAfter2(bool is_draw);                                 // addNewFeatures tail, prune, publish, rotate
    bool isFirstImage() const { return is_first_img; }
    void enableFileOutputs() { if (!debug_.is_open()) debug_.open("debug_imageprocessor.txt"); }   // image_processor.cpp:134
    // records [zeroTailStart(), features.size()) of feature_msg_ptr_ were pushed but never written (Q1)
    size_t zeroTailStart() const { return max_published_; }

    // debug / parity: live grid in flatten order
    void dumpCurrent(std::vector<FeatureIDType> &ids, std::vector<int> &lifetime, std::vector<Point2f> &cam0,
                     std::vector<Point2f> &cam1) const;
    TrackingInfo last_tracking_info{0, 0, 0, 0, 0};
    const std::string &error() const { return error_; }

  private:
    struct FeatureMetaData {
        FeatureIDType id;
        float response;
        int lifetime;
        cg::Point2f cam0_point;
        cg::Point2f cam1_point;
        cg::Point2f und0, und1;   // undistorted normalised coordinates (what publish() sends)
    };
    typedef std::map<int, std::vector<FeatureMetaData>> GridFeatures;

    bool loadParameters();
    void detectFeatures(std::vector<Point2f> &pts, std::vector<double> &responses);  // CornerDetector::detect_features
    void setGridPosition(float x, float y);                                         // CornerDetector::set_grid_position
    void integrateImuData(hm::Mat3 &cam0_R_p_c, hm::Mat3 &cam1_R_p_c);
    void computeHpred(const hm::Mat3 &R_p_c, double H[9]) const;
  public:
    // twoPointRansac (image_processor.cpp:911-1135; dead code in the reference, runs when MSKF_COMPAT_Q5_NO_RANSAC is
    // cleared).  pts1 / pts2: previous / current points ALREADY undistorted to normalised coordinates (the device
    // returns them with every track, :929-930 happen there).  Host code: a few hundred points, seven sequential
    // hypotheses, index-order sums — see DESIGN.md section 7.
    void twoPointRansac(const std::vector<cg::Point2f> &pts1_undistorted, const std::vector<cg::Point2f> &pts2_undistorted,
                        const hm::Mat3 &R_p_c, const double intrinsics[4], double inlier_error, double success_probability,
                        std::vector<int> &inlier_markers);
    unsigned long long ransac_draws = 0;    // state of the counter-based draw generator (cg::uniform_integer is in the absent vikit_cg)
  private:
    void initializeFirstFrameTail();
    void trackFeaturesTail();
    void addNewFeaturesHead();
    void addNewFeaturesTail();
    void pruneGridFeatures();
    void resetGrid(GridFeatures &g) const;
    void publish();
    void fail(const char *what, int rc);

    YAML::Node cfg_cam_imu_;
    bool have_yaml_ = false;
    mskf_calib calib_;
    mskf_fe_cfg cfg_;
    mskf_stream *stream_ = nullptr;
    mskf_ctx *own_ctx_ = nullptr;   // only when constructed stand-alone
    bool own_stream_ = false;
    std::string error_;

    bool is_first_img = true;
    FeatureIDType next_feature_id = 0;   // Q3: defined as 0
    std::vector<cg::Imu> imu_msg_buffer;
    hm::Mat3 R_cam0_imu, R_cam1_imu;
    hm::Vec3 t_cam0_imu, t_cam1_imu;
    double cam0_prev_time = 0, cam0_curr_time = 0;
    int img_w = 0, img_h = 0;
    int grid_height = 0, grid_width = 0;
    int det_cell_w = 0, det_cell_h = 0;
    std::vector<uint8_t> occupancy_;
    std::shared_ptr<GridFeatures> prev_features_ptr, curr_features_ptr;
    int before_tracking = 0, after_tracking = 0, after_matching = 0, after_ransac = 0;

    // per-frame scratch shared between phases
    std::vector<mskf_point2f> in_pts_, out0_, out1_, und0_, und1_;
    std::vector<uint8_t> status_;
    std::vector<FeatureIDType> t_ids_;
    std::vector<int> t_lifetime_;
    std::vector<cg::Point2f> t_und0_, t_und1_;    // undistorted previous points of the tracked features (RANSAC input)
    hm::Mat3 cam0_R_p_c_, cam1_R_p_c_;            // integrateImuData result of this frame
    std::vector<double> cand_responses_det_;     // responses in detection order (Q4)
    std::vector<double> cand_responses_sieved_;  // responses in sieve order
    std::vector<int> cand_index_;                // position of every candidate sent to the device in the reference's full candidate list
    std::vector<mskf_corner> cell_max_;
    GridFeatures grid_new_features_;                                     // per-frame scratch, storage reused
    std::vector<std::vector<std::pair<Point2f, double>>> sieve_;         // per-frame scratch, storage reused
    size_t max_published_ = 0;
    int stage_ = 0;   // 0 idle, 1 first-frame stereo pending, 2 temporal pending, 3 candidates pending
    std::ofstream debug_;
};

typedef ImageProcessor::Ptr ImageProcessorPtr;
typedef ImageProcessor::ConstPtr ImageProcessorConstPtr;

}  // namespace cg
