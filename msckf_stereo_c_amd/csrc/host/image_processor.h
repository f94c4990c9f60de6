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
    // (the detector floor: this class only ever asks for the cells above its threshold, image_processor.cpp:132)
    void attach(mskf_stream *s) { stream_ = s; if (s) mskf_fe_set_detect_floor(s, cfg_.fast_threshold * 256); }
    mskf_stream *stream() const { return stream_; }
    void phaseBegin(double time_stamp, int width, int height);      // timestamps, Q2 aliasing, grid size (Q7)
    void phasePrepare1(mskf_fe_track_args &args);                   // first frame: detections; else prev features
    void phaseAfter1(mskf_fe_track_args &args);                     // consume results; prepare new-feature candidates
    void phaseAfter2(bool is_draw);                                 // addNewFeatures tail, prune, publish, rotate
    bool isFirstImage() const { return is_first_img; }
    // ---- a whole frame on the device (mskf_fe_frame_batch_*): the bookkeeping of phaseAfter1 / phaseAfter2 runs in kernels
    // between the track calls, the grid stays in device memory; the host only integrates the gyro for the prediction, receives
    // the published grid and writes the message.  Possible for every frame but the first of a stream unless the configuration
    // needs the host between the tracks (2-point RANSAC) or exceeds the kernels' per-cell bound; MSKF_FE_BOOKS=host forces
    // the phased path.
    bool canDeviceFrame() const;
    static void setFeBooksOnHost(int on);   // 1: every frame on the phased (host bookkeeping) path, 0: device frames where possible, -1: MSKF_FE_BOOKS decides
    static bool feBooksOnHost();
    bool frameBegin(double time_stamp, mskf_fe_frame_args &args);   // phaseBegin + prediction + output arrays; false on error
    void frameEnd(const mskf_fe_frame_args &args, bool is_draw);    // take the published grid, publish(), rotate
    void enableFileOutputs() { if (!debug_.is_open()) debug_.open("debug_imageprocessor.txt"); }   // image_processor.cpp:134
    // records [zeroTailStart(), features.size()) of feature_msg_ptr_ were pushed but never written (Q1)
    size_t zeroTailStart() const { return max_published_; }
    // Q1 makes the message grow by a frame's features every frame, for ever (a 2000-frame run: 35 MB per stream, every frame's
    // records written into fresh pages).  Everything behind zeroTailStart() is value-initialised records that are never written, so
    // a caller that knows the convention (the batch runner; MsckfVio::setZeroTailHint collapses that tail anyway) may ask for the tail to
    // be kept as a COUNT: the vector then holds the live and stale records plus the first tail record, messageSize() is what
    // features.size() would be.  Off by default: the public member behaves as in the reference.
    void setCompactTail(bool on) { compact_tail_ = on; }
    size_t messageSize() const { return compact_tail_ ? logical_size_ : feature_msg_ptr_->features.size(); }

    // debug / parity: live grid in flatten order
    void dumpCurrent(std::vector<FeatureIDType> &ids, std::vector<int> &lifetime, std::vector<Point2f> &cam0,
                     std::vector<Point2f> &cam1) const;
    TrackingInfo last_tracking_info{0, 0, 0, 0, 0};
    const std::string &error() const { return error_; }

  private:
    // The reference keeps the live features in GridFeatures = std::map<int code, std::vector<FeatureMetaData>>
    // (image_processor.h:100-113): iteration = ascending grid code, insertion order inside a cell.  Here the same
    // sequence is ONE set of flat arrays in that iteration order (a frame's grid is rebuilt by a stable counting sort
    // over the codes): no per-cell containers, the previous frame's pixels go to the device as they lie.
    struct FeatureArrays {
        std::vector<FeatureIDType> id;
        std::vector<int> lifetime, code;
        std::vector<float> response;
        std::vector<mskf_point2f> cam0, cam1, und0, und1;   // pixels; undistorted normalised coordinates (what publish() sends)
        size_t size() const { return id.size(); }
        void clear() { id.clear(); lifetime.clear(); code.clear(); response.clear(); cam0.clear(); cam1.clear(); und0.clear(); und1.clear(); }
        void push(FeatureIDType i, int life, int c, float resp, mskf_point2f a, mskf_point2f b, mskf_point2f ua, mskf_point2f ub) {
            id.push_back(i); lifetime.push_back(life); code.push_back(c); response.push_back(resp);
            cam0.push_back(a); cam1.push_back(b); und0.push_back(ua); und1.push_back(ub);
        }
        void push_from(const FeatureArrays &o, size_t k) { push(o.id[k], o.lifetime[k], o.code[k], o.response[k], o.cam0[k], o.cam1[k], o.und0[k], o.und1[k]); }
    };

    bool loadParameters();
    void detectFeatures(std::vector<Point2f> &pts, std::vector<double> &responses);  // CornerDetector::detect_features
    void setGridPosition(float x, float y);                                         // CornerDetector::set_grid_position
    void integrateImuData(hm::Mat3 &cam0_R_p_c, hm::Mat3 &cam1_R_p_c);
    void computeHpred(const hm::Mat3 &R_p_c, double H[9]) const;
  public:
    // twoPointRansac (image_processor.cpp:911-1135; dead code in the reference, runs when MSKF_COMPAT_Q5_NO_RANSAC is
    // cleared).  pts1 / pts2: previous / current points ALREADY undistorted to normalised coordinates (the device
    // returns them with every track, :929-930 happen there).  Host code: a few hundred points, seven sequential
    // hypotheses, index-order sums — see DESIGN.md, "Out of scope".
    void twoPointRansac(const std::vector<cg::Point2f> &pts1_undistorted, const std::vector<cg::Point2f> &pts2_undistorted,
                        const hm::Mat3 &R_p_c, const double intrinsics[4], double inlier_error, double success_probability,
                        std::vector<int> &inlier_markers);
    unsigned long long ransac_draws = 0;    // state of the counter-based draw generator (cg::uniform_integer is in the absent vikit_cg)
  private:
    void initializeFirstFrameTail();
    void trackFeaturesTail();
    void addNewFeaturesHead();
    void addNewFeaturesTail();
    void selectNewFeatures(bool first_frame);   // per-cell response sort + vacancy fill + ids (:690-750 / :270-316)
    void assembleGrid();                         // grid of this frame in iteration order + pruneGridFeatures (:758-768)
    int gridCode(const mskf_point2f &p) const {  // :452-454 (Q7: the column may equal grid_col)
        return static_cast<int>(p.y / grid_height) * cfg_.grid_col + static_cast<int>(p.x / grid_width);
    }
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
    FeatureArrays prev_, curr_, tracked_, new_;      // published grid of the last frame / of this frame; this frame's survivors (track order); new features (code order)
    std::vector<int> cell_count_, cell_start_;       // tracked features per grid code of this frame; counting-sort offsets
    int n_codes_ = 0;                                 // grid codes that can occur: 0 .. n_codes_ - 1
    int before_tracking = 0, after_tracking = 0, after_matching = 0, after_ransac = 0;

    // per-frame scratch shared between phases
    std::vector<mskf_point2f> in_pts_, out0_, out1_, und0_, und1_;
    std::vector<uint8_t> status_;
    hm::Mat3 cam0_R_p_c_, cam1_R_p_c_;            // integrateImuData result of this frame
    std::vector<double> cand_responses_det_;     // responses in detection order (Q4)
    std::vector<double> cand_responses_sieved_;  // responses in sieve order
    std::vector<int> cand_index_;                // position of every candidate sent to the device in the reference's full candidate list
    std::vector<mskf_corner> cell_max_;
    std::vector<std::vector<std::pair<mskf_point2f, double>>> sieve_;   // candidates of the cells with a vacancy (per-frame scratch)
    std::vector<int> sieve_count_;                                       // candidates per grid cell (all cells)
    std::vector<int> order_;                                             // scratch: sort permutation
    size_t max_published_ = 0;
    bool compact_tail_ = false;
    size_t logical_size_ = 0;
    int stage_ = 0;   // 0 idle, 1 first-frame stereo pending, 2 temporal pending, 3 candidates pending
    bool device_grid_valid_ = false;   // the device's grid is the one this object published last (false after a host-side frame)
    long long device_frames_ = 0;      // frames that ran as one device call (mskf_fe_frame_batch_*)
  public:
    long long deviceFrames() const { return device_frames_; }
  private:
    std::ofstream debug_;
};

typedef ImageProcessor::Ptr ImageProcessorPtr;
typedef ImageProcessor::ConstPtr ImageProcessorConstPtr;

}  // namespace cg
