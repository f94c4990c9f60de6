// oracle/o_frontend.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into the product.
//
// CPU restatement of cg::ImageProcessor (msckf_core/include/image_processor.h:27-367,
// msckf_core/src/image_processor.cpp).  Control flow, gates, grid logic and the quirks Q1-Q4/Q7
// of SURVEY.md §2.3 follow the reference line by line; pixel arithmetic is o_image.*.
// parity unpinned (see o_image.h).
#pragma once
#include <map>
#include <memory>
#include <vector>
#include "o_image.h"
#include "o_math.h"

namespace orc {

struct FeatureMetaData {  // image_processor.h:100-106
    unsigned long long id = 0;
    float response = 0.f;
    int lifetime = 0;
    mskf_point2f cam0_point{0.f, 0.f};
    mskf_point2f cam1_point{0.f, 0.f};
};
typedef std::map<int, std::vector<FeatureMetaData>> GridFeatures;  // image_processor.h:113

struct CameraMeasurement {  // data_msg.h:41-44
    double time_stamp = 0;
    std::vector<mskf_feature_meas> features;
};

// per-frame debug dump used by the parity tests (ids + float pixels of the live grid, flatten order)
struct FrameDump {
    std::vector<unsigned long long> ids;
    std::vector<int> lifetime;
    std::vector<mskf_point2f> cam0, cam1;
    mskf_tracking_info info;
};

class ImageProcessor {
  public:
    ImageProcessor(const mskf_calib &calib, const mskf_fe_cfg &cfg);
    void stereoCallback(const Img &cam0, const Img &cam1, double t0, double t1);  // image_processor.cpp:139-203
    void imuCallback(const mskf_imu_sample &msg);                                 // :205-211
    std::shared_ptr<CameraMeasurement> feature_msg_ptr_;                          // image_processor.h:58
    FrameDump last_dump;

    // exposed for unit tests
    void stereoMatch(const std::vector<mskf_point2f> &cam0_points, std::vector<mskf_point2f> &cam1_points,
                     std::vector<uint8_t> &inlier_markers);
    std::vector<Img> prev_cam0_pyramid_, curr_cam0_pyramid_, curr_cam1_pyramid_;
    void test_set_images(const Img &cam0, const Img &cam1) { cam0_curr_img = cam0; cam1_curr_img = cam1; createImagePyramids(); }

  private:
    void createImagePyramids();
    void initializeFirstFrame();
    void trackFeatures();
    void addNewFeatures();
    void pruneGridFeatures();
    void publish();
    void integrateImuData(M3 &cam0_R_p_c, M3 &cam1_R_p_c);
    void predictFeatureTracking(const std::vector<mskf_point2f> &in, const M3 &R_p_c, const double intr[4],
                                std::vector<mskf_point2f> &out);
    void undistortPoints(const std::vector<mskf_point2f> &in, const CamModel &cam, std::vector<mskf_point2f> &out,
                         const M3 &R = M3::eye());
    void distortPoints(const std::vector<mskf_point2f> &in, const CamModel &cam, std::vector<mskf_point2f> &out);
  public:
    // :911-1135 (dead code in the reference, Q5); draws come from a counter-based generator because
    // cg::uniform_integer lives in the absent vikit_cg
    void twoPointRansac(const std::vector<mskf_point2f> &pts1, const std::vector<mskf_point2f> &pts2, const M3 &R_p_c,
                        const CamModel &cam, double inlier_error, double success_probability, std::vector<int> &inlier_markers);
    unsigned long long ransac_draws = 0;   // state of the draw counter
  private:
    int uniformInteger(int lo, int hi);
    static void rescalePoints(std::vector<mskf_point2f> &pts1, std::vector<mskf_point2f> &pts2, float &scaling_factor);   // :888-908

    mskf_calib calib_;
    mskf_fe_cfg cfg_;
    bool is_first_img = true;
    unsigned long long next_feature_id = 0;  // Q3: defined as 0
    CornerDetector detector_;
    std::vector<mskf_imu_sample> imu_msg_buffer;
    CamModel cam0_, cam1_;
    M3 R_cam0_imu, R_cam1_imu;
    V3 t_cam0_imu, t_cam1_imu;
    Img cam0_curr_img, cam1_curr_img;
    double cam0_prev_time = 0, cam0_curr_time = 0;
    int grid_height = 0, grid_width = 0;  // Q7 statics -> per object
    std::shared_ptr<GridFeatures> prev_features_ptr, curr_features_ptr;
    int before_tracking = 0, after_tracking = 0, after_matching = 0, after_ransac = 0;
};

}  // namespace orc
