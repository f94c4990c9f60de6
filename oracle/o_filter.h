// oracle/o_filter.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into the product.
//
// CPU restatement of cg::MsckfVio (msckf_core/include/msckf_vio.h:35-200, msckf_core/src/msckf_vio.cpp),
// cg::Feature (msckf_core/include/feature.hpp) and the state structs (common/imu_state.h, cam_state.h).
// Math follows SURVEY.md Appendix A.  Where the reference calls absent third-party code the
// substitution is: cg::svd_fulluv left null space (msckf_vio.cpp:757-766) -> Householder QR null space
// (same filter, isotropic noise); Eigen SPQR (:795-811) -> dense Householder QR; Eigen LDLT
// (:850,:924, feature.hpp:395) -> Cholesky; cg::chi_square_table_p95 (:184) -> scipy chi2.ppf table.
// parity unpinned: the reference has no tests or fixtures for this path.
#pragma once
#include <map>
#include <memory>
#include <vector>
#include "o_math.h"
#include "o_frontend.h"

namespace orc {

typedef long long StateIDType;
typedef long long FeatureIDType;

struct IMUState {  // common/imu_state.h:28-88 (class statics made per-filter members)
    StateIDType id = 0;
    double time = 0;
    Quat orientation;
    V3 position, velocity, gyro_bias, acc_bias;
    M3 R_imu_cam0 = M3::eye();
    V3 t_cam0_imu;
    Quat orientation_null;
    V3 position_null, velocity_null;
};

struct CAMState {  // common/cam_state.h:25-55
    StateIDType id = 0;
    double time = 0;
    Quat orientation;
    V3 position;
    Quat orientation_null;
    V3 position_null;
};
typedef std::map<StateIDType, CAMState> CamStateServer;

struct Feature {  // feature.hpp:31-163
    FeatureIDType id = 0;
    std::map<StateIDType, std::array<double, 4>> observations;
    V3 position;
    bool is_initialized = false;
};
typedef std::map<FeatureIDType, Feature> MapServer;

struct FilterShared {  // the reference's class statics (msckf_vio.cpp:33-47)
    double gyro_noise, acc_noise, gyro_bias_noise, acc_bias_noise, observation_noise;
    V3 gravity{0, 0, -9.81};
    SE3 T_imu_body, T_cam0_cam1;
    double feat_translation_threshold = 0.2, huber_epsilon = 0.01, estimation_precision = 5e-7, initial_damping = 1e-3;
    int outer_loop_max_iteration = 10, inner_loop_max_iteration = 10;
};

bool feature_check_motion(const Feature &f, const CamStateServer &cams, const FilterShared &sh);       // feature.hpp:257-287
bool feature_initialize_position(Feature &f, const CamStateServer &cams, const FilterShared &sh);     // feature.hpp:289-450

class MsckfVio {
  public:
    MsckfVio(const mskf_calib &calib, const mskf_ekf_cfg &cfg);
    void imuCallback(const mskf_imu_sample &msg);              // msckf_vio.cpp:190-207
    void featureCallback(const CameraMeasurement &msg);        // :306-375
    bool resetCallback();                                      // :243-304

    std::vector<mskf_pose> poses;                              // what publish() writes to pose_out.txt (:1256-1258)
    std::vector<V3> path_;

    // exposed for unit tests
    struct StateServer {
        IMUState imu_state;
        CamStateServer cam_states;
        Mat state_cov;
        Mat continuous_noise_cov;
    } state_server;
    MapServer map_server;
    FilterShared sh;
    void measurementJacobian(StateIDType cam_state_id, FeatureIDType feature_id, Mat &H_x, Mat &H_f, double r[4]);  // :610-677
    void featureJacobian(FeatureIDType feature_id, const std::vector<StateIDType> &cam_state_ids, Mat &H_x, std::vector<double> &r);  // :679-775
    void measurementUpdate(const Mat &H, const std::vector<double> &r);  // :778-907
    bool gatingTest(const Mat &H, const std::vector<double> &r, int dof);  // :909-935
    void processModel(double time, const V3 &m_gyro, const V3 &m_acc);   // :409-480
    void stateAugmentation(double time);                                // :533-585
    bool is_gravity_set = false;
    bool is_first_img = true;
    long long online_reset_counter = 0;
    int n_update = 0;

  private:
    void resetCov();
    void initializeGravityAndBias();
    void batchImuProcessing(double time_bound);
    void predictNewState(double dt, const V3 &gyro, const V3 &acc);
    void addFeatureObservations(const CameraMeasurement &msg);
    void removeLostFeatures();
    void findRedundantCamStates(std::vector<StateIDType> &rm);
    void pruneCamStateBuffer();
    void onlineReset();
    void publish(double t);

    mskf_calib calib_;
    mskf_ekf_cfg cfg_;
    StateIDType next_state_id = 0;
    double chi2_table[100];
    std::vector<mskf_imu_sample> imu_msg_buffer;
    double tracking_rate = 0;
};

}  // namespace orc
