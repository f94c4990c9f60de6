// msckf_vio.h — host mirror of cg::MsckfVio (reference msckf_core/include/msckf_vio.h:35-200) with the
// state structs of common/imu_state.h, common/cam_state.h and the Feature bookkeeping of feature.hpp.
//
// Same public surface (ctor from the camchain YAML node, initialize(), resetCallback(), imuCallback(),
// featureCallback(), get_path(), get_points3d()).  The covariance never leaves the GPU: propagation,
// augmentation, triangulation, Jacobians, gating, QR and the Kalman update run behind the C-ABI
// (include/mskf_hip.h); this class integrates the 16-dim nominal state, keeps the clone / feature maps
// and applies the returned correction vector.  featureCallback() is phaseA -> update -> phaseB ->
// update -> phaseC; BatchRunner drives the same phases for many streams with batched device calls.
#pragma once
#include <array>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../../include/mskf_hip.h"
#include "cg_types.h"
#include "feature_store.h"
#include "yaml_lite.h"

namespace cg {

mskf_ekf_cfg ekf_cfg_from_yaml(const YAML::Node &cfg_msckfvio);

struct Quat { double q[4] = {0, 0, 0, 1}; };   // JPL [x y z w]

struct IMUState {   // common/imu_state.h:28-88; the class statics live in MsckfVio (per stream)
    StateIDType id = 0;
    double time = 0;
    Quat orientation;
    Vector3 position, velocity, gyro_bias, acc_bias;
    hm::Mat3 R_imu_cam0 = hm::Mat3::identity();
    Vector3 t_cam0_imu;
    Quat orientation_null;
    Vector3 position_null, velocity_null;
};

struct CAMState {   // common/cam_state.h:25-55
    StateIDType id = 0;
    double time = 0;
    Quat orientation;
    Vector3 position;
    Quat orientation_null;
    Vector3 position_null;
    int slot = -1;      // row of the observation table that holds this clone's observations (feature_store.h)
};
typedef std::map<StateIDType, CAMState> CamStateServer;
// Feature / MapServer (feature.hpp:31-168): see feature_store.h — flat tables with the maps' semantics

class MsckfVio {
  public:
    explicit MsckfVio(YAML::Node cfg_cam_imu);
    MsckfVio(const mskf_calib &calib, const mskf_ekf_cfg &cfg);
    MsckfVio(const MsckfVio &) = delete;
    MsckfVio operator=(const MsckfVio &) = delete;
    ~MsckfVio();

    bool initialize();
    bool resetCallback();
    void imuCallback(const cg::ImuConstPtr &msg);
    void featureCallback(const CameraMeasurementConstPtr &msg);
    std::vector<cg::Vector3> get_path() { return path_; }
    std::vector<cg::Point3f> get_points3d() { return points3d_; }

    typedef std::shared_ptr<MsckfVio> Ptr;
    typedef std::shared_ptr<const MsckfVio> ConstPtr;

    // ---- device attachment + phased interface
    void attach(mskf_stream *s) { stream_ = s; }
    mskf_stream *stream() const { return stream_; }
    // phase A: IMU propagation, augmentation, observations; fills the lost-feature update (n_feat may be 0)
    // defer_device: do not issue the propagation / augmentation here; the caller batches them from
    // predictSteps() / predictJ() with mskf_ekf_predict_batch before running the update
    bool phaseA(const CameraMeasurementConstPtr &msg, mskf_ekf_update_args &upd, bool defer_device = false);
    const std::vector<mskf_imu_step> &predictSteps() const { return imu_steps_; }
    const double *predictJ() const { return have_J_ ? J_ : nullptr; }
    // phase B: apply the lost-feature update, then prepare the pruning update (n_feat may be 0)
    void phaseB(mskf_ekf_update_args &upd);
    // phase C: apply the pruning update, delete clones, publish; returns false if nothing ran this frame
    // defer_device: the caller removes the clones listed in pendingRemovals() with mskf_ekf_remove_clones_batch
    void phaseC(bool defer_device = false);
    const int32_t *pendingRemovals() const { return pending_rm_; }   // two clone indices (current state order) or -1
    // phase D: online reset decision from the position variances (msckf_vio.cpp:1186-1236)
    void phaseD(const double pos_var[3]);
    // the position variances the last update of this frame brought back (mskf_ekf_update_args.pos_var_out), if any
    bool havePosVar() const { return pos_var_valid_; }
    const double *posVar() const { return pos_var_; }
    bool frameActive() const { return frame_active_; }
    void enableFileOutputs() {   // msckf_vio.cpp:169-171
        if (!pose_outfile_.is_open()) pose_outfile_.open("pose_out.txt");
        if (!debug_.is_open()) debug_.open("debug_msckfvio.txt");
    }
    // optional: records [start, size) of `msg` are untouched value-initialised records (Q1 tail)
    // total_size > msg->features.size(): `msg` is a snapshot truncated inside the zero tail of a longer message
    void setZeroTailHint(const CameraMeasurement *msg, size_t start, size_t total_size = 0) {
        zero_tail_msg_ = msg; zero_tail_start_ = start; zero_tail_total_ = total_size;
    }

    const std::vector<mskf_pose> &poses() const { return poses_; }
    const IMUState &imuState() const { return state_server.imu_state; }
    int numClones() const { return (int)state_server.cam_states.size(); }
    int numUpdates() const { return n_update_; }
    // diagnostics of the QR compression (include/mskf_hip.h diag_out): updates that ran as Householder TSQR, sum of stacked rows
    int numTsqrUpdates() const { return n_tsqr_; }
    int numUncompressedUpdates() const { return n_direct_; }
    long long stackedRows() const { return rows_sum_; }
    long long numResets() const { return online_reset_counter_; }
    size_t mapSize() const { return map_server.size(); }
    bool keepTrajectory = true;   // path_/points3d_ grow forever in the reference (Q20); benches may switch it off
    const std::string &error() const { return error_; }

  private:
    struct StateServer {
        IMUState imu_state;
        CamStateServer cam_states;
    } state_server;

    bool loadParameters();
    void resetCov();
    void initializeGravityAndBias();
    void batchImuProcessing(double time_bound);
    void predictNewState(double dt, const Vector3 &gyro, const Vector3 &acc);
    void stateAugmentation(double time);
    void addFeatureObservations(const CameraMeasurementConstPtr &msg);
    void buildLostFeatureUpdate(mskf_ekf_update_args &upd);
    void findRedundantCamStates(std::vector<StateIDType> &rm);
    void buildPruneUpdate(mskf_ekf_update_args &upd);
    void applyCorrection(const std::vector<double> &delta_x);
    void publish(double time_stamp);
    bool checkMotion(int slot, uint64_t mask) const;
    void packClones();
    int allocCloneSlot();
    void resetCloneSlots();
    void finishArgs(mskf_ekf_update_args &upd, int dof_offset, int apply_cap);
    void fail(const char *what, int rc);
    void dumpFeatureJacobians();   // debug_msckfvio.txt, frame n_pub == 9 (msckf_vio.cpp:719-723)

    YAML::Node cfg_cam_imu_;
    bool have_yaml_ = false;
    mskf_calib calib_;
    mskf_ekf_cfg cfg_;
    mskf_stream *stream_ = nullptr;
    std::string error_;

    // the reference's class statics (msckf_vio.cpp:33-47), per stream here
    StateIDType next_state_id_ = 0;
    double gyro_noise_ = 0, acc_noise_ = 0, gyro_bias_noise_ = 0, acc_bias_noise_ = 0, observation_noise_ = 0;
    Vector3 gravity_{0, 0, -9.81};
    hm::Rigid T_imu_body_, T_cam0_cam1_;
    double feat_translation_threshold_ = 0.2;

    std::vector<cg::Vector3> path_;
    std::vector<cg::Point3f> points3d_;
    std::vector<mskf_pose> poses_;
    MapServer map_server;
    std::vector<cg::Imu> imu_msg_buffer;
    bool is_gravity_set = false;
    bool is_first_img = true;
    double tracking_rate = 0;
    int n_update_ = 0;
    long long online_reset_counter_ = 0;
    bool frame_active_ = false;
    double frame_time_ = 0;

    // per-frame staging shared between phases
    std::vector<mskf_imu_step> imu_steps_;
    std::vector<mskf_clone_state> clones_;
    std::vector<mskf_ekf_feature> feats_;
    std::vector<int> feat_slots_;                          // map_server slots of feats_
    std::vector<size_t> erase_ranks_;                      // map_server ranks to erase after the lost-feature update (ascending)
    std::vector<int> order_slot_;                          // observation-table row of the k-th oldest clone (packClones)
    std::vector<int> free_clone_slots_;
    std::vector<int32_t> obs_clone_;
    std::vector<double> obs_z_;
    std::vector<double> delta_x_, gamma_;
    std::vector<uint8_t> feat_status_;
    int32_t rows_out_ = 0;
    int32_t diag_out_[2] = {0, 0};
    double pos_var_[3] = {-1, -1, -1}, pos_var_buf_[3] = {-1, -1, -1};
    bool pos_var_valid_ = false;
    void takePosVar() { if (pos_var_buf_[0] >= 0) { pos_var_[0] = pos_var_buf_[0]; pos_var_[1] = pos_var_buf_[1]; pos_var_[2] = pos_var_buf_[2]; pos_var_valid_ = true; } }
    int n_tsqr_ = 0, n_direct_ = 0;
    long long rows_sum_ = 0;
    std::vector<StateIDType> rm_cam_state_ids_;
    bool prune_pending_ = false;
    bool defer_device_ = false, have_J_ = false;
    double J_[6 * 21];
    int32_t pending_rm_[2] = {-1, -1};
    int rm_order_[2] = {-1, -1};                           // window positions of the two clones being pruned
    const CameraMeasurement *zero_tail_msg_ = nullptr;
    size_t zero_tail_start_ = 0, zero_tail_total_ = 0;
    std::ofstream pose_outfile_, debug_;
    int n_pub_ = 0;                                        // published frames (msckf_vio.cpp:356)
};

typedef MsckfVio::Ptr MsckfVioPtr;
typedef MsckfVio::ConstPtr MsckfVioConstPtr;

}  // namespace cg
