// msckf_vio.cpp — host mirror of cg::MsckfVio; see msckf_vio.h.
// Reference: msckf_core/src/msckf_vio.cpp and msckf_core/include/feature.hpp (lines cited per function).
#include "msckf_vio.h"
#include "host_prof.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iostream>
#include "image_processor.h"
#include "kinematics.h"

namespace cg {

// config/app_msckfvio.yaml keys (msckf_vio.cpp:58-128)
mskf_ekf_cfg ekf_cfg_from_yaml(const YAML::Node &y) {
    mskf_ekf_cfg c;
    std::memset(&c, 0, sizeof(c));
    c.frame_rate = y["frame_rate"].as<double>();
    c.position_std_threshold = y["position_std_threshold"].as<double>();
    c.rotation_threshold = y["rotation_threshold"].as<double>();
    c.translation_threshold = y["translation_threshold"].as<double>();
    c.tracking_rate_threshold = y["tracking_rate_threshold"].as<double>();
    c.feature_translation_threshold = y["feature/config/translation_threshold"].as<double>();
    c.noise_gyro = y["noise/gyro"].as<double>();
    c.noise_acc = y["noise/acc"].as<double>();
    c.noise_gyro_bias = y["noise/gyro_bias"].as<double>();
    c.noise_acc_bias = y["noise/acc_bias"].as<double>();
    c.noise_feature = y["noise/feature"].as<double>();
    std::vector<double> v = y["initial_state/velocity"].as<std::vector<double>>();
    for (int i = 0; i < 3 && i < (int)v.size(); ++i) c.init_velocity[i] = v[i];
    c.cov_velocity = y["initial_covariance/velocity"].as<double>();
    c.cov_gyro_bias = y["initial_covariance/gyro_bias"].as<double>();
    c.cov_acc_bias = y["initial_covariance/acc_bias"].as<double>();
    c.cov_ext_rot = y["initial_covariance/extrinsic_rotation_cov"].as<double>();
    c.cov_ext_trans = y["initial_covariance/extrinsic_translation_cov"].as<double>();
    c.max_cam_state_size = (int)y["max_cam_state_size"].as<double>();
    c.chi2_mode = 0;
    c.max_stack_rows = 1500;   // msckf_vio.cpp:1009
    // how the stacked Jacobian is compressed (not a key of the reference's file): by default the reference's own rule, Householder
    // QR when there are more rows than columns (:795-821); "compression_mode: 0" selects Gram + regularised Cholesky (include/mskf_types.h)
    c.compression_mode = y["compression_mode"].IsDefined() ? y["compression_mode"].as<int>() : 3;
    return c;
}

MsckfVio::MsckfVio(YAML::Node cfg_cam_imu) : cfg_cam_imu_(cfg_cam_imu), have_yaml_(true) {
    std::memset(&calib_, 0, sizeof(calib_));
    std::memset(&cfg_, 0, sizeof(cfg_));
}

MsckfVio::MsckfVio(const mskf_calib &calib, const mskf_ekf_cfg &cfg) : calib_(calib), cfg_(cfg) {}

MsckfVio::~MsckfVio() {
    if (pose_outfile_.is_open()) pose_outfile_.close();
    if (debug_.is_open()) debug_.close();
}

void MsckfVio::fail(const char *what, int rc) {
    error_ = std::string(what) + ": " + mskf_last_error();
    std::fprintf(stderr, "MsckfVio: %s failed (%d): %s\n", what, rc, mskf_last_error());
}

// msckf_vio.cpp:58-162
bool MsckfVio::loadParameters() {
    if (have_yaml_) {
        calib_ = calib_from_yaml(cfg_cam_imu_);
        YAML::Node y = YAML::LoadFile("../config/app_msckfvio.yaml");   // Q16
        cfg_ = ekf_cfg_from_yaml(y);
    }
    feat_translation_threshold_ = cfg_.feature_translation_threshold;
    gyro_noise_ = cfg_.noise_gyro * cfg_.noise_gyro;            // :77-81 variance, not std
    acc_noise_ = cfg_.noise_acc * cfg_.noise_acc;
    gyro_bias_noise_ = cfg_.noise_gyro_bias * cfg_.noise_gyro_bias;
    acc_bias_noise_ = cfg_.noise_acc_bias * cfg_.noise_acc_bias;
    observation_noise_ = cfg_.noise_feature * cfg_.noise_feature;
    state_server.imu_state.velocity = Vector3(cfg_.init_velocity[0], cfg_.init_velocity[1], cfg_.init_velocity[2]);
    hm::Rigid T_cam0_imu = hm::Rigid::from_rowmajor16(calib_.T_cam0_imu).inverse();   // :115-119
    state_server.imu_state.R_imu_cam0 = T_cam0_imu.R.transpose();
    state_server.imu_state.t_cam0_imu = T_cam0_imu.t;
    T_cam0_cam1_ = hm::Rigid::from_rowmajor16(calib_.T_cam1_cam0);                      // :121-122
    T_imu_body_ = hm::Rigid::from_rowmajor16(calib_.T_imu_body).inverse();             // :124-125
    return true;
}

// :102-112
void MsckfVio::resetCov() {
    double P0[21 * 21];
    std::memset(P0, 0, sizeof(P0));
    for (int i = 3; i < 6; ++i) P0[i * 21 + i] = cfg_.cov_gyro_bias;
    for (int i = 6; i < 9; ++i) P0[i * 21 + i] = cfg_.cov_velocity;
    for (int i = 9; i < 12; ++i) P0[i * 21 + i] = cfg_.cov_acc_bias;
    for (int i = 15; i < 18; ++i) P0[i * 21 + i] = cfg_.cov_ext_rot;
    for (int i = 18; i < 21; ++i) P0[i * 21 + i] = cfg_.cov_ext_trans;
    if (stream_) {
        int rc = mskf_ekf_reset(stream_, P0);
        if (rc != MSKF_OK) fail("mskf_ekf_reset", rc);
    }
}

// :164-188
bool MsckfVio::initialize() {
    if (!loadParameters()) return false;
    if (!stream_) { error_ = "MsckfVio: no device stream attached"; return false; }
    resetCloneSlots();
    resetCov();
    if (have_yaml_) enableFileOutputs();
    return error_.empty();
}

// :190-207
void MsckfVio::imuCallback(const cg::ImuConstPtr &msg) {
    imu_msg_buffer.push_back(*msg);
    if (!is_gravity_set) {
        if (imu_msg_buffer.size() < 200) return;
        initializeGravityAndBias();
        is_gravity_set = true;
    }
}

// :209-241
void MsckfVio::initializeGravityAndBias() {
    Vector3 sum_w, sum_a;
    for (const auto &m : imu_msg_buffer) { sum_w = sum_w + m.angular_velocity; sum_a = sum_a + m.linear_acceleration; }
    state_server.imu_state.gyro_bias = sum_w / (double)imu_msg_buffer.size();
    Vector3 gravity_imu = sum_a / (double)imu_msg_buffer.size();
    const double gn = hm::norm(gravity_imu);
    gravity_ = Vector3(0.0, 0.0, -gn);
    kin::quaternion_of(kin::shortest_arc(gravity_imu, -gravity_).transpose(), state_server.imu_state.orientation.q);
}

// :243-304
bool MsckfVio::resetCallback() {
    IMUState &s = state_server.imu_state;
    s.time = 0.0;
    s.orientation = Quat(); s.position = Vector3(); s.velocity = Vector3(); s.gyro_bias = Vector3(); s.acc_bias = Vector3();
    s.orientation_null = Quat(); s.position_null = Vector3(); s.velocity_null = Vector3();
    state_server.cam_states.clear();
    resetCloneSlots();
    resetCov();
    map_server.clear();
    imu_msg_buffer.clear();
    is_gravity_set = false;
    is_first_img = true;
    return true;
}

// :306-375
void MsckfVio::featureCallback(const CameraMeasurementConstPtr &msg) {
    mskf_ekf_update_args upd;
    if (!phaseA(msg, upd)) return;
    if (upd.n_feat > 0) {
        int rc = mskf_ekf_update(stream_, &upd);
        if (rc != MSKF_OK) { fail("mskf_ekf_update", rc); return; }
    }
    phaseB(upd);
    if (upd.n_feat > 0) {
        int rc = mskf_ekf_update(stream_, &upd);
        if (rc != MSKF_OK) { fail("mskf_ekf_update", rc); return; }
    }
    phaseC();
    if (cfg_.position_std_threshold > 0) {
        if (pos_var_valid_) { phaseD(pos_var_); return; }      // came back with the frame's last update (clone removal does not touch it)
        double pv[3];
        int rc = mskf_ekf_get_pos_var(stream_, pv);
        if (rc != MSKF_OK) { fail("mskf_ekf_get_pos_var", rc); return; }
        phaseD(pv);
    }
}

bool MsckfVio::phaseA(const CameraMeasurementConstPtr &msg, mskf_ekf_update_args &upd, bool defer_device) {
    std::memset(&upd, 0, sizeof(upd));
    frame_active_ = false;
    pos_var_valid_ = false;
    defer_device_ = defer_device;
    have_J_ = false;
    imu_steps_.clear();
    if (!is_gravity_set) return false;
    if (is_first_img) { is_first_img = false; state_server.imu_state.time = msg->time_stamp; }
    frame_active_ = true;
    frame_time_ = msg->time_stamp;
    { hostprof::Scope hp(hostprof::EKF_IMU); batchImuProcessing(msg->time_stamp); }
    { hostprof::Scope hp(hostprof::EKF_AUGMENT); stateAugmentation(msg->time_stamp); }
    { hostprof::Scope hp(hostprof::EKF_ADD_OBS); addFeatureObservations(msg); }
    { hostprof::Scope hp(hostprof::EKF_BUILD_LOST); buildLostFeatureUpdate(upd); }
    return true;
}

// :377-407 + the state part of processModel (:409-480); the covariance part runs on the device
void MsckfVio::batchImuProcessing(double time_bound) {
    int used = 0;
    imu_steps_.clear();
    IMUState &s = state_server.imu_state;
    for (const auto &m : imu_msg_buffer) {
        const double t = m.time_stamp;
        if (t < s.time) { ++used; continue; }
        if (t > time_bound) break;
        // processModel
        mskf_imu_step st;
        const Vector3 gyro = m.angular_velocity - s.gyro_bias;
        const Vector3 acc = m.linear_acceleration - s.acc_bias;
        const double dtime = t - s.time;
        st.dt = dtime;
        const hm::Mat3 Rt = kin::rotation_of(s.orientation.q).transpose();
        for (int i = 0; i < 3; ++i) { st.gyro[i] = gyro[i]; st.acc[i] = acc[i]; }
        std::memcpy(st.R_t, Rt.m, sizeof(st.R_t));
        predictNewState(dtime, gyro, acc);
        const hm::Mat3 R_kk_1 = kin::rotation_of(s.orientation_null.q);
        const hm::Mat3 Phi00 = kin::rotation_of(s.orientation.q) * R_kk_1.transpose();
        std::memcpy(st.Phi00, Phi00.m, sizeof(st.Phi00));
        const Vector3 u = R_kk_1 * gravity_;
        const Vector3 sv = (1.0 / hm::dot(u, u)) * u;
        const Vector3 w1 = hm::skew(s.velocity_null - s.velocity) * gravity_;
        const Vector3 w2 = hm::skew(dtime * s.velocity_null + s.position_null - s.position) * gravity_;
        for (int i = 0; i < 3; ++i) { st.u[i] = u[i]; st.s[i] = sv[i]; st.w1[i] = w1[i]; st.w2[i] = w2[i]; }
        imu_steps_.push_back(st);
        s.orientation_null = s.orientation;
        s.position_null = s.position;
        s.velocity_null = s.velocity;
        s.time = t;
        ++used;
    }
    s.id = next_state_id_++;
    imu_msg_buffer.erase(imu_msg_buffer.begin(), imu_msg_buffer.begin() + used);
    if (!imu_steps_.empty() && !defer_device_) {
        int rc = mskf_ekf_propagate_imu(stream_, (int)imu_steps_.size(), imu_steps_.data());
        if (rc != MSKF_OK) fail("mskf_ekf_propagate_imu", rc);
    }
}

// :482-531
void MsckfVio::predictNewState(double dt, const Vector3 &gyro, const Vector3 &acc) {
    const double gn = hm::norm(gyro);
    double Om[4][4] = {{0}};
    const hm::Mat3 ms = -hm::skew(gyro);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Om[i][j] = ms(i, j); Om[i][3] = gyro[i]; Om[3][i] = -gyro[i]; }
    double *q = state_server.imu_state.orientation.q;
    Vector3 &v = state_server.imu_state.velocity;
    Vector3 &p = state_server.imu_state.position;
    auto apply = [&](double cI, double cO, double post, double *o) {
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int j = 0; j < 4; ++j) s += ((i == j ? cI : 0.0) + cO * Om[i][j]) * q[j];
            o[i] = s * post;
        }
    };
    double dq_dt[4], dq_dt2[4];
    if (gn > 1e-5) {
        apply(std::cos(gn * dt * 0.5), 1 / gn * std::sin(gn * dt * 0.5), 1.0, dq_dt);
        apply(std::cos(gn * dt * 0.25), 1 / gn * std::sin(gn * dt * 0.25), 1.0, dq_dt2);
    } else {
        apply(1.0, 0.5 * dt, std::cos(gn * dt * 0.5), dq_dt);
        apply(1.0, 0.25 * dt, std::cos(gn * dt * 0.25), dq_dt2);
    }
    const hm::Mat3 dR_dt_t = kin::rotation_of(dq_dt).transpose();
    const hm::Mat3 dR_dt2_t = kin::rotation_of(dq_dt2).transpose();
    const Vector3 &g = gravity_;
    const Vector3 k1_v_dot = kin::rotation_of(q).transpose() * acc + g;
    const Vector3 k1_p_dot = v;
    const Vector3 k1_v = v + k1_v_dot * dt / 2;
    const Vector3 k2_v_dot = dR_dt2_t * acc + g;
    const Vector3 k2_p_dot = k1_v;
    const Vector3 k2_v = v + k2_v_dot * dt / 2;
    const Vector3 k3_v_dot = dR_dt2_t * acc + g;
    const Vector3 k3_p_dot = k2_v;
    const Vector3 k3_v = v + k3_v_dot * dt;
    const Vector3 k4_v_dot = dR_dt_t * acc + g;
    const Vector3 k4_p_dot = k3_v;
    for (int i = 0; i < 4; ++i) q[i] = dq_dt[i];
    kin::normalize4(q);
    v = v + dt / 6 * (k1_v_dot + 2 * k2_v_dot + 2 * k3_v_dot + k4_v_dot);
    p = p + dt / 6 * (k1_p_dot + 2 * k2_p_dot + 2 * k3_p_dot + k4_p_dot);
}

// :533-585
void MsckfVio::stateAugmentation(double time) {
    const hm::Mat3 &R_i_c = state_server.imu_state.R_imu_cam0;
    const Vector3 &t_c_i = state_server.imu_state.t_cam0_imu;
    const hm::Mat3 R_w_i = kin::rotation_of(state_server.imu_state.orientation.q);
    const hm::Mat3 R_w_c = R_i_c * R_w_i;
    const Vector3 t_c_w = state_server.imu_state.position + R_w_i.transpose() * t_c_i;
    CAMState &cs = state_server.cam_states[state_server.imu_state.id];
    cs.id = state_server.imu_state.id;
    cs.time = time;
    kin::quaternion_of(R_w_c, cs.orientation.q);
    cs.position = t_c_w;
    cs.orientation_null = cs.orientation;
    cs.position_null = cs.position;
    cs.slot = allocCloneSlot();

    double *J = J_;
    have_J_ = true;
    std::memset(J, 0, sizeof(J_));
    const hm::Mat3 sk = hm::skew(R_w_i.transpose() * t_c_i);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { J[i * 21 + j] = R_i_c(i, j); J[(3 + i) * 21 + j] = sk(i, j); }
        J[i * 21 + 15 + i] = 1.0;
        J[(3 + i) * 21 + 12 + i] = 1.0;
        J[(3 + i) * 21 + 18 + i] = 1.0;
    }
    if (!defer_device_) {
        int rc = mskf_ekf_augment(stream_, J);
        if (rc != MSKF_OK) fail("mskf_ekf_augment", rc);
    }
}

// :587-608
// Q1: the message is never cleared, so behind the live entries it carries stale entries and then a
// growing tail of value-initialised records (id 0, zero coordinates).  Every tail record does the same
// thing (observation (0,0,0,0) for feature 0 at this state id), so when the producer told us where the
// tail starts (setZeroTailHint) it is collapsed: first record processed normally, the remaining T-1
// only bump tracked_feature_num exactly as the reference's loop would.  No hint -> plain loop.
// The loop itself is the reference's, in message order: a record either finds its feature (observation of the newest
// clone written, last write wins, tracked) or creates it.  The observation goes into the newest clone's column of the
// observation table, the feature's mask gets the newest clone's bit (feature_store.h).
void MsckfVio::addFeatureObservations(const CameraMeasurementConstPtr &msg) {
    const int curr_feature_num = (int)map_server.size();
    long long tracked = 0;
    size_t n_full = msg->features.size();
    size_t total = n_full;
    if (zero_tail_msg_ == msg.get()) {
        if (zero_tail_total_ > total) total = zero_tail_total_;          // truncated snapshot of a longer message
        if (zero_tail_start_ < total) n_full = std::min(n_full, zero_tail_start_ + 1);
    }
    const CAMState &newest = state_server.cam_states.rbegin()->second;   // the clone stateAugmentation just added
    const int cs = newest.slot;
    const uint64_t bit = 1ULL << (state_server.cam_states.size() - 1);
    for (size_t k = 0; k < n_full; ++k) {
        // two-stage prefetch: the hash bucket of the feature 12 ahead, then (bucket in cache) the rows of the feature 6 ahead
        if (k + 12 < n_full) map_server.prefetch_id((FeatureIDType)msg->features[k + 12].id);
        if (k + 6 < n_full) { const int s6 = map_server.find((FeatureIDType)msg->features[k + 6].id); if (s6 >= 0) map_server.prefetch_slot(cs, s6); }
        const FeatureMeasurement &f = msg->features[k];
        bool created = false;
        const int s = map_server.find_or_add((FeatureIDType)f.id, created);
        double *z = map_server.z(cs, s);
        z[0] = f.u0; z[1] = f.v0; z[2] = f.u1; z[3] = f.v1;
        map_server.mask(s) |= bit;
        if (!created) ++tracked;
    }
    tracked += (long long)(total - n_full);
    tracking_rate = static_cast<double>(tracked) / static_cast<double>(curr_feature_num);   // Q18: 0/0 = NaN on the first frame
}

// feature.hpp:257-287: first and last observation of the feature (bit positions of its mask = clone order)
bool MsckfVio::checkMotion(int slot, uint64_t mask) const {
    const int k0 = __builtin_ctzll(mask), k1 = 63 - __builtin_clzll(mask);
    const mskf_clone_state &c0 = clones_[k0], &c1 = clones_[k1];
    const hm::Mat3 R0 = kin::rotation_of(c0.q).transpose();
    const double *z = map_server.z(order_slot_[k0], slot);
    Vector3 dir(z[0], z[1], 1.0);
    dir = dir / hm::norm(dir);
    dir = R0 * dir;
    const Vector3 tr = Vector3(c1.p[0], c1.p[1], c1.p[2]) - Vector3(c0.p[0], c0.p[1], c0.p[2]);
    const double par = hm::dot(tr, dir);
    const Vector3 orth = tr - par * dir;
    return hm::norm(orth) > feat_translation_threshold_;
}

int MsckfVio::allocCloneSlot() {
    const int s = free_clone_slots_.back();
    free_clone_slots_.pop_back();
    return s;
}

void MsckfVio::resetCloneSlots() {
    // the window holds at most max_cam_state_size clones (pruning starts when it is full, :1079)
    const int rows = std::min((int)MapServer::kMaxClones, std::max(cfg_.max_cam_state_size, 4) + 1);
    map_server.set_clone_rows(rows);
    free_clone_slots_.clear();
    for (int s = rows - 1; s >= 0; --s) free_clone_slots_.push_back(s);
}

// clone states in window order (ascending state id) + the observation-table row of each
void MsckfVio::packClones() {
    clones_.clear(); order_slot_.clear();
    for (const auto &kv : state_server.cam_states) {
        mskf_clone_state c;
        for (int i = 0; i < 4; ++i) { c.q[i] = kv.second.orientation.q[i]; c.q_null[i] = kv.second.orientation_null.q[i]; }
        for (int i = 0; i < 3; ++i) { c.p[i] = kv.second.position[i]; c.p_null[i] = kv.second.position_null[i]; }
        clones_.push_back(c);
        order_slot_.push_back(kv.second.slot);
    }
}

void MsckfVio::finishArgs(mskf_ekf_update_args &upd, int dof_offset, int apply_cap) {
    std::memset(&upd, 0, sizeof(upd));
    upd.n_clones = (int)clones_.size();
    upd.n_feat = (int)feats_.size();
    upd.n_obs = (int)obs_clone_.size();
    upd.dof_offset = dof_offset;
    upd.apply_row_cap = apply_cap;
    for (int i = 0; i < 3; ++i) upd.gravity[i] = gravity_[i];
    delta_x_.assign(21 + 6 * clones_.size(), 0.0);
    gamma_.assign(feats_.size() + 1, 0.0);
    feat_status_.assign(feats_.size() + 1, 0);
    rows_out_ = 0;
    upd.clones = clones_.data();
    upd.features = feats_.data();
    upd.obs_clone = obs_clone_.data();
    upd.obs_z = obs_z_.data();
    upd.delta_x = delta_x_.data();
    upd.feat_status = feat_status_.data();
    upd.gamma = gamma_.data();
    upd.rows_out = &rows_out_;
    upd.diag_out = diag_out_;
    diag_out_[0] = diag_out_[1] = 0;
    upd.pos_var_out = pos_var_buf_;         // P(12..14, 12..14) diagonal after the update, for onlineReset (phaseD)
    pos_var_buf_[0] = pos_var_buf_[1] = pos_var_buf_[2] = -1.0;
}

// removeLostFeatures, selection part (:937-984).  Features are visited in ascending id (the reference's map order);
// the observations of a feature come out in ascending state id = ascending bit position of its mask, and the bit
// position IS the clone's index in the state vector.
void MsckfVio::buildLostFeatureUpdate(mskf_ekf_update_args &upd) {
    feats_.clear(); feat_slots_.clear(); erase_ranks_.clear(); obs_clone_.clear(); obs_z_.clear();
    packClones();
    const uint64_t cur_bit = 1ULL << (clones_.size() - 1);       // the clone of this frame
    const size_t nf = map_server.size();
    for (size_t rank = 0; rank < nf; ++rank) {
        map_server.prefetch_rank(rank + 24);
        map_server.prefetch_all_obs(rank + 8, cur_bit, order_slot_);
        const int slot = map_server.slot_at(rank);
        const uint64_t m = map_server.mask(slot);
        if (m & cur_bit) continue;                                // still tracked
        if (__builtin_popcountll(m) < 3) { erase_ranks_.push_back(rank); continue; }
        int needs_init = 0;
        if (!map_server.is_initialized(slot)) {
            if (!checkMotion(slot, m)) { erase_ranks_.push_back(rank); continue; }
            needs_init = 1;   // Feature::initializePosition runs on the device; an invalid result drops the feature there
        }
        mskf_ekf_feature f;
        std::memset(&f, 0, sizeof(f));
        f.obs_start = (int)obs_clone_.size();
        for (uint64_t b = m; b; b &= b - 1) {
            const int k = __builtin_ctzll(b);
            obs_clone_.push_back(k);
            const double *z = map_server.z(order_slot_[k], slot);
            obs_z_.insert(obs_z_.end(), z, z + 4);
        }
        f.n_obs = (int)obs_clone_.size() - f.obs_start;
        f.needs_init = needs_init; f.init_start = f.obs_start; f.n_init = f.n_obs;
        const Vector3 &p = map_server.position(slot);
        for (int i = 0; i < 3; ++i) f.position[i] = p[i];
        feats_.push_back(f);
        feat_slots_.push_back(slot);
        erase_ranks_.push_back(rank);      // processed features leave the map after the update (:1016-1021), in phaseB
    }
    finishArgs(upd, -1, 1);   // Q12 dof = #obs - 1, Q13 row cap
}

// measurementUpdate, state part (:862-894)
void MsckfVio::applyCorrection(const std::vector<double> &dx) {
    IMUState &s = state_server.imu_state;
    double dq[4];
    kin::small_angle(Vector3(dx[0], dx[1], dx[2]), dq);
    kin::multiply(dq, s.orientation.q, s.orientation.q);
    for (int i = 0; i < 3; ++i) {
        s.gyro_bias[i] += dx[3 + i];
        s.velocity[i] += dx[6 + i];
        s.acc_bias[i] += dx[9 + i];
        s.position[i] += dx[12 + i];
    }
    kin::small_angle(Vector3(dx[15], dx[16], dx[17]), dq);
    s.R_imu_cam0 = kin::rotation_of(dq) * s.R_imu_cam0;
    for (int i = 0; i < 3; ++i) s.t_cam0_imu[i] += dx[18 + i];
    n_tsqr_ += diag_out_[0] == 1 ? 1 : 0;
    n_direct_ += diag_out_[0] == 2 ? 1 : 0;
    rows_sum_ += rows_out_;
    int ci = 0;
    for (auto &kv : state_server.cam_states) {
        const double *d = &dx[21 + 6 * ci];
        kin::small_angle(Vector3(d[0], d[1], d[2]), dq);
        kin::multiply(dq, kv.second.orientation.q, kv.second.orientation.q);
        for (int i = 0; i < 3; ++i) kv.second.position[i] += d[3 + i];
        ++ci;
    }
    ++n_update_;
}

// featureJacobian's debug output (msckf_vio.cpp:719-723): in the frame with n_pub == 9 the reference writes the
// un-projected stacked Jacobians H_xj (4 M x d), H_fj (4 M x 3) and the residual r_j of every feature it linearises.
// The device keeps those only in LDS, so this (cold, file-output-only) path restates measurementJacobian (:610-677) on
// the host from the clone states and observations the update was built from and the position the device returned.
// Matrices are written row by row, blank separated.
void MsckfVio::dumpFeatureJacobians() {
    if (!debug_.is_open() || n_pub_ != 9) return;
    const int d = 21 + 6 * (int)clones_.size();
    for (size_t j = 0; j < feats_.size(); ++j) {
        if (!(feat_status_[j] & 1)) continue;      // featureJacobian is only reached with a valid position (:962-970, :1113-1116)
        const mskf_ekf_feature &f = feats_[j];
        const int M = f.n_obs;
        std::vector<double> Hx((size_t)4 * M * d, 0.0), Hf((size_t)4 * M * 3, 0.0), r((size_t)4 * M, 0.0);
        const Vector3 p_w(f.position[0], f.position[1], f.position[2]);
        for (int o = 0; o < M; ++o) {
            const int ci = obs_clone_[f.obs_start + o];
            const mskf_clone_state &cam = clones_[ci];
            const double *z = &obs_z_[(size_t)4 * (f.obs_start + o)];
            const hm::Mat3 R_w_c0 = kin::rotation_of(cam.q);
            const Vector3 t_c0_w(cam.p[0], cam.p[1], cam.p[2]);
            const hm::Mat3 R_c0_c1 = T_cam0_cam1_.R;
            const hm::Mat3 R_w_c1 = R_c0_c1 * R_w_c0;
            const Vector3 t_c1_w = t_c0_w - R_w_c1.transpose() * T_cam0_cam1_.t;
            const Vector3 p_c0 = R_w_c0 * (p_w - t_c0_w), p_c1 = R_w_c1 * (p_w - t_c1_w);
            double dz0[4][3] = {{0}}, dz1[4][3] = {{0}};
            dz0[0][0] = 1 / p_c0[2]; dz0[1][1] = 1 / p_c0[2]; dz0[0][2] = -p_c0[0] / (p_c0[2] * p_c0[2]); dz0[1][2] = -p_c0[1] / (p_c0[2] * p_c0[2]);
            dz1[2][0] = 1 / p_c1[2]; dz1[3][1] = 1 / p_c1[2]; dz1[2][2] = -p_c1[0] / (p_c1[2] * p_c1[2]); dz1[3][2] = -p_c1[1] / (p_c1[2] * p_c1[2]);
            const hm::Mat3 sk = hm::skew(p_c0), Rsk = R_c0_c1 * sk;
            double A[4][6];
            for (int i = 0; i < 4; ++i)
                for (int c = 0; c < 3; ++c) {
                    double a = 0, b = 0;
                    for (int k = 0; k < 3; ++k) { a += dz0[i][k] * sk(k, c) + dz1[i][k] * Rsk(k, c); b += -dz0[i][k] * R_w_c0(k, c) - dz1[i][k] * R_w_c1(k, c); }
                    A[i][c] = a; A[i][3 + c] = b;
                }
            const Vector3 g = gravity_;
            const Vector3 u0 = kin::rotation_of(cam.q_null) * g;
            const Vector3 u1 = hm::skew(p_w - Vector3(cam.p_null[0], cam.p_null[1], cam.p_null[2])) * g;
            const double u[6] = {u0[0], u0[1], u0[2], u1[0], u1[1], u1[2]};
            double uu = 0;
            for (int k = 0; k < 6; ++k) uu += u[k] * u[k];
            for (int i = 0; i < 4; ++i) {
                double Au = 0;
                for (int k = 0; k < 6; ++k) Au += A[i][k] * u[k];
                for (int c = 0; c < 6; ++c) {
                    const double h = A[i][c] - Au * (1.0 / uu) * u[c];
                    Hx[(size_t)(4 * o + i) * d + 21 + 6 * ci + c] = h;
                    if (c >= 3) Hf[(size_t)(4 * o + i) * 3 + (c - 3)] = -h;
                }
            }
            r[4 * o + 0] = z[0] - p_c0[0] / p_c0[2]; r[4 * o + 1] = z[1] - p_c0[1] / p_c0[2];
            r[4 * o + 2] = z[2] - p_c1[0] / p_c1[2]; r[4 * o + 3] = z[3] - p_c1[1] / p_c1[2];
        }
        auto mat = [&](const char *name, const std::vector<double> &m, int cols) {
            debug_ << name << "\n";
            for (size_t i = 0; i < m.size(); ++i) debug_ << m[i] << (((int)(i % cols) == cols - 1) ? "\n" : " ");
            debug_ << std::endl;
        };
        mat("featureJacobian H_xj:", Hx, d);
        mat("featureJacobian H_fj:", Hf, 3);
        mat("featureJacobian r_j:", r, 1);
    }
}

void MsckfVio::phaseB(mskf_ekf_update_args &upd) {
    if (!feats_.empty()) takePosVar();           // what the lost-feature update brought back
    dumpFeatureJacobians();                      // (no-op unless file outputs are on and this is frame 9)
    // tail of removeLostFeatures (:1016-1021)
    if (!feats_.empty() && rows_out_ > 0) { hostprof::Scope hp(hostprof::EKF_APPLY1); applyCorrection(delta_x_); }
    {
        // the features that were too short-lived or failed the motion check (:944-960) and the ones just used
        hostprof::Scope hp(hostprof::EKF_ERASE_LOST);
        map_server.erase_ranks(erase_ranks_);
        erase_ranks_.clear();
    }
    hostprof::Scope hp(hostprof::EKF_BUILD_PRUNE);
    buildPruneUpdate(upd);
}

// :1026-1071
void MsckfVio::findRedundantCamStates(std::vector<StateIDType> &rm) {
    auto key_it = state_server.cam_states.end();
    for (int i = 0; i < 4; ++i) --key_it;
    auto cam_it = key_it; ++cam_it;
    auto first_it = state_server.cam_states.begin();
    const Vector3 key_position = key_it->second.position;
    const hm::Mat3 key_rotation = kin::rotation_of(key_it->second.orientation.q);
    for (int i = 0; i < 2; ++i) {
        const Vector3 position = cam_it->second.position;
        const hm::Mat3 rotation = kin::rotation_of(cam_it->second.orientation.q);
        const double distance = hm::norm(position - key_position);
        const double angle = kin::angle_of(rotation * key_rotation.transpose());
        if (angle < cfg_.rotation_threshold && distance < cfg_.translation_threshold && tracking_rate > cfg_.tracking_rate_threshold) {
            rm.push_back(cam_it->first);
            ++cam_it;
        } else {
            rm.push_back(first_it->first);
            ++first_it;
        }
    }
    std::sort(rm.begin(), rm.end());
}

// pruneCamStateBuffer, selection part (:1073-1153)
void MsckfVio::buildPruneUpdate(mskf_ekf_update_args &upd) {
    feats_.clear(); feat_slots_.clear(); obs_clone_.clear(); obs_z_.clear();
    rm_cam_state_ids_.clear();
    prune_pending_ = false;
    std::memset(&upd, 0, sizeof(upd));
    if ((int)state_server.cam_states.size() < cfg_.max_cam_state_size) return;
    prune_pending_ = true;
    findRedundantCamStates(rm_cam_state_ids_);
    packClones();
    // order indices (= mask bit positions) and table rows of the two clones being removed; rm ids are sorted
    int ka = -1, kb = -1;
    { int k = 0; for (const auto &kv : state_server.cam_states) { if (kv.first == rm_cam_state_ids_[0]) ka = k; if (kv.first == rm_cam_state_ids_[1]) kb = k; ++k; } }
    rm_order_[0] = ka; rm_order_[1] = kb;
    const uint64_t bit_a = 1ULL << ka, bit_b = 1ULL << kb;
    const int row_a = order_slot_[ka], row_b = order_slot_[kb];
    const size_t nf = map_server.size();
    for (size_t rank = 0; rank < nf; ++rank) {
        map_server.prefetch_rank(rank + 24);
        map_server.prefetch_obs(rank + 8, row_a, row_b);
        const int slot = map_server.slot_at(rank);
        const uint64_t m = map_server.mask(slot);
        // a feature observed by only one of the two loses that observation (:1096-1101), one observed by both but
        // without enough motion loses both (:1104-1112): both happen when the clone bits are removed in phaseC
        if ((m & bit_a) == 0 || (m & bit_b) == 0) continue;
        int needs_init = 0;
        if (!map_server.is_initialized(slot)) {
            if (!checkMotion(slot, m)) continue;
            needs_init = 1;
        }
        mskf_ekf_feature f;
        std::memset(&f, 0, sizeof(f));
        if (needs_init) {   // triangulation uses ALL observations of the feature (feature.hpp:298-320)
            f.init_start = (int)obs_clone_.size();
            for (uint64_t b = m; b; b &= b - 1) {
                const int k = __builtin_ctzll(b);
                obs_clone_.push_back(k);
                const double *z = map_server.z(order_slot_[k], slot);
                obs_z_.insert(obs_z_.end(), z, z + 4);
            }
            f.n_init = (int)obs_clone_.size() - f.init_start;
        }
        f.obs_start = (int)obs_clone_.size();
        {   // the Jacobian block uses only the clones being removed (:1143)
            obs_clone_.push_back(ka);
            const double *za = map_server.z(row_a, slot);
            obs_z_.insert(obs_z_.end(), za, za + 4);
            obs_clone_.push_back(kb);
            const double *zb = map_server.z(row_b, slot);
            obs_z_.insert(obs_z_.end(), zb, zb + 4);
        }
        f.n_obs = 2;
        f.needs_init = needs_init;
        const Vector3 &p = map_server.position(slot);
        for (int i = 0; i < 3; ++i) f.position[i] = p[i];
        feats_.push_back(f);
        feat_slots_.push_back(slot);
    }
    finishArgs(upd, 0, 0);   // Q12 dof = #involved, no row cap in the pruning path
}

void MsckfVio::phaseC(bool defer_device) {
    pending_rm_[0] = pending_rm_[1] = -1;
    if (!frame_active_) return;
    if (prune_pending_) {
        if (!feats_.empty()) takePosVar();       // the pruning update ran after the lost-feature update: its values are the current ones
        dumpFeatureJacobians();
        hostprof::Scope hp(hostprof::EKF_TAIL_PRUNE);
        // tail of pruneCamStateBuffer (:1100-1181)
        for (size_t j = 0; j < feats_.size(); ++j)
            if (feats_[j].needs_init && (feat_status_[j] & 1))
                map_server.set_position(feat_slots_[j], Vector3(feats_[j].position[0], feats_[j].position[1], feats_[j].position[2]));
        if (!feats_.empty() && rows_out_ > 0) applyCorrection(delta_x_);
        // every observation of the two clones goes (:1096-1101, :1104-1112, :1157-1159): drop their bit positions from
        // all masks, higher position first so that the lower one stays valid
        map_server.remove_clone_bit(rm_order_[1]);
        map_server.remove_clone_bit(rm_order_[0]);
        // indices of both clones in the state order BEFORE either is erased (:1161-1181 removes them one by one)
        pending_rm_[0] = rm_order_[0]; pending_rm_[1] = rm_order_[1];
        for (const auto &cid : rm_cam_state_ids_) {
            auto it = state_server.cam_states.find(cid);
            free_clone_slots_.push_back(it->second.slot);
            state_server.cam_states.erase(it);
        }
        if (!defer_device) {
            mskf_stream *ss[1] = {stream_};
            int rc = mskf_ekf_remove_clones_batch(mskf_stream_ekf_ctx(stream_), 1, ss, pending_rm_);
            if (rc != MSKF_OK) fail("mskf_ekf_remove_clones_batch", rc);
            pending_rm_[0] = pending_rm_[1] = -1;
        }
        prune_pending_ = false;
    }
    hostprof::Scope hp_pub(hostprof::EKF_PUBLISH);
    publish(frame_time_);
    ++n_pub_;                                    // :356
}

// :1186-1236
void MsckfVio::phaseD(const double pos_var[3]) {
    if (!frame_active_ || cfg_.position_std_threshold <= 0) return;
    const double sx = std::sqrt(pos_var[0]), sy = std::sqrt(pos_var[1]), sz = std::sqrt(pos_var[2]);
    if (sx < cfg_.position_std_threshold && sy < cfg_.position_std_threshold && sz < cfg_.position_std_threshold) return;
    ++online_reset_counter_;
    state_server.cam_states.clear();
    resetCloneSlots();
    map_server.clear();
    resetCov();
}

// :1238-1305
void MsckfVio::publish(double time_stamp) {
    const IMUState &s = state_server.imu_state;
    const hm::Rigid T_i_w(kin::rotation_of(s.orientation.q).transpose(), s.position);
    const hm::Rigid T_b_w = T_imu_body_ * T_i_w * T_imu_body_.inverse();
    mskf_pose pose;
    pose.time_stamp = time_stamp;
    for (int i = 0; i < 3; ++i) pose.p[i] = T_b_w.t[i];
    kin::hamilton_of(T_b_w.R, pose.q);
    if (keepTrajectory) {
        poses_.push_back(pose);
        path_.push_back(T_b_w.t);
        for (size_t rank = 0; rank < map_server.size(); ++rank) {
            const int slot = map_server.slot_at(rank);
            if (!map_server.is_initialized(slot)) continue;
            const Vector3 fp = T_imu_body_.R * map_server.position(slot);
            points3d_.push_back(Point3f((float)fp[0], (float)fp[1], (float)fp[2]));
        }
    } else {
        if (poses_.empty()) poses_.push_back(pose); else poses_[0] = pose;
    }
    if (pose_outfile_.is_open()) {
        pose_outfile_ << std::fixed << time_stamp << " " << pose.p[0] << " " << pose.p[1] << " " << pose.p[2] << " " << pose.q[0] << " "
                      << pose.q[1] << " " << pose.q[2] << " " << pose.q[3] << std::endl;   // TUM format, Q15
    }
}

}  // namespace cg
