// oracle/o_filter.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).  See o_filter.h.
#include "o_filter.h"
#include <algorithm>
#include <array>
#include <cstdio>
#include "../include/mskf_chi2_table.h"

namespace orc {

static inline V3 v3(const double *a) { return V3(a[0], a[1], a[2]); }

// ---------------------------------------------------------------- Feature (feature.hpp)
// feature.hpp:171-190
static void feat_cost(const SE3 &T_c0_ci, const V3 &x, const double z[2], double &e) {
    const double alpha = x[0], beta = x[1], rho = x[2];
    V3 h = T_c0_ci.R * V3(alpha, beta, 1.0) + rho * T_c0_ci.t;
    const double zh0 = h[0] / h[2], zh1 = h[1] / h[2];
    e = (zh0 - z[0]) * (zh0 - z[0]) + (zh1 - z[1]) * (zh1 - z[1]);
}
// feature.hpp:192-229
static void feat_jacobian(const SE3 &T_c0_ci, const V3 &x, const double z[2], double J[2][3], double r[2], double &w,
                          const FilterShared &sh) {
    const double alpha = x[0], beta = x[1], rho = x[2];
    V3 h = T_c0_ci.R * V3(alpha, beta, 1.0) + rho * T_c0_ci.t;
    const double h1 = h[0], h2 = h[1], h3 = h[2];
    double W[3][3];
    for (int i = 0; i < 3; ++i) { W[i][0] = T_c0_ci.R(i, 0); W[i][1] = T_c0_ci.R(i, 1); W[i][2] = T_c0_ci.t[i]; }
    for (int j = 0; j < 3; ++j) {
        J[0][j] = 1 / h3 * W[0][j] - h1 / (h3 * h3) * W[2][j];
        J[1][j] = 1 / h3 * W[1][j] - h2 / (h3 * h3) * W[2][j];
    }
    r[0] = h1 / h3 - z[0];
    r[1] = h2 / h3 - z[1];
    const double e = std::sqrt(r[0] * r[0] + r[1] * r[1]);
    if (e <= sh.huber_epsilon) w = 1.0;
    else w = std::sqrt(2.0 * sh.huber_epsilon / e);
}
// feature.hpp:231-255
static void feat_initial_guess(const SE3 &T_c1_c2, const double z1[2], const double z2[2], V3 &p) {
    V3 m = T_c1_c2.R * V3(z1[0], z1[1], 1.0);
    double A[2] = {m[0] - z2[0] * m[2], m[1] - z2[1] * m[2]};
    double b[2] = {z2[0] * T_c1_c2.t[2] - T_c1_c2.t[0], z2[1] * T_c1_c2.t[2] - T_c1_c2.t[1]};
    const double depth = (A[0] * b[0] + A[1] * b[1]) / (A[0] * A[0] + A[1] * A[1]);
    p = V3(z1[0] * depth, z1[1] * depth, depth);
}

bool feature_check_motion(const Feature &f, const CamStateServer &cams, const FilterShared &sh) {
    const StateIDType first_id = f.observations.begin()->first;
    const StateIDType last_id = (--f.observations.end())->first;
    const CAMState &c0 = cams.find(first_id)->second;
    const CAMState &c1 = cams.find(last_id)->second;
    M3 R0 = quat_to_rot(c0.orientation).t();
    const auto &z = f.observations.begin()->second;
    V3 dir(z[0], z[1], 1.0);
    dir = dir / norm(dir);
    dir = R0 * dir;
    V3 tr = c1.position - c0.position;
    const double par = dot(tr, dir);
    V3 orth = tr - par * dir;
    return norm(orth) > sh.feat_translation_threshold;
}

// 3x3 symmetric solve (the reference uses Eigen ldlt(), feature.hpp:388-395)
static bool solve3(const double A[3][3], const double b[3], double x[3]) {
    Mat S(3, 3), B(3, 1);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) S(i, j) = A[i][j]; B(i, 0) = b[i]; }
    if (!chol_solve(S, B)) return false;
    for (int i = 0; i < 3; ++i) x[i] = B(i, 0);
    return true;
}

bool feature_initialize_position(Feature &f, const CamStateServer &cams, const FilterShared &sh) {
    std::vector<SE3> cam_poses;
    std::vector<std::array<double, 2>> meas;
    for (const auto &m : f.observations) {
        auto it = cams.find(m.first);
        if (it == cams.end()) continue;
        meas.push_back({m.second[0], m.second[1]});
        meas.push_back({m.second[2], m.second[3]});
        SE3 cam0_pose(quat_to_rot(it->second.orientation).t(), it->second.position);
        SE3 cam1_pose = cam0_pose * sh.T_cam0_cam1.inv();
        cam_poses.push_back(cam0_pose);
        cam_poses.push_back(cam1_pose);
    }
    SE3 T_c0_w = cam_poses[0];
    for (auto &pose : cam_poses) pose = pose.inv() * T_c0_w;

    V3 init;
    feat_initial_guess(cam_poses[cam_poses.size() - 1], meas[0].data(), meas[meas.size() - 1].data(), init);
    V3 solution(init[0] / init[2], init[1] / init[2], 1.0 / init[2]);

    double lambda = sh.initial_damping;
    int inner = 0, outer = 0;
    bool is_cost_reduced = false;
    double delta_norm = 0;
    double total_cost = 0.0;
    for (size_t i = 0; i < cam_poses.size(); ++i) { double c; feat_cost(cam_poses[i], solution, meas[i].data(), c); total_cost += c; }
    do {
        double A[3][3] = {{0}}, b[3] = {0, 0, 0};
        for (size_t i = 0; i < cam_poses.size(); ++i) {
            double J[2][3], r[2], w;
            feat_jacobian(cam_poses[i], solution, meas[i].data(), J, r, w, sh);
            const double ws = (w == 1) ? 1.0 : w * w;
            for (int a = 0; a < 3; ++a) {
                for (int c = 0; c < 3; ++c) A[a][c] += ws * (J[0][a] * J[0][c] + J[1][a] * J[1][c]);
                b[a] += ws * (J[0][a] * r[0] + J[1][a] * r[1]);
            }
        }
        do {
            double At[3][3];
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) At[a][c] = A[a][c] + (a == c ? lambda : 0.0);
            double delta[3] = {0, 0, 0};
            solve3(At, b, delta);
            V3 new_solution(solution[0] - delta[0], solution[1] - delta[1], solution[2] - delta[2]);
            delta_norm = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
            double new_cost = 0.0;
            for (size_t i = 0; i < cam_poses.size(); ++i) { double c; feat_cost(cam_poses[i], new_solution, meas[i].data(), c); new_cost += c; }
            if (new_cost < total_cost) {
                is_cost_reduced = true;
                solution = new_solution;
                total_cost = new_cost;
                lambda = lambda / 10 > 1e-10 ? lambda / 10 : 1e-10;
            } else {
                is_cost_reduced = false;
                lambda = lambda * 10 < 1e12 ? lambda * 10 : 1e12;
            }
        } while (inner++ < sh.inner_loop_max_iteration && !is_cost_reduced);
        inner = 0;
    } while (outer++ < sh.outer_loop_max_iteration && delta_norm > sh.estimation_precision);

    V3 final_position(solution[0] / solution[2], solution[1] / solution[2], 1.0 / solution[2]);
    bool valid = true;
    for (const auto &pose : cam_poses) {
        V3 p = pose.R * final_position + pose.t;
        if (p[2] <= 0) { valid = false; break; }
    }
    f.position = T_c0_w.R * final_position + T_c0_w.t;
    if (valid) f.is_initialized = true;
    return valid;
}

// ---------------------------------------------------------------- MsckfVio
// msckf_vio.cpp:51-188 (ctor, loadParameters, initialize)
MsckfVio::MsckfVio(const mskf_calib &calib, const mskf_ekf_cfg &cfg) : calib_(calib), cfg_(cfg) {
    sh.gyro_noise = cfg.noise_gyro * cfg.noise_gyro;
    sh.acc_noise = cfg.noise_acc * cfg.noise_acc;
    sh.gyro_bias_noise = cfg.noise_gyro_bias * cfg.noise_gyro_bias;
    sh.acc_bias_noise = cfg.noise_acc_bias * cfg.noise_acc_bias;
    sh.observation_noise = cfg.noise_feature * cfg.noise_feature;
    sh.feat_translation_threshold = cfg.feature_translation_threshold;
    state_server.imu_state.velocity = v3(cfg.init_velocity);
    resetCov();
    SE3 T_cam0_imu = SE3::from16(calib.T_cam0_imu).inv();  // :115-119
    state_server.imu_state.R_imu_cam0 = T_cam0_imu.R.t();
    state_server.imu_state.t_cam0_imu = T_cam0_imu.t;
    sh.T_cam0_cam1 = SE3::from16(calib.T_cam1_cam0);       // :121-122
    sh.T_imu_body = SE3::from16(calib.T_imu_body).inv();   // :124-125
    state_server.continuous_noise_cov = Mat(12, 12);       // :174-178
    for (int i = 0; i < 3; ++i) {
        state_server.continuous_noise_cov(i, i) = sh.gyro_noise;
        state_server.continuous_noise_cov(3 + i, 3 + i) = sh.gyro_bias_noise;
        state_server.continuous_noise_cov(6 + i, 6 + i) = sh.acc_noise;
        state_server.continuous_noise_cov(9 + i, 9 + i) = sh.acc_bias_noise;
    }
    chi2_table[0] = 0;
    for (int i = 1; i < 100; ++i) chi2_table[i] = cfg.chi2_mode == 1 ? mskf_chi2_ppf95[i - 1] : mskf_chi2_ppf05[i - 1];  // :181-185, Q11
}

void MsckfVio::resetCov() {  // :102-112
    state_server.state_cov = Mat(21, 21);
    for (int i = 3; i < 6; ++i) state_server.state_cov(i, i) = cfg_.cov_gyro_bias;
    for (int i = 6; i < 9; ++i) state_server.state_cov(i, i) = cfg_.cov_velocity;
    for (int i = 9; i < 12; ++i) state_server.state_cov(i, i) = cfg_.cov_acc_bias;
    for (int i = 15; i < 18; ++i) state_server.state_cov(i, i) = cfg_.cov_ext_rot;
    for (int i = 18; i < 21; ++i) state_server.state_cov(i, i) = cfg_.cov_ext_trans;
}

void MsckfVio::imuCallback(const mskf_imu_sample &msg) {
    imu_msg_buffer.push_back(msg);
    if (!is_gravity_set) {
        if (imu_msg_buffer.size() < 200) return;
        initializeGravityAndBias();
        is_gravity_set = true;
    }
}

// :209-241
void MsckfVio::initializeGravityAndBias() {
    V3 sum_w, sum_a;
    for (const auto &m : imu_msg_buffer) { sum_w = sum_w + v3(m.angular_velocity); sum_a = sum_a + v3(m.linear_acceleration); }
    state_server.imu_state.gyro_bias = sum_w / (double)imu_msg_buffer.size();
    V3 gravity_imu = sum_a / (double)imu_msg_buffer.size();
    const double gn = norm(gravity_imu);
    sh.gravity = V3(0.0, 0.0, -gn);
    state_server.imu_state.orientation = rot_to_quat(from_two_vectors(gravity_imu, -sh.gravity).t());
}

bool MsckfVio::resetCallback() {  // :243-304
    IMUState &s = state_server.imu_state;
    s.time = 0.0;
    s.orientation = Quat(); s.position = V3(); s.velocity = V3(); s.gyro_bias = V3(); s.acc_bias = V3();
    s.orientation_null = Quat(); s.position_null = V3(); s.velocity_null = V3();
    state_server.cam_states.clear();
    resetCov();
    map_server.clear();
    imu_msg_buffer.clear();
    is_gravity_set = false;
    is_first_img = true;
    return true;
}

void MsckfVio::featureCallback(const CameraMeasurement &msg) {
    if (!is_gravity_set) return;
    if (is_first_img) { is_first_img = false; state_server.imu_state.time = msg.time_stamp; }
    batchImuProcessing(msg.time_stamp);
    stateAugmentation(msg.time_stamp);
    addFeatureObservations(msg);
    removeLostFeatures();
    pruneCamStateBuffer();
    publish(msg.time_stamp);
    onlineReset();
}

// :377-407
void MsckfVio::batchImuProcessing(double time_bound) {
    int used = 0;
    for (const auto &m : imu_msg_buffer) {
        const double t = m.time_stamp;
        if (t < state_server.imu_state.time) { ++used; continue; }
        if (t > time_bound) break;
        processModel(t, v3(m.angular_velocity), v3(m.linear_acceleration));
        ++used;
    }
    state_server.imu_state.id = next_state_id++;
    imu_msg_buffer.erase(imu_msg_buffer.begin(), imu_msg_buffer.begin() + used);
}

// :409-480
void MsckfVio::processModel(double time, const V3 &m_gyro, const V3 &m_acc) {
    IMUState &s = state_server.imu_state;
    V3 gyro = m_gyro - s.gyro_bias;
    V3 acc = m_acc - s.acc_bias;
    const double dtime = time - s.time;

    Mat F(21, 21), G(21, 12);
    const M3 Rt = quat_to_rot(s.orientation).t();
    F.set(0, 0, -skew(gyro));
    F.set(0, 3, -M3::eye());
    F.set(6, 0, -(Rt * skew(acc)));
    F.set(6, 9, -Rt);
    F.set(12, 6, M3::eye());
    G.set(0, 0, -M3::eye());
    G.set(3, 3, M3::eye());
    G.set(6, 6, -Rt);
    G.set(9, 9, M3::eye());

    Mat Fdt = dtime * F;
    Mat Fdt2 = Fdt * Fdt;
    Mat Fdt3 = Fdt2 * Fdt;
    Mat Phi = Mat::eye(21) + Fdt + 0.5 * Fdt2 + (1.0 / 6.0) * Fdt3;

    predictNewState(dtime, gyro, acc);

    M3 R_kk_1 = quat_to_rot(s.orientation_null);
    Phi.set(0, 0, quat_to_rot(s.orientation) * R_kk_1.t());
    V3 u = R_kk_1 * sh.gravity;
    V3 sv = (1.0 / dot(u, u)) * u;
    auto fix = [&](int r0, const V3 &w) {
        M3 A;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A(i, j) = Phi(r0 + i, j);
        V3 Au_w = A * u - w;
        M3 out;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out(i, j) = A(i, j) - Au_w[i] * sv[j];
        Phi.set(r0, 0, out);
    };
    fix(6, skew(s.velocity_null - s.velocity) * sh.gravity);
    fix(12, skew(dtime * s.velocity_null + s.position_null - s.position) * sh.gravity);

    Mat Q = dtime * (Phi * G * state_server.continuous_noise_cov * G.t() * Phi.t());
    Mat &P = state_server.state_cov;
    Mat PII = Phi * P.block(0, 0, 21, 21) * Phi.t() + Q;
    if (!state_server.cam_states.empty()) {
        Mat PIC = Phi * P.block(0, 21, 21, P.c - 21);
        Mat PCI = P.block(21, 0, P.r - 21, 21) * Phi.t();
        P.set(0, 21, PIC);
        P.set(21, 0, PCI);
    }
    P.set(0, 0, PII);
    Mat Pt = P.t();
    for (size_t i = 0; i < P.d.size(); ++i) P.d[i] = (P.d[i] + Pt.d[i]) * 0.5;

    s.orientation_null = s.orientation;
    s.position_null = s.position;
    s.velocity_null = s.velocity;
    s.time = time;
}

// :482-531
void MsckfVio::predictNewState(double dt, const V3 &gyro, const V3 &acc) {
    const double gn = norm(gyro);
    double Om[4][4] = {{0}};
    M3 ms = -skew(gyro);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Om[i][j] = ms(i, j); Om[i][3] = gyro[i]; Om[3][i] = -gyro[i]; }
    Quat &q = state_server.imu_state.orientation;
    V3 &v = state_server.imu_state.velocity;
    V3 &p = state_server.imu_state.position;
    auto apply = [&](double cI, double cO, double post) {
        Quat o;
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int j = 0; j < 4; ++j) s += ((i == j ? cI : 0.0) + cO * Om[i][j]) * q[j];
            o[i] = s * post;
        }
        return o;
    };
    Quat dq_dt, dq_dt2;
    if (gn > 1e-5) {
        dq_dt = apply(std::cos(gn * dt * 0.5), 1 / gn * std::sin(gn * dt * 0.5), 1.0);
        dq_dt2 = apply(std::cos(gn * dt * 0.25), 1 / gn * std::sin(gn * dt * 0.25), 1.0);
    } else {
        dq_dt = apply(1.0, 0.5 * dt, std::cos(gn * dt * 0.5));
        dq_dt2 = apply(1.0, 0.25 * dt, std::cos(gn * dt * 0.25));
    }
    M3 dR_dt_t = quat_to_rot(dq_dt).t();
    M3 dR_dt2_t = quat_to_rot(dq_dt2).t();
    const V3 &g = sh.gravity;
    V3 k1_v_dot = quat_to_rot(q).t() * acc + g;
    V3 k1_p_dot = v;
    V3 k1_v = v + k1_v_dot * dt / 2;
    V3 k2_v_dot = dR_dt2_t * acc + g;
    V3 k2_p_dot = k1_v;
    V3 k2_v = v + k2_v_dot * dt / 2;
    V3 k3_v_dot = dR_dt2_t * acc + g;
    V3 k3_p_dot = k2_v;
    V3 k3_v = v + k3_v_dot * dt;
    V3 k4_v_dot = dR_dt_t * acc + g;
    V3 k4_p_dot = k3_v;
    q = qnormalized(dq_dt);
    v = v + dt / 6 * (k1_v_dot + 2 * k2_v_dot + 2 * k3_v_dot + k4_v_dot);
    p = p + dt / 6 * (k1_p_dot + 2 * k2_p_dot + 2 * k3_p_dot + k4_p_dot);
}

// :533-585
void MsckfVio::stateAugmentation(double time) {
    const M3 &R_i_c = state_server.imu_state.R_imu_cam0;
    const V3 &t_c_i = state_server.imu_state.t_cam0_imu;
    M3 R_w_i = quat_to_rot(state_server.imu_state.orientation);
    M3 R_w_c = R_i_c * R_w_i;
    V3 t_c_w = state_server.imu_state.position + R_w_i.t() * t_c_i;
    CAMState &cs = state_server.cam_states[state_server.imu_state.id];
    cs.id = state_server.imu_state.id;
    cs.time = time;
    cs.orientation = rot_to_quat(R_w_c);
    cs.position = t_c_w;
    cs.orientation_null = cs.orientation;
    cs.position_null = cs.position;

    Mat J(6, 21);
    J.set(0, 0, R_i_c);
    J.set(0, 15, M3::eye());
    J.set(3, 0, skew(R_w_i.t() * t_c_i));
    J.set(3, 12, M3::eye());
    J.set(3, 18, M3::eye());
    Mat &P = state_server.state_cov;
    const int old = P.r;
    Mat P11 = P.block(0, 0, 21, 21);
    Mat P12 = P.block(0, 21, 21, old - 21);
    P.conservative_resize(old + 6, old + 6);
    Mat JP11 = J * P11;
    P.set(old, 0, JP11);
    if (old > 21) P.set(old, 21, J * P12);
    P.set(0, old, P.block(old, 0, 6, old).t());
    P.set(old, old, JP11 * J.t());
    Mat Pt = P.t();
    for (size_t i = 0; i < P.d.size(); ++i) P.d[i] = (P.d[i] + Pt.d[i]) / 2.0;
}

// :587-608
void MsckfVio::addFeatureObservations(const CameraMeasurement &msg) {
    StateIDType state_id = state_server.imu_state.id;
    int curr_feature_num = (int)map_server.size();
    int tracked = 0;
    for (const auto &f : msg.features) {
        FeatureIDType fid = (FeatureIDType)f.id;
        if (map_server.find(fid) == map_server.end()) {
            map_server[fid].id = fid;
            map_server[fid].observations[state_id] = {f.u0, f.v0, f.u1, f.v1};
        } else {
            map_server[fid].observations[state_id] = {f.u0, f.v0, f.u1, f.v1};
            ++tracked;
        }
    }
    tracking_rate = static_cast<double>(tracked) / static_cast<double>(curr_feature_num);  // Q18: 0/0 = NaN on the first frame
}

// :610-677
void MsckfVio::measurementJacobian(StateIDType cam_state_id, FeatureIDType feature_id, Mat &H_x, Mat &H_f, double r[4]) {
    const CAMState &cam = state_server.cam_states[cam_state_id];
    const Feature &feature = map_server[feature_id];
    M3 R_w_c0 = quat_to_rot(cam.orientation);
    const V3 &t_c0_w = cam.position;
    M3 R_c0_c1 = sh.T_cam0_cam1.R;
    M3 R_w_c1 = sh.T_cam0_cam1.R * R_w_c0;
    V3 t_c1_w = t_c0_w - R_w_c1.t() * sh.T_cam0_cam1.t;
    const V3 &p_w = feature.position;
    const auto &z = feature.observations.find(cam_state_id)->second;
    V3 p_c0 = R_w_c0 * (p_w - t_c0_w);
    V3 p_c1 = R_w_c1 * (p_w - t_c1_w);

    Mat dz_dpc0(4, 3), dz_dpc1(4, 3);
    dz_dpc0(0, 0) = 1 / p_c0[2];
    dz_dpc0(1, 1) = 1 / p_c0[2];
    dz_dpc0(0, 2) = -p_c0[0] / (p_c0[2] * p_c0[2]);
    dz_dpc0(1, 2) = -p_c0[1] / (p_c0[2] * p_c0[2]);
    dz_dpc1(2, 0) = 1 / p_c1[2];
    dz_dpc1(3, 1) = 1 / p_c1[2];
    dz_dpc1(2, 2) = -p_c1[0] / (p_c1[2] * p_c1[2]);
    dz_dpc1(3, 2) = -p_c1[1] / (p_c1[2] * p_c1[2]);
    Mat dpc0_dxc(3, 6), dpc1_dxc(3, 6);
    dpc0_dxc.set(0, 0, skew(p_c0));
    dpc0_dxc.set(0, 3, -R_w_c0);
    dpc1_dxc.set(0, 0, R_c0_c1 * skew(p_c0));
    dpc1_dxc.set(0, 3, -R_w_c1);
    H_x = dz_dpc0 * dpc0_dxc + dz_dpc1 * dpc1_dxc;
    H_f = dz_dpc0 * to_mat(R_w_c0) + dz_dpc1 * to_mat(R_w_c1);

    // observability constraint (:666-671)
    Mat A = H_x;
    V3 u0 = quat_to_rot(cam.orientation_null) * sh.gravity;
    V3 u1 = skew(p_w - cam.position_null) * sh.gravity;
    double u[6] = {u0[0], u0[1], u0[2], u1[0], u1[1], u1[2]};
    double uu = 0;
    for (int i = 0; i < 6; ++i) uu += u[i] * u[i];
    for (int i = 0; i < 4; ++i) {
        double Au = 0;
        for (int k = 0; k < 6; ++k) Au += A(i, k) * u[k];
        for (int j = 0; j < 6; ++j) H_x(i, j) = A(i, j) - Au * (1.0 / uu) * u[j];
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) H_f(i, j) = -H_x(i, 3 + j);
    r[0] = z[0] - p_c0[0] / p_c0[2];
    r[1] = z[1] - p_c0[1] / p_c0[2];
    r[2] = z[2] - p_c1[0] / p_c1[2];
    r[3] = z[3] - p_c1[1] / p_c1[2];
}

// Householder QR applied in place: reduces the first `ncol` columns of M (m x n, n >= ncol) and
// applies the same reflectors to all remaining columns.  Rows [ncol, m) of the trailing columns
// are then Q2^T * (trailing), the projection on the left null space of M[:, :ncol].
static void householder_reduce(Mat &M, int ncol) {
    const int m = M.r, n = M.c;
    for (int k = 0; k < ncol && k < m - 1; ++k) {
        double nrm = 0;
        for (int i = k; i < m; ++i) nrm += M(i, k) * M(i, k);
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = M(k, k) > 0 ? -nrm : nrm;
        std::vector<double> v(m - k);
        for (int i = k; i < m; ++i) v[i - k] = M(i, k);
        v[0] -= alpha;
        double vn = 0;
        for (double x : v) vn += x * x;
        if (vn == 0.0) continue;
        for (int j = k; j < n; ++j) {
            double s = 0;
            for (int i = k; i < m; ++i) s += v[i - k] * M(i, j);
            s = 2.0 * s / vn;
            for (int i = k; i < m; ++i) M(i, j) -= s * v[i - k];
        }
    }
}

// :679-775
void MsckfVio::featureJacobian(FeatureIDType feature_id, const std::vector<StateIDType> &cam_state_ids, Mat &H_x, std::vector<double> &r) {
    const auto &feature = map_server[feature_id];
    std::vector<StateIDType> valid;
    for (const auto &cid : cam_state_ids) {
        if (feature.observations.find(cid) == feature.observations.end()) continue;
        valid.push_back(cid);
    }
    const int rows = 4 * (int)valid.size();
    const int d = 21 + (int)state_server.cam_states.size() * 6;
    // [H_fj | H_xj | r_j]
    Mat M(rows, 3 + d + 1);
    int stack = 0;
    for (const auto &cid : valid) {
        Mat H_xi(4, 6), H_fi(4, 3);
        double r_i[4];
        measurementJacobian(cid, feature.id, H_xi, H_fi, r_i);
        auto it = state_server.cam_states.find(cid);
        int cntr = (int)std::distance(state_server.cam_states.begin(), it);
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 3; ++j) M(stack + i, j) = H_fi(i, j);
            for (int j = 0; j < 6; ++j) M(stack + i, 3 + 21 + 6 * cntr + j) = H_xi(i, j);
            M(stack + i, 3 + d) = r_i[i];
        }
        stack += 4;
    }
    householder_reduce(M, 3);
    H_x = Mat(rows - 3, d);
    r.assign(rows - 3, 0.0);
    for (int i = 3; i < rows; ++i) {
        for (int j = 0; j < d; ++j) H_x(i - 3, j) = M(i, 3 + j);
        r[i - 3] = M(i, 3 + d);
    }
}

// :778-907
void MsckfVio::measurementUpdate(const Mat &H, const std::vector<double> &r) {
    if (H.r == 0 || r.empty()) return;
    const int d = H.c;
    Mat H_thin; std::vector<double> r_thin;
    if (H.r > H.c) {
        Mat M(H.r, d + 1);
        for (int i = 0; i < H.r; ++i) { for (int j = 0; j < d; ++j) M(i, j) = H(i, j); M(i, d) = r[i]; }
        householder_reduce(M, d);
        H_thin = Mat(d, d);
        r_thin.assign(d, 0.0);
        for (int i = 0; i < d; ++i) { for (int j = i; j < d; ++j) H_thin(i, j) = M(i, j); r_thin[i] = M(i, d); }
    } else {
        H_thin = H; r_thin = r;
    }
    Mat &P = state_server.state_cov;
    Mat HP = H_thin * P;
    Mat S = HP * H_thin.t();
    for (int i = 0; i < S.r; ++i) S(i, i) += sh.observation_noise;
    Mat Kt = HP;  // S^-1 (H P)
    chol_solve(S, Kt);
    std::vector<double> delta_x(d, 0.0);
    for (int i = 0; i < d; ++i) { double s = 0; for (int k = 0; k < Kt.r; ++k) s += Kt(k, i) * r_thin[k]; delta_x[i] = s; }
    ++n_update;

    IMUState &s = state_server.imu_state;  // :876-885
    s.orientation = qmul(small_angle_quat(V3(delta_x[0], delta_x[1], delta_x[2])), s.orientation);
    for (int i = 0; i < 3; ++i) {
        s.gyro_bias[i] += delta_x[3 + i];
        s.velocity[i] += delta_x[6 + i];
        s.acc_bias[i] += delta_x[9 + i];
        s.position[i] += delta_x[12 + i];
    }
    s.R_imu_cam0 = quat_to_rot(small_angle_quat(V3(delta_x[15], delta_x[16], delta_x[17]))) * s.R_imu_cam0;
    for (int i = 0; i < 3; ++i) s.t_cam0_imu[i] += delta_x[18 + i];
    int ci = 0;
    for (auto &kv : state_server.cam_states) {  // :888-894
        const double *dx = &delta_x[21 + 6 * ci];
        kv.second.orientation = qmul(small_angle_quat(V3(dx[0], dx[1], dx[2])), kv.second.orientation);
        for (int i = 0; i < 3; ++i) kv.second.position[i] += dx[3 + i];
        ++ci;
    }
    // :897-904  P <- (I - K H) P, then symmetrise
    Mat I_KH = Mat::eye(d) - Kt.t() * H_thin;
    Mat Pn = I_KH * P;
    Mat Pt = Pn.t();
    for (size_t i = 0; i < Pn.d.size(); ++i) Pn.d[i] = (Pn.d[i] + Pt.d[i]) / 2.0;
    P = Pn;
}

// :909-935
bool MsckfVio::gatingTest(const Mat &H, const std::vector<double> &r, int dof) {
    Mat S = H * state_server.state_cov * H.t();
    for (int i = 0; i < S.r; ++i) S(i, i) += sh.observation_noise;
    Mat x(S.r, 1);
    for (int i = 0; i < S.r; ++i) x(i, 0) = r[i];
    chol_solve(S, x);
    double gamma = 0;
    for (int i = 0; i < S.r; ++i) gamma += r[i] * x(i, 0);
    return gamma < chi2_table[dof];
}

// :937-1024
void MsckfVio::removeLostFeatures() {
    int jacobian_row_size = 0;
    std::vector<FeatureIDType> invalid_ids, processed_ids;
    for (auto &kv : map_server) {
        auto &feature = kv.second;
        if (feature.observations.find(state_server.imu_state.id) != feature.observations.end()) continue;
        if (feature.observations.size() < 3) { invalid_ids.push_back(feature.id); continue; }
        if (!feature.is_initialized) {
            if (!feature_check_motion(feature, state_server.cam_states, sh)) { invalid_ids.push_back(feature.id); continue; }
            if (!feature_initialize_position(feature, state_server.cam_states, sh)) { invalid_ids.push_back(feature.id); continue; }
        }
        jacobian_row_size += 4 * (int)feature.observations.size() - 3;
        processed_ids.push_back(feature.id);
    }
    for (const auto &id : invalid_ids) map_server.erase(id);
    if (processed_ids.empty()) return;

    const int d = 21 + 6 * (int)state_server.cam_states.size();
    Mat H_x(jacobian_row_size, d);
    std::vector<double> r(jacobian_row_size, 0.0);
    int stack = 0;
    for (const auto &fid : processed_ids) {
        auto &feature = map_server[fid];
        std::vector<StateIDType> ids;
        for (const auto &m : feature.observations) ids.push_back(m.first);
        Mat H_xj; std::vector<double> r_j;
        featureJacobian(feature.id, ids, H_xj, r_j);
        if (gatingTest(H_xj, r_j, (int)ids.size() - 1)) {  // Q12
            H_x.set(stack, 0, H_xj);
            for (size_t i = 0; i < r_j.size(); ++i) r[stack + i] = r_j[i];
            stack += H_xj.r;
        }
        if (stack > cfg_.max_stack_rows) break;  // Q13
    }
    H_x.conservative_resize(stack, d);
    r.resize(stack);
    measurementUpdate(H_x, r);
    for (const auto &fid : processed_ids) map_server.erase(fid);
}

// :1026-1071
void MsckfVio::findRedundantCamStates(std::vector<StateIDType> &rm) {
    auto key_it = state_server.cam_states.end();
    for (int i = 0; i < 4; ++i) --key_it;
    auto cam_it = key_it; ++cam_it;
    auto first_it = state_server.cam_states.begin();
    const V3 key_position = key_it->second.position;
    const M3 key_rotation = quat_to_rot(key_it->second.orientation);
    for (int i = 0; i < 2; ++i) {
        const V3 position = cam_it->second.position;
        const M3 rotation = quat_to_rot(cam_it->second.orientation);
        const double distance = norm(position - key_position);
        const double angle = rot_angle(rotation * key_rotation.t());
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

// :1073-1184
void MsckfVio::pruneCamStateBuffer() {
    if ((int)state_server.cam_states.size() < cfg_.max_cam_state_size) return;
    std::vector<StateIDType> rm;
    findRedundantCamStates(rm);

    int jacobian_row_size = 0;
    for (auto &item : map_server) {
        auto &feature = item.second;
        std::vector<StateIDType> inv;
        for (const auto &cid : rm) if (feature.observations.find(cid) != feature.observations.end()) inv.push_back(cid);
        if (inv.empty()) continue;
        if (inv.size() == 1) { feature.observations.erase(inv[0]); continue; }
        if (!feature.is_initialized) {
            if (!feature_check_motion(feature, state_server.cam_states, sh)) {
                for (const auto &cid : inv) feature.observations.erase(cid);
                continue;
            }
            if (!feature_initialize_position(feature, state_server.cam_states, sh)) {
                for (const auto &cid : inv) feature.observations.erase(cid);
                continue;
            }
        }
        jacobian_row_size += 4 * (int)inv.size() - 3;
    }
    const int d = 21 + 6 * (int)state_server.cam_states.size();
    Mat H_x(jacobian_row_size, d);
    std::vector<double> r(jacobian_row_size, 0.0);
    int stack = 0;
    for (auto &item : map_server) {
        auto &feature = item.second;
        std::vector<StateIDType> inv;
        for (const auto &cid : rm) if (feature.observations.find(cid) != feature.observations.end()) inv.push_back(cid);
        if (inv.empty()) continue;
        Mat H_xj; std::vector<double> r_j;
        featureJacobian(feature.id, inv, H_xj, r_j);
        if (gatingTest(H_xj, r_j, (int)inv.size())) {  // Q12
            H_x.set(stack, 0, H_xj);
            for (size_t i = 0; i < r_j.size(); ++i) r[stack + i] = r_j[i];
            stack += H_xj.r;
        }
        for (const auto &cid : inv) feature.observations.erase(cid);
    }
    H_x.conservative_resize(stack, d);
    r.resize(stack);
    measurementUpdate(H_x, r);

    for (const auto &cid : rm) {  // :1161-1181
        int seq = (int)std::distance(state_server.cam_states.begin(), state_server.cam_states.find(cid));
        int s0 = 21 + 6 * seq, s1 = s0 + 6;
        Mat &P = state_server.state_cov;
        const int n = P.r;
        Mat Pn(n - 6, n - 6);
        for (int i = 0, ii = 0; i < n; ++i) {
            if (i >= s0 && i < s1) continue;
            for (int j = 0, jj = 0; j < n; ++j) {
                if (j >= s0 && j < s1) continue;
                Pn(ii, jj) = P(i, j);
                ++jj;
            }
            ++ii;
        }
        P = Pn;
        state_server.cam_states.erase(cid);
    }
}

// :1186-1236
void MsckfVio::onlineReset() {
    if (cfg_.position_std_threshold <= 0) return;
    const Mat &P = state_server.state_cov;
    const double sx = std::sqrt(P(12, 12)), sy = std::sqrt(P(13, 13)), sz = std::sqrt(P(14, 14));
    if (sx < cfg_.position_std_threshold && sy < cfg_.position_std_threshold && sz < cfg_.position_std_threshold) return;
    ++online_reset_counter;
    state_server.cam_states.clear();
    map_server.clear();
    resetCov();
}

// :1238-1305
void MsckfVio::publish(double t) {
    const IMUState &s = state_server.imu_state;
    SE3 T_i_w(quat_to_rot(s.orientation).t(), s.position);
    SE3 T_b_w = sh.T_imu_body * T_i_w * sh.T_imu_body.inv();
    mskf_pose pose;
    pose.time_stamp = t;
    for (int i = 0; i < 3; ++i) pose.p[i] = T_b_w.t[i];
    rot_to_hamilton(T_b_w.R, pose.q);
    poses.push_back(pose);
    path_.push_back(T_b_w.t);
}

}  // namespace orc
