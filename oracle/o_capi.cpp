// oracle/o_capi.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).  C entry points for ctypes
// (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Never linked into the product.
#include <cstring>
#include <memory>
#include "o_filter.h"
#include "o_frontend.h"
#include "o_image.h"

using namespace orc;

namespace {
Img wrap(const uint8_t *p, int w, int h) {
    Img im(w, h);
    std::memcpy(im.d.data(), p, (size_t)w * h);
    return im;
}
struct OSystem {  // cg::System (system.cpp:11-54): one ImageProcessor + one MsckfVio, three callbacks
    ImageProcessor fe;
    MsckfVio vio;
    OSystem(const mskf_calib &c, const mskf_fe_cfg &f, const mskf_ekf_cfg &e) : fe(c, f), vio(c, e) {}
};
}  // namespace

extern "C" {

void orc_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst) {
    Img d;
    pyr_down(wrap(src, w, h), d);
    std::memcpy(dst, d.d.data(), d.d.size());
}

// levels 1..3 written back to back into dst (sizes ((w+1)/2 x (h+1)/2) ...)
void orc_build_pyramid(const uint8_t *src, int w, int h, uint8_t *dst) {
    std::vector<Img> pyr;
    build_pyramid(wrap(src, w, h), pyr);
    size_t off = 0;
    for (int l = 1; l < LK_LEVELS; ++l) { std::memcpy(dst + off, pyr[l].d.data(), pyr[l].d.size()); off += pyr[l].d.size(); }
}

void orc_lk_stats(long long *out3, int reset) {
    for (int i = 0; i < 3; ++i) { out3[i] = g_lk_stats[i]; if (reset) g_lk_stats[i] = 0; }
}

void orc_lk_track(const uint8_t *imgA, const uint8_t *imgB, int w, int h, int n, const mskf_point2f *ptsA,
                  mskf_point2f *ptsB, uint8_t *status) {
    std::vector<Img> pa, pb;
    build_pyramid(wrap(imgA, w, h), pa);
    build_pyramid(wrap(imgB, w, h), pb);
    std::vector<mskf_point2f> a(ptsA, ptsA + n), b(ptsB, ptsB + n);
    std::vector<uint8_t> st;
    lk_track(pa, pb, a, b, st);
    for (int i = 0; i < n; ++i) { ptsB[i] = b[i]; status[i] = st[i]; }
}

// all det_rows*det_cols per-cell maxima (score 0 = none)
void orc_cell_maxima(const uint8_t *img, int w, int h, int det_rows, int det_cols, mskf_corner *out) {
    CornerDetector d(det_rows, det_cols, 0);
    d.set_image_size(w, h);
    std::vector<mskf_corner> mx;
    d.cell_maxima(wrap(img, w, h), mx);
    std::memcpy(out, mx.data(), mx.size() * sizeof(mskf_corner));
}

int orc_detect(const uint8_t *img, int w, int h, int det_rows, int det_cols, int thr, const uint8_t *occupancy,
               mskf_point2f *pts, double *responses) {
    CornerDetector d(det_rows, det_cols, thr);
    d.set_image_size(w, h);
    if (occupancy) d.occupancy.assign(occupancy, occupancy + (size_t)det_rows * det_cols);
    std::vector<mskf_point2f> p;
    std::vector<double> r;
    d.detect_features(wrap(img, w, h), p, r);
    for (size_t i = 0; i < p.size(); ++i) { pts[i] = p[i]; responses[i] = r[i]; }
    return (int)p.size();
}

void orc_undistort(const double K[4], const double D[4], int model, const double R[9], const double Pn[4], int n,
                   const mskf_point2f *in, mskf_point2f *out) {
    CamModel c; for (int i = 0; i < 4; ++i) { c.K[i] = K[i]; c.D[i] = D[i]; } c.model = model;
    for (int i = 0; i < n; ++i) undistort_point(c, R, Pn, in[i].x, in[i].y, out[i].x, out[i].y);
}
void orc_distort(const double K[4], const double D[4], int model, int n, const mskf_point2f *in, mskf_point2f *out) {
    CamModel c; for (int i = 0; i < 4; ++i) { c.K[i] = K[i]; c.D[i] = D[i]; } c.model = model;
    for (int i = 0; i < n; ++i) distort_point(c, in[i].x, in[i].y, out[i].x, out[i].y);
}

// stereoMatch on a fresh image pair (image_processor.cpp:534-620): cam1 initial guess by projection
void orc_stereo_match(const mskf_calib *calib, const mskf_fe_cfg *cfg, const uint8_t *cam0, const uint8_t *cam1, int n,
                      const mskf_point2f *pts0, mskf_point2f *pts1, uint8_t *inliers) {
    ImageProcessor ip(*calib, *cfg);
    ip.test_set_images(wrap(cam0, calib->width, calib->height), wrap(cam1, calib->width, calib->height));
    std::vector<mskf_point2f> a(pts0, pts0 + n), b;
    std::vector<uint8_t> m;
    ip.stereoMatch(a, b, m);
    for (int i = 0; i < n; ++i) { pts1[i] = b[i]; inliers[i] = m[i]; }
}

// twoPointRansac (image_processor.cpp:911-1135) on one camera's temporal pairs; `draws` is the state of the draw counter
void orc_two_point_ransac(const mskf_calib *calib, const mskf_fe_cfg *cfg, int cam, int n, const mskf_point2f *pts1, const mskf_point2f *pts2,
                          const double *R_p_c, double inlier_error, double success_probability, unsigned long long *draws, int32_t *markers) {
    ImageProcessor ip(*calib, *cfg);
    CamModel c;
    for (int i = 0; i < 4; ++i) { c.K[i] = cam ? calib->cam1_intrinsics[i] : calib->cam0_intrinsics[i]; c.D[i] = cam ? calib->cam1_distortion[i] : calib->cam0_distortion[i]; }
    c.model = cam ? calib->cam1_model : calib->cam0_model;
    M3 R; for (int i = 0; i < 9; ++i) R.m[i] = R_p_c[i];
    ip.ransac_draws = *draws;
    std::vector<mskf_point2f> a(pts1, pts1 + n), b(pts2, pts2 + n);
    std::vector<int> m;
    ip.twoPointRansac(a, b, R, c, inlier_error, success_probability, m);
    for (int i = 0; i < n; ++i) markers[i] = m[i];
    *draws = ip.ransac_draws;
}

// ---------------------------------------------------------------- System
void *orc_system_create(const mskf_calib *c, const mskf_fe_cfg *f, const mskf_ekf_cfg *e) { return new OSystem(*c, *f, *e); }
void orc_system_destroy(void *h) { delete (OSystem *)h; }
void orc_system_imu(void *h, const mskf_imu_sample *s) {  // system.cpp:45-48
    OSystem *S = (OSystem *)h;
    S->fe.imuCallback(*s);
    S->vio.imuCallback(*s);
}
void orc_system_stereo(void *h, const uint8_t *cam0, const uint8_t *cam1, int w, int hgt, double t) {  // system.cpp:40-43
    OSystem *S = (OSystem *)h;
    S->fe.stereoCallback(wrap(cam0, w, hgt), wrap(cam1, w, hgt), t, t);
}
void orc_system_backend(void *h) {  // system.cpp:50-54
    OSystem *S = (OSystem *)h;
    S->vio.featureCallback(*S->fe.feature_msg_ptr_);
}
int orc_system_num_features(void *h) { return (int)((OSystem *)h)->fe.last_dump.ids.size(); }
void orc_system_get_dump(void *h, uint64_t *ids, int32_t *lifetime, mskf_point2f *cam0, mskf_point2f *cam1, mskf_tracking_info *info) {
    const FrameDump &d = ((OSystem *)h)->fe.last_dump;
    for (size_t i = 0; i < d.ids.size(); ++i) { ids[i] = d.ids[i]; lifetime[i] = d.lifetime[i]; cam0[i] = d.cam0[i]; cam1[i] = d.cam1[i]; }
    *info = d.info;
}
int orc_system_msg_size(void *h) { return (int)((OSystem *)h)->fe.feature_msg_ptr_->features.size(); }
void orc_system_get_msg(void *h, mskf_feature_meas *out) {
    const auto &f = ((OSystem *)h)->fe.feature_msg_ptr_->features;
    std::memcpy(out, f.data(), f.size() * sizeof(mskf_feature_meas));
}
int orc_system_num_poses(void *h) { return (int)((OSystem *)h)->vio.poses.size(); }
void orc_system_get_poses(void *h, mskf_pose *out) {
    const auto &p = ((OSystem *)h)->vio.poses;
    std::memcpy(out, p.data(), p.size() * sizeof(mskf_pose));
}
int orc_system_state_dim(void *h) { return ((OSystem *)h)->vio.state_server.state_cov.r; }
void orc_system_get_cov(void *h, double *out) {
    const Mat &P = ((OSystem *)h)->vio.state_server.state_cov;
    std::memcpy(out, P.d.data(), P.d.size() * sizeof(double));
}
// imu state: q(4) p(3) v(3) bg(3) ba(3) R_imu_cam0(9) t_cam0_imu(3) = 28 doubles
void orc_system_get_imu_state(void *h, double *out) {
    const IMUState &s = ((OSystem *)h)->vio.state_server.imu_state;
    int k = 0;
    for (int i = 0; i < 4; ++i) out[k++] = s.orientation[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.position[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.velocity[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.gyro_bias[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.acc_bias[i];
    for (int i = 0; i < 9; ++i) out[k++] = s.R_imu_cam0.m[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.t_cam0_imu[i];
}
int orc_system_num_updates(void *h) { return ((OSystem *)h)->vio.n_update; }
long long orc_system_num_resets(void *h) { return ((OSystem *)h)->vio.online_reset_counter; }
int orc_system_map_size(void *h) { return (int)((OSystem *)h)->vio.map_server.size(); }

// ---------------------------------------------------------------- EKF unit problem
// One measurement update on an explicit problem (rows a9-a13 of SURVEY §8a):
//   clones: n_clones x 14 doubles (q[4] p[3] q_null[4] p_null[3]); P: d x d row-major, d = 21 + 6 n_clones
//   features: position[3] per feature, obs_start[n_feat+1], obs_clone[n_obs], obs_z[n_obs*4]
//   dof_offset: gating dof = n_obs_j + dof_offset (-1 lost features, 0 pruning; Q12)
// Outputs: gamma[n_feat], pass[n_feat], delta_x[d], P updated in place; returns stacked row count.
// last stacked system handed to measurementUpdate by orc_ekf_update_problem (row-major H: rows x d, then r)
static std::vector<double> g_last_H, g_last_r;
static int g_last_rows = 0, g_last_d = 0;
int orc_ekf_last_system(double *H, double *r, int capacity_rows) {
    if (H && r && capacity_rows >= g_last_rows) {
        std::memcpy(H, g_last_H.data(), sizeof(double) * g_last_H.size());
        std::memcpy(r, g_last_r.data(), sizeof(double) * g_last_r.size());
    }
    return g_last_rows;
}

int orc_ekf_update_problem(const mskf_calib *calib, const mskf_ekf_cfg *cfg, const double gravity[3], int n_clones,
                           const double *clones, double *P, int n_feat, const double *positions, const int *obs_start,
                           const int *obs_clone, const double *obs_z, int dof_offset, double *gamma_out,
                           uint8_t *pass_out, double *delta_x) {
    MsckfVio vio(*calib, *cfg);
    vio.sh.gravity = V3(gravity[0], gravity[1], gravity[2]);
    const int d = 21 + 6 * n_clones;
    for (int i = 0; i < n_clones; ++i) {
        CAMState cs;
        cs.id = i;
        const double *c = clones + 14 * i;
        cs.orientation = Quat(c[0], c[1], c[2], c[3]);
        cs.position = V3(c[4], c[5], c[6]);
        cs.orientation_null = Quat(c[7], c[8], c[9], c[10]);
        cs.position_null = V3(c[11], c[12], c[13]);
        vio.state_server.cam_states[i] = cs;
    }
    vio.state_server.state_cov = Mat(d, d);
    std::memcpy(vio.state_server.state_cov.d.data(), P, sizeof(double) * d * d);
    // snapshot pre-update state to derive delta_x from the additive parts
    int total_rows = 0;
    for (int j = 0; j < n_feat; ++j) total_rows += 4 * (obs_start[j + 1] - obs_start[j]) - 3;
    Mat H_x(total_rows, d);
    std::vector<double> r(total_rows, 0.0);
    int stack = 0;
    for (int j = 0; j < n_feat; ++j) {
        Feature &f = vio.map_server[j];
        f.id = j;
        f.position = V3(positions[3 * j], positions[3 * j + 1], positions[3 * j + 2]);
        f.is_initialized = true;
        std::vector<StateIDType> ids;
        for (int o = obs_start[j]; o < obs_start[j + 1]; ++o) {
            f.observations[obs_clone[o]] = {obs_z[4 * o], obs_z[4 * o + 1], obs_z[4 * o + 2], obs_z[4 * o + 3]};
            ids.push_back(obs_clone[o]);
        }
    }
    bool capped = false;
    for (int j = 0; j < n_feat; ++j) {
        gamma_out[j] = -1; pass_out[j] = 0;
        if (capped) continue;
        Feature &f = vio.map_server[j];
        std::vector<StateIDType> ids;
        for (const auto &m : f.observations) ids.push_back(m.first);
        Mat H_xj; std::vector<double> r_j;
        vio.featureJacobian(j, ids, H_xj, r_j);
        // gamma (same algebra as gatingTest, msckf_vio.cpp:909-935)
        Mat S = H_xj * vio.state_server.state_cov * H_xj.t();
        for (int i = 0; i < S.r; ++i) S(i, i) += vio.sh.observation_noise;
        Mat x(S.r, 1);
        for (int i = 0; i < S.r; ++i) x(i, 0) = r_j[i];
        chol_solve(S, x);
        double g = 0;
        for (int i = 0; i < S.r; ++i) g += r_j[i] * x(i, 0);
        gamma_out[j] = g;
        const bool pass = vio.gatingTest(H_xj, r_j, (int)ids.size() + dof_offset);
        pass_out[j] = pass ? 1 : 0;
        if (pass) {
            H_x.set(stack, 0, H_xj);
            for (size_t i = 0; i < r_j.size(); ++i) r[stack + i] = r_j[i];
            stack += H_xj.r;
        }
        if (dof_offset < 0 && stack > cfg->max_stack_rows) capped = true;  // Q13 applies to removeLostFeatures only
    }
    H_x.conservative_resize(stack, d);
    r.resize(stack);
    // delta_x: recompute exactly as measurementUpdate does (K r) so callers can compare it
    IMUState before = vio.state_server.imu_state;
    std::vector<V3> cam_p_before;
    for (auto &kv : vio.state_server.cam_states) cam_p_before.push_back(kv.second.position);
    g_last_rows = stack; g_last_d = d;
    g_last_H.assign((size_t)stack * d, 0.0);
    for (int i = 0; i < stack; ++i) for (int j = 0; j < d; ++j) g_last_H[(size_t)i * d + j] = H_x(i, j);
    g_last_r = r;
    vio.measurementUpdate(H_x, r);
    for (int i = 0; i < d; ++i) delta_x[i] = 0;
    if (stack > 0) {
        const IMUState &s = vio.state_server.imu_state;
        for (int i = 0; i < 3; ++i) {
            delta_x[3 + i] = s.gyro_bias[i] - before.gyro_bias[i];
            delta_x[6 + i] = s.velocity[i] - before.velocity[i];
            delta_x[9 + i] = s.acc_bias[i] - before.acc_bias[i];
            delta_x[12 + i] = s.position[i] - before.position[i];
            delta_x[18 + i] = s.t_cam0_imu[i] - before.t_cam0_imu[i];
        }
        // rotation parts: dq = q_new * q_old^-1 ~ [dtheta/2, 1]
        auto dth = [](const Quat &qn, const Quat &qo, double *o) {
            Quat qi(-qo[0], -qo[1], -qo[2], qo[3]);
            Quat dq = qmul(qn, qi);
            for (int i = 0; i < 3; ++i) o[i] = 2 * dq[i];
        };
        dth(s.orientation, before.orientation, delta_x);
        dth(rot_to_quat(s.R_imu_cam0), rot_to_quat(before.R_imu_cam0), delta_x + 15);
        int ci = 0;
        for (auto &kv : vio.state_server.cam_states) {
            const double *c = clones + 14 * ci;
            dth(kv.second.orientation, Quat(c[0], c[1], c[2], c[3]), delta_x + 21 + 6 * ci);
            for (int i = 0; i < 3; ++i) delta_x[21 + 6 * ci + 3 + i] = kv.second.position[i] - cam_p_before[ci][i];
            ++ci;
        }
    }
    std::memcpy(P, vio.state_server.state_cov.d.data(), sizeof(double) * d * d);
    return stack;
}

// batched LM triangulation (feature.hpp:289-450) on explicit inputs: returns validity per feature
void orc_triangulate(const mskf_calib *calib, int n_clones, const double *clones, int n_feat, const int *obs_start,
                     const int *obs_clone, const double *obs_z, double *positions, uint8_t *valid) {
    FilterShared sh;
    sh.T_cam0_cam1 = SE3::from16(calib->T_cam1_cam0);
    CamStateServer cams;
    for (int i = 0; i < n_clones; ++i) {
        CAMState cs;
        cs.id = i;
        const double *c = clones + 14 * i;
        cs.orientation = Quat(c[0], c[1], c[2], c[3]);
        cs.position = V3(c[4], c[5], c[6]);
        cams[i] = cs;
    }
    for (int j = 0; j < n_feat; ++j) {
        Feature f;
        f.id = j;
        for (int o = obs_start[j]; o < obs_start[j + 1]; ++o)
            f.observations[obs_clone[o]] = {obs_z[4 * o], obs_z[4 * o + 1], obs_z[4 * o + 2], obs_z[4 * o + 3]};
        valid[j] = feature_initialize_position(f, cams, sh) ? 1 : 0;
        positions[3 * j] = f.position[0]; positions[3 * j + 1] = f.position[1]; positions[3 * j + 2] = f.position[2];
    }
}

}  // extern "C"
