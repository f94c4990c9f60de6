/* mskf_types.h — plain-C POD types shared by the C-ABI (mskf_hip.h), the host C++
 * mirror of the reference classes and the test oracle.  No behaviour here.
 *
 * Field provenance (reference file:line, relative to the reference tree):
 *   mskf_calib     : config/camchain-imucam-euroc.yaml:4-37, read at
 *                    msckf_core/src/image_processor.cpp:52-72 and msckf_core/src/msckf_vio.cpp:114-125
 *   mskf_fe_cfg    : config/app_imgproc.yaml:2-12 (image_processor.cpp:75-86) + the
 *                    CornerDetector(30, 47, thr) literal at image_processor.cpp:132
 *   mskf_ekf_cfg   : config/app_msckfvio.yaml:2-26 (msckf_vio.cpp:58-128), Feature LM defaults feature.hpp:46-52
 *   mskf_feature_meas : msckf_core/include/common/data_msg.h:30-36 (FeatureMeasurement, 40 bytes)
 *   mskf_imu_sample   : data_msg.h:22-26 (Imu)
 */
#ifndef MSKF_TYPES_H
#define MSKF_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MSKF_MODEL_RADTAN = 0, MSKF_MODEL_EQUIDISTANT = 1 };   /* distortion_model of camchain yaml (image_processor.cpp:809-816,840); both run on the device */

/* compat switches for the reference quirks of SURVEY.md §2.3 */
enum {
    MSKF_COMPAT_Q1_MSG_ACCUMULATE = 1 << 0, /* feature message never cleared (image_processor.cpp:1157-1164) */
    MSKF_COMPAT_Q2_PREV_ALIAS     = 1 << 1, /* prev/curr image timestamps alias -> R_p_c == I (image_processor.cpp:192) */
    MSKF_COMPAT_Q4_RESPONSE_INDEX = 1 << 2, /* responses read in detection order after the sieve (image_processor.cpp:698) */
    MSKF_COMPAT_Q5_NO_RANSAC      = 1 << 3, /* both twoPointRansac calls commented out (image_processor.cpp:482-500); cleared: the
                                               2-point RANSAC of :911-1135 runs on the cam0 and cam1 temporal pairs */
    MSKF_COMPAT_REFERENCE         = 15
};

typedef struct mskf_calib {
    int32_t width, height;
    double cam0_intrinsics[4];   /* fx fy cx cy */
    double cam0_distortion[4];   /* k1 k2 p1 p2 (radtan) */
    int32_t cam0_model, cam1_model;
    double cam1_intrinsics[4];
    double cam1_distortion[4];
    double T_cam0_imu[16];       /* cam0.T_cam_imu, row-major 4x4 */
    double T_cam1_cam0[16];      /* cam1.T_cn_cnm1 */
    double T_imu_body[16];
} mskf_calib;

typedef struct mskf_fe_cfg {
    int32_t grid_row, grid_col, grid_min_feature_num, grid_max_feature_num;
    int32_t pyramid_levels, patch_size, fast_threshold, max_iteration; /* levels/patch/iter are inert in the reference (Q6) */
    double track_precision, ransac_threshold, stereo_threshold;
    int32_t det_rows, det_cols;  /* CornerDetector(30, 47, thr) */
    int32_t compat_flags;
    int32_t _pad;
} mskf_fe_cfg;

typedef struct mskf_ekf_cfg {
    double frame_rate;
    int32_t max_cam_state_size;
    int32_t chi2_mode;           /* 0: ppf(0.05) (upstream, msckf_vio.cpp:182-183), 1: ppf(0.95) (Q11) */
    double position_std_threshold, rotation_threshold, translation_threshold, tracking_rate_threshold;
    double feature_translation_threshold;
    double noise_gyro, noise_acc, noise_gyro_bias, noise_acc_bias, noise_feature; /* std, squared on load */
    double init_velocity[3];
    double cov_velocity, cov_gyro_bias, cov_acc_bias, cov_ext_rot, cov_ext_trans;
    int32_t max_stack_rows;      /* 1500, msckf_vio.cpp:1009 */
    int32_t compression_mode;    /* QR compression of the stacked Jacobian (msckf_vio.cpp:795-817): 0 = auto (Gram + regularised
                                    Cholesky; Householder TSQR when the stack has no more STACKED rows than active columns, or
                                    when the factorisation finds that its lambda prior would move the posterior covariance by
                                    more than 1e-6 relative: lambda max(P_aa) / sigma^2; the count of near-zero pivots is
                                    reported in diag_out[1] but decides nothing), 1 = Gram only, 2 = Householder TSQR always,
                                    3 = the reference's own rule, literally (:795-821): Householder QR when the stack has more
                                    rows than (active) columns, no compression otherwise.  Any other value is refused by
                                    mskf_stream_create */
} mskf_ekf_cfg;

typedef struct mskf_feature_meas {  /* == cg::FeatureMeasurement */
    uint32_t id;
    uint32_t _pad;
    double u0, v0, u1, v1;
} mskf_feature_meas;

typedef struct mskf_imu_sample {    /* == cg::Imu */
    double time_stamp;
    double angular_velocity[3];
    double linear_acceleration[3];
} mskf_imu_sample;

typedef struct mskf_point2f { float x, y; } mskf_point2f;

/* one detector candidate: per-cell best corner */
typedef struct mskf_corner {
    float x, y;
    int32_t score;     /* integer Shi-Tomasi score, response = score / 256.0 */
    int32_t cell;
} mskf_corner;

typedef struct mskf_tracking_info { /* == cg::TrackingInfo, data_msg.h:48-55 */
    double time_stamp;
    int32_t before_tracking, after_tracking, after_matching, after_ransac;
} mskf_tracking_info;

/* pose record written to pose_out.txt (msckf_vio.cpp:1256-1258): t, p, Hamilton q (x y z w) */
typedef struct mskf_pose {
    double time_stamp;
    double p[3];
    double q[4];
} mskf_pose;

#ifdef __cplusplus
}
#endif
#endif /* MSKF_TYPES_H */
