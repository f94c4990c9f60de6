"""ctypes mirrors of include/mskf_types.h (POD only; no behaviour)."""
import ctypes as C

import numpy as np


class Calib(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("cam0_intrinsics", C.c_double * 4), ("cam0_distortion", C.c_double * 4),
        ("cam0_model", C.c_int32), ("cam1_model", C.c_int32),
        ("cam1_intrinsics", C.c_double * 4), ("cam1_distortion", C.c_double * 4),
        ("T_cam0_imu", C.c_double * 16), ("T_cam1_cam0", C.c_double * 16), ("T_imu_body", C.c_double * 16),
    ]


class FeCfg(C.Structure):
    _fields_ = [
        ("grid_row", C.c_int32), ("grid_col", C.c_int32),
        ("grid_min_feature_num", C.c_int32), ("grid_max_feature_num", C.c_int32),
        ("pyramid_levels", C.c_int32), ("patch_size", C.c_int32),
        ("fast_threshold", C.c_int32), ("max_iteration", C.c_int32),
        ("track_precision", C.c_double), ("ransac_threshold", C.c_double), ("stereo_threshold", C.c_double),
        ("det_rows", C.c_int32), ("det_cols", C.c_int32),
        ("compat_flags", C.c_int32), ("_pad", C.c_int32),
    ]


class EkfCfg(C.Structure):
    _fields_ = [
        ("frame_rate", C.c_double),
        ("max_cam_state_size", C.c_int32), ("chi2_mode", C.c_int32),
        ("position_std_threshold", C.c_double), ("rotation_threshold", C.c_double),
        ("translation_threshold", C.c_double), ("tracking_rate_threshold", C.c_double),
        ("feature_translation_threshold", C.c_double),
        ("noise_gyro", C.c_double), ("noise_acc", C.c_double), ("noise_gyro_bias", C.c_double),
        ("noise_acc_bias", C.c_double), ("noise_feature", C.c_double),
        ("init_velocity", C.c_double * 3),
        ("cov_velocity", C.c_double), ("cov_gyro_bias", C.c_double), ("cov_acc_bias", C.c_double),
        ("cov_ext_rot", C.c_double), ("cov_ext_trans", C.c_double),
        ("max_stack_rows", C.c_int32), ("compression_mode", C.c_int32),
    ]


class ImuSample(C.Structure):
    _fields_ = [("time_stamp", C.c_double), ("angular_velocity", C.c_double * 3),
                ("linear_acceleration", C.c_double * 3)]


class TrackingInfo(C.Structure):
    _fields_ = [("time_stamp", C.c_double), ("before_tracking", C.c_int32), ("after_tracking", C.c_int32),
                ("after_matching", C.c_int32), ("after_ransac", C.c_int32)]


# numpy dtypes of the array records
POINT2F = np.dtype([("x", "<f4"), ("y", "<f4")])
CORNER = np.dtype([("x", "<f4"), ("y", "<f4"), ("score", "<i4"), ("cell", "<i4")])
FEATURE_MEAS = np.dtype([("id", "<u4"), ("_pad", "<u4"), ("u0", "<f8"), ("v0", "<f8"), ("u1", "<f8"), ("v1", "<f8")])
POSE = np.dtype([("t", "<f8"), ("p", "<f8", 3), ("q", "<f8", 4)])

COMPAT_REFERENCE = 15     # Q1 | Q2 | Q4 | Q5 (include/mskf_types.h)


def default_fe_cfg(grid_row=4, grid_col=5, grid_min=3, grid_max=4, compat=COMPAT_REFERENCE):
    """config/app_imgproc.yaml values of the reference + CornerDetector(30, 47, thr) (image_processor.cpp:132)."""
    c = FeCfg()
    c.grid_row, c.grid_col, c.grid_min_feature_num, c.grid_max_feature_num = grid_row, grid_col, grid_min, grid_max
    c.pyramid_levels, c.patch_size, c.fast_threshold, c.max_iteration = 3, 15, 10, 30
    c.track_precision, c.ransac_threshold, c.stereo_threshold = 0.01, 3.0, 5.0
    c.det_rows, c.det_cols = 30, 47
    c.compat_flags = compat
    return c


def default_ekf_cfg(max_cam_state_size=20, compression_mode=3):
    """config/app_msckfvio.yaml values of the reference.  compression_mode: 3 (default) the reference's own rule - Householder QR
    when the stack has more rows than columns, nothing otherwise (msckf_vio.cpp:795-821) -, 0 auto (Gram + regularised Cholesky,
    Householder where the device asks for it), 1 Gram + Cholesky only, 2 Householder TSQR always."""
    c = EkfCfg()
    c.frame_rate = 20.0
    c.max_cam_state_size = max_cam_state_size
    c.chi2_mode = 0
    c.position_std_threshold, c.rotation_threshold = 8.0, 0.2618
    c.translation_threshold, c.tracking_rate_threshold = 0.4, 0.5
    c.feature_translation_threshold = -1.0
    c.noise_gyro, c.noise_acc, c.noise_gyro_bias, c.noise_acc_bias, c.noise_feature = 0.005, 0.05, 0.001, 0.01, 0.035
    c.init_velocity[:] = [0.0, 0.0, 0.0]
    c.cov_velocity, c.cov_gyro_bias, c.cov_acc_bias = 0.25, 0.01, 0.01
    c.cov_ext_rot, c.cov_ext_trans = 3.0462e-4, 2.5e-5
    c.max_stack_rows = 1500
    c.compression_mode = compression_mode
    return c


# synth::RenderImg (csrc/synth/synth.h): what one image of a synthetic sequence is rendered from (input generator, host and device)
RENDER_IMG = np.dtype([("R_wc", "<f8", 9), ("o", "<f8", 3), ("pixel_sigma", "<f8"), ("seed", "<u4"), ("k2c", "<i4"), ("cam", "<i4"), ("pad", "<i4")])
