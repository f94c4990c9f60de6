// fe_device.h — device-side descriptors shared by the front-end kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/mskf_types.h"

#define MSKF_COPY_SEGS 6        // segments of one staging-copy launch (k_mskf_copy)

#define MSKF_LEVELS 4

// One image pyramid resident in HBM: level l is a dense row-major u8 plane, pitch == width.
struct PyrDev {
    const uint8_t *lvl[MSKF_LEVELS];
    int w[MSKF_LEVELS], h[MSKF_LEVELS];
};

struct CamDev {
    double K[4];  // fx fy cx cy
    double D[4];  // radtan: k1 k2 p1 p2; equidistant: k1 k2 k3 k4
    int model;    // MSKF_MODEL_RADTAN / MSKF_MODEL_EQUIDISTANT
    int pad_;
};

// Per VIO stream, per launch: everything the point kernels need.
struct FeStreamDev {
    PyrDev prev0, curr0, curr1;
    CamDev cam0, cam1;
    double R01[9];     // R_cam0_cam1 (image_processor.cpp:544)
    double E[9];       // [t]x R       (:587-591)
    double Hpred[9];   // K R_p_c K^-1 (:335-340)
    double epi_thresh; // stereo_threshold * norm_pixel_unit (:606,:615)
    int n_pts;
    const int *n_pts_dev;          // when set: the point count lives on the device (fe_book1's candidate count), n_pts is ignored
    int do_temporal;   // 1: prev0 -> curr0 LK first (trackFeatures), 0: stereo only (new candidates)
    const mskf_point2f *in_pts;   // prev cam0 points (temporal) or cam0 candidates (stereo only)
    mskf_point2f *out0;           // tracked cam0 point (temporal) / copy of the input
    mskf_point2f *out1;           // matched cam1 point
    mskf_point2f *und0, *und1;    // undistorted normalised coords of out0/out1 (publish, :1154-1155)
    uint8_t *status;              // bit0: temporal track ok (incl. bounds), bit1: stereo inlier
    // detector
    int det_rows, det_cols, cell_w, cell_h;
    int det_floor;                 // only scores above this are recorded (mskf_fe_set_detect_floor; 0 = every positive score)
    unsigned long long *cell_keys; // det_rows*det_cols per-cell maxima of curr0 level 0 as keys score<<32 | ~order (0 = none)
};
