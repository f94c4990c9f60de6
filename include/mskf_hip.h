/* mskf_hip.h — C ABI of the MI355X (gfx950) hot path: stereo KLT front-end + MSCKF measurement update.
 *
 * This is the drop-in boundary: the host-side C++ mirror of the reference classes
 * (msckf_stereo_c_amd/csrc/host/: cg::ImageProcessor, cg::MsckfVio, cg::System) calls ONLY these
 * entry points; a maintainer of the reference would bind the same symbols from
 * msckf_core/src/image_processor.cpp and msckf_core/src/msckf_vio.cpp.
 *
 * Conventions: plain pointers and sizes, no C++ / torch types; every function returns MSKF_OK (0)
 * or a negative mskf_status and never throws; the caller owns all host buffers; device buffers are
 * owned by the handles; calls on one mskf_ctx are serialised by the caller, different contexts may be
 * driven from different host threads.  One mskf_ctx = one HIP stream on one GPU plus the batch
 * staging buffers; one mskf_stream = one VIO stream (one ImageProcessor + one MsckfVio state).
 * The *_batch entry points process one frame of many VIO streams per kernel launch; the single
 * stream entry points are batches of one.
 *
 * There is no CPU fallback: every compute entry point fails with MSKF_ERR_NO_DEVICE / MSKF_ERR_HIP
 * when no gfx950 device or code object is available.
 */
#ifndef MSKF_HIP_H
#define MSKF_HIP_H

#include "mskf_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mskf_status {
    MSKF_OK = 0,
    MSKF_ERR_INVALID = -1,      /* bad argument */
    MSKF_ERR_HIP = -2,          /* a HIP runtime call failed (mskf_last_error() has the text) */
    MSKF_ERR_UNSUPPORTED = -3,  /* e.g. an unknown distortion model */
    MSKF_ERR_CAPACITY = -4,     /* more points / clones / rows than the stream was created for */
    MSKF_ERR_NO_DEVICE = -5
} mskf_status;

typedef struct mskf_ctx mskf_ctx;
typedef struct mskf_stream mskf_stream;

const char *mskf_last_error(void);
int mskf_abi_version(void);   /* 2: update args carry diag_out, mskf_ekf_cfg.compression_mode, *_begin / *_end entry points; 3 (round 3): update args carry
                                 pos_var_out, mskf_fe_frame_batch_* (whole front-end frames on the device), mskf_ctx_timing_gate, mskf_ctx_set_wait_mode;
                                 4 (round 4): the 2-point RANSAC inside the device frame (mskf_fe_frame_args.R_p_c / ransac_draws, mskf_fe_set_grid's draw counter) */

int mskf_ctx_create(int device, mskf_ctx **out);
/* Same, with the context's HIP stream created at the device's most urgent priority when high_priority != 0.
 * The batch runner puts the filter stage of a pipelined group (the serial dependency chain of the step) on such a
 * context so that its short kernels are dispatched ahead of the front-end's wide ones. */
int mskf_ctx_create_prio(int device, int high_priority, mskf_ctx **out);
/* A second context (own staging arenas, events, timing) whose work is enqueued on `parent`'s HIP stream: two host threads
 * (e.g. the front-end and the filter of one group of streams) can then feed one hardware queue.  `parent` must outlive it. */
int mskf_ctx_create_shared(mskf_ctx *parent, mskf_ctx **out);
void mskf_ctx_destroy(mskf_ctx *ctx);
int mskf_ctx_sync(mskf_ctx *ctx);
/* How the *_end calls of this context wait for the device: 0 = spin on a completion mark in pinned host memory (lowest
 * latency, occupies a core), 1 = park the thread on a blocking HIP event.  Default: MSKF_WAIT=block in the environment
 * selects 1, else 0.  A stage that waits once per frame for a long chain of kernels (the device front-end frame) loses
 * nothing by parking and leaves its core to the threads that have host work. */
int mskf_ctx_set_wait_mode(mskf_ctx *ctx, int block);
/* the HIP stream of this context as a void* (hipStream_t), for event timing by the caller */
void *mskf_ctx_hip_stream(mskf_ctx *ctx);

/* Optional per-kernel timing with HIP events recorded on the context's own stream (bench / roofline).
 * `units` are the algorithmic work units of a launch (LK: point tracks, temporal + stereo of one track call in one launch — MSKF_K_PT_GEOM
 * is kept for ABI stability and stays empty since the per-point geometry moved into that launch; pyramid: output pixels;
 * EKF feature / GEMM / Cholesky / TRSM kernels: algorithmic FP64 flops; others: streams).  Disabled by default.
 * `enable` = n > 1 times every n-th launch of each kind only and scales the sums to all launches: two event records per launch cost
 * 7 % of the throughput at the C2 bench shape and 36 % at C5 (one stream per launch), measured. */
enum {
    MSKF_K_PYR = 0, MSKF_K_DETECT, MSKF_K_LK, MSKF_K_EKF_PROPAGATE, MSKF_K_EKF_AUGMENT, MSKF_K_EKF_FEATURES,
    MSKF_K_EKF_TSQR /* k_ekf_tsqr; rounds 1-3: the stacking-decision kernel, which no longer exists */, MSKF_K_EKF_GEMM, MSKF_K_EKF_CHOL, MSKF_K_EKF_TRSM, MSKF_K_EKF_SMALL, MSKF_K_EKF_REMOVE, MSKF_K_PT_GEOM, MSKF_K_FE_BOOK, MSKF_K_COUNT
};
int mskf_ctx_set_timing(mskf_ctx *ctx, int enable);
/* Accounting gate (default on): while it is off, launches are not timed and host seconds not accumulated.  Unlike
 * mskf_ctx_set_timing it does not synchronise, so the thread that drives the context can open and close a measurement
 * window in the middle of a running pipeline; launches already begun keep the state they were begun with. */
int mskf_ctx_timing_gate(mskf_ctx *ctx, int on);
/* host seconds spent inside the batched entry points of this context, outside the device waits: [0] mskf_ekf_update_batch
 * packing, [1] its unpacking, [2] mskf_fe_track_batch packing, [3] its unpacking */
int mskf_ctx_get_host_time(mskf_ctx *ctx, double out[4], int reset);
/* arrays of MSKF_K_COUNT entries: accumulated milliseconds, launches and units since the last reset */
int mskf_ctx_get_timing(mskf_ctx *ctx, double *ms, long long *launches, long long *units, int reset);

int mskf_stream_create(mskf_ctx *ctx, const mskf_calib *calib, const mskf_fe_cfg *fe, const mskf_ekf_cfg *ekf,
                       mskf_stream **out);
void mskf_stream_destroy(mskf_stream *s);
mskf_ctx *mskf_stream_ctx(mskf_stream *s);
/* Run the EKF half of a stream (all mskf_ekf_* calls) on another context of the same GPU, so a front-end
 * thread and a filter thread can drive one VIO stream concurrently (they share no device buffers). */
int mskf_stream_set_ekf_ctx(mskf_stream *s, mskf_ctx *ekf_ctx);
mskf_ctx *mskf_stream_ekf_ctx(mskf_stream *s);
/* Move a stream's front-end half and / or its filter half to other contexts of the same GPU (NULL = stays where it is) WITHOUT
 * synchronising anything: for a scheduler that hands whole batches of streams to whichever worker (context + host thread) is
 * free.  The caller guarantees that the half being moved has no work in flight that the new context's queue is not ordered
 * behind: it has waited for the half's last batch (every *_end call does), or it chains the queues with
 * mskf_ctx_record_point / mskf_ctx_wait_point. */
int mskf_stream_rebind(mskf_stream *s, mskf_ctx *fe_ctx, mskf_ctx *ekf_ctx);
/* Device-side ordering between two contexts' queues: record a point in `ctx`'s stream (the handle is created on first use
 * and reused), make another context's stream wait for it.  No host wait on either side. */
typedef struct mskf_point mskf_point;
int mskf_ctx_record_point(mskf_ctx *ctx, mskf_point **point);
int mskf_ctx_wait_point(mskf_ctx *ctx, mskf_point *point);
void mskf_point_destroy(mskf_point *point);

/* ------------------------------------------------------------------ front-end
 * Replaces, inside cg::ImageProcessor::stereoCallback (image_processor.cpp:139-203):
 *   createImagePyramids()              :213-245  -> mskf_fe_push_stereo*   (pyramids of cam0/cam1 in HBM)
 *   detector_.detect_features()        :259,:657 -> mskf_fe_push_stereo*   (per-cell maxima) + mskf_fe_get_cell_maxima
 *   predictFeatureTracking + optical_flow_multi_level + stereoMatch  :389-463 -> mskf_fe_track (do_temporal = 1)
 *   stereoMatch of new candidates      :268,:688 -> mskf_fe_track (do_temporal = 0)
 *   std::swap(prev pyramid, curr pyramid) :194   -> mskf_fe_swap
 */

/* Upload one stereo pair (host memory, row pitch in bytes), build the 4-level pyramids of both
 * cameras and the detector's per-cell maxima of cam0 level 0.  Asynchronous on the context stream. */
int mskf_fe_push_stereo(mskf_stream *s, const uint8_t *cam0, const uint8_t *cam1, int width, int height, int pitch,
                        double time_stamp);
/* Same with the images already resident in device memory (dense, pitch == width). */
int mskf_fe_push_stereo_device(mskf_stream *s, const uint8_t *d_cam0, const uint8_t *d_cam1, int width, int height,
                               double time_stamp);
/* on_device: 0 = host images (copied in), 1 = device images (copied D2D), 2 = device images BORROWED: read in
 * place, the caller keeps each pair valid and unchanged until the second-next push of that stream.
 * (3 is internal to mskf_fe_push_stereo: level 0 already copied into the stream's own planes.) */
int mskf_fe_push_stereo_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const uint8_t *const *cam0,
                              const uint8_t *const *cam1, int on_device);

/* Detector floor of a stream, in score units (1/256 of the response): from the next push on only per-cell maxima ABOVE the
 * floor are recorded (default 0: every cell with a positive score).  A caller that only ever asks for the candidates above
 * its detector threshold (image_processor.cpp:132: fast_threshold) sets the floor to that threshold: the comparison is made
 * exactly, before the integer square root of the Shi-Tomasi score, which most pixels then never need. */
int mskf_fe_set_detect_floor(mskf_stream *s, int min_score);
/* All det_rows*det_cols per-cell maxima of the last pushed cam0 image (score 0 = no corner above the floor). Synchronises. */
int mskf_fe_get_cell_maxima(mskf_stream *s, mskf_corner *out, int capacity, int *n_out);
/* Same, restricted to the cells whose maximum score exceeds min_score (the detector threshold of
 * image_processor.cpp:132, in 1/256 units), in cell order; out[k].cell identifies the cell.  This is what
 * detect_features() needs — converting and copying the ~1400 sub-threshold cells of a 752x480 frame was a
 * measurable share of the host time per frame.  min_score must not be below the stream's detector floor. */
int mskf_fe_get_cell_candidates(mskf_stream *s, int min_score, mskf_corner *out, int capacity, int *n_out);

typedef struct mskf_fe_track_args {
    int32_t n;                    /* number of points */
    int32_t do_temporal;          /* 1: prev cam0 -> curr cam0 LK, then stereo; 0: stereo only */
    const mskf_point2f *in_pts;   /* prev cam0 points (temporal) or curr cam0 points (stereo only) */
    double Hpred[9];              /* K R_p_c K^-1 (image_processor.cpp:335-340); identity-like under Q2 */
    mskf_point2f *out0, *out1;    /* tracked cam0 / matched cam1 pixels */
    mskf_point2f *und0, *und1;    /* undistorted normalised coordinates of out0 / out1 */
    uint8_t *status;              /* bit0 temporal ok, bit1 stereo inlier */
} mskf_fe_track_args;

int mskf_fe_track(mskf_stream *s, const mskf_fe_track_args *args);
int mskf_fe_track_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const mskf_fe_track_args *args);
/* The same in two halves, so that a host thread can prepare another batch (on another context, possibly sharing this
 * one's HIP stream: mskf_ctx_create_shared) while the device works: _begin validates, stages and enqueues everything and
 * returns; _end waits and copies the results into the args passed to _begin (they and their arrays must still be valid).
 * One pending batch per context. */
int mskf_fe_track_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, const mskf_fe_track_args *args);
int mskf_fe_track_batch_end(mskf_ctx *ctx);

/* curr cam0 pyramid becomes the prev pyramid (image_processor.cpp:194) */
int mskf_fe_swap(mskf_stream *s);

/* ---- a whole front-end frame on the device
 * Replaces, for every frame after the first, ALL of cg::ImageProcessor::stereoCallback between the image copy and
 * publish() (image_processor.cpp:148-200): createImagePyramids, trackFeatures (:352-532), addNewFeatures (:622-756),
 * pruneGridFeatures (:758-768) and the state rotation (:192-200).  The live grid stays in device memory from frame to
 * frame; one call enqueues pyramids + detector, the temporal and stereo track of the previous features, the bookkeeping
 * between (survivors, occupancy, detections, per-cell sieve, candidates), the candidates' stereo track and the bookkeeping
 * after it (vacancy fill, ids, pruning), and returns the published grid: ids, lifetimes, pixels and the undistorted
 * points publish() writes into the message (:1137-1182).  One host wait per frame instead of two, no points travel to
 * the device.  With MSKF_COMPAT_Q5_NO_RANSAC cleared the 2-point RANSAC of :482-500 / :911-1135 runs inside the call too,
 * between the track of the previous features and the bookkeeping (its draws come from a counter the stream keeps on the
 * device).  Limits: grid_min / grid_max_feature_num <= 16 and bookkeeping lists that fit the kernel's LDS
 * (mskf_fe_grid_capacity returns 0 otherwise); the caller keeps such streams, and the first frame of every stream, on the
 * mskf_fe_track path and hands the grid over with mskf_fe_set_grid. */
typedef struct mskf_fe_frame_args {
    double Hpred[9];              /* in: K R_p_c K^-1 of this frame (image_processor.cpp:335-340) */
    int32_t capacity;             /* in: entries each output array holds, >= mskf_fe_grid_capacity(stream) */
    int32_t n;                    /* out: features of the published grid, in flatten order (ascending grid code) */
    uint64_t *id;                 /* out */
    int32_t *lifetime;            /* out */
    mskf_point2f *cam0, *cam1;    /* out: pixels */
    mskf_point2f *und0, *und1;    /* out: undistorted normalised coordinates */
    int32_t before_tracking, after_tracking, after_matching, after_ransac;   /* out: TrackingInfo (:514-530) */
    int32_t n_candidates, n_new;  /* out: candidates stereo-matched for the vacancies, features created */
    uint64_t next_feature_id;     /* out: the stream's id counter after this frame */
    double R_p_c[2][9];           /* in: rotation previous -> current frame of cam0 and of cam1 (integrateImuData, :850-889); read by the
                                   *     2-point RANSAC only (MSKF_COMPAT_Q5_NO_RANSAC cleared) */
    uint64_t ransac_draws;        /* out: numbers the stream's RANSAC generator has drawn so far (a host-side frame continues from it) */
} mskf_fe_frame_args;
/* entries a published grid can have: (grid codes) x grid_max_feature_num; 0 = this stream keeps its books on the host */
int mskf_fe_grid_capacity(mskf_stream *s);
/* Hand the device the grid a host-side frame has published (the first frame of a stream), with the id counter, the
 * tracking counters that survive frames without features (:383) and the draw counter of the RANSAC generator.  Synchronous. */
int mskf_fe_set_grid(mskf_stream *s, int n, const uint64_t *id, const int32_t *lifetime, const mskf_point2f *cam0, const mskf_point2f *cam1,
                     const mskf_point2f *und0, const mskf_point2f *und1, uint64_t next_feature_id, const int32_t tracking_counters[3],
                     uint64_t ransac_draws);
/* streams / args must stay valid until _end; the pyramid swap of :194 is part of the call */
int mskf_fe_frame_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, const uint8_t *const *cam0, const uint8_t *const *cam1,
                              int on_device, mskf_fe_frame_args *args);
int mskf_fe_frame_batch_end(mskf_ctx *ctx);

/* download pyramid level `level` (0..3) of image role 0: prev cam0, 1: curr cam0, 2: curr cam1 (parity tests) */
int mskf_fe_get_level(mskf_stream *s, int role, int level, uint8_t *out, int capacity, int *w, int *h);

/* ------------------------------------------------------------------ EKF (covariance resident in HBM)
 * Replaces, inside cg::MsckfVio (msckf_vio.cpp):
 *   state_cov part of processModel     :458-469  -> mskf_ekf_propagate
 *   state_cov part of stateAugmentation :564-582 -> mskf_ekf_augment
 *   Feature::initializePosition        feature.hpp:289-450 -> inside mskf_ekf_update (needs_init)
 *   featureJacobian / measurementJacobian / gatingTest / stacking / measurementUpdate
 *                                      :610-935, :986-1017, :1124-1159 -> mskf_ekf_update
 *   clone row/column deletion          :1161-1181 -> mskf_ekf_remove_clone
 *   covariance reset                   :102-112, :1222-1232 -> mskf_ekf_reset
 */
typedef struct mskf_clone_state {   /* CAMState, common/cam_state.h:25-55 */
    double q[4], p[3], q_null[4], p_null[3];
} mskf_clone_state;

typedef struct mskf_ekf_feature {   /* one feature of an update */
    int32_t obs_start, n_obs;       /* range in the observation arrays */
    int32_t needs_init;             /* 1: triangulate from ALL its observations listed in init_* first */
    int32_t init_start, n_init;     /* observation range used for triangulation (all of the feature's obs) */
    int32_t _pad;
    double position[3];             /* in: world position if !needs_init; out: triangulated position */
} mskf_ekf_feature;

typedef struct mskf_ekf_update_args {
    int32_t n_clones;
    int32_t n_feat;
    int32_t n_obs;
    int32_t dof_offset;             /* gating dof = n_obs_j + dof_offset: -1 lost features, 0 pruning (Q12) */
    int32_t apply_row_cap;          /* 1: stop stacking once rows > max_stack_rows (removeLostFeatures only, Q13) */
    int32_t _pad;
    double gravity[3];
    const mskf_clone_state *clones; /* n_clones, in state order */
    mskf_ekf_feature *features;     /* in/out: position, and status in `feat_status` */
    const int32_t *obs_clone;       /* n_obs: clone index of each observation */
    const double *obs_z;            /* n_obs x 4: u0 v0 u1 v1 */
    double *delta_x;                /* out: 21 + 6 n_clones */
    uint8_t *feat_status;           /* out per feature: bit0 triangulation valid, bit1 gating passed (stacked) */
    double *gamma;                  /* out per feature (may be NULL): Mahalanobis gate value */
    int32_t *rows_out;              /* out: number of stacked rows (0 => no update applied) */
    int32_t *diag_out;              /* out, optional (may be NULL), 2 ints: [0] how the stack was compressed: 0 Gram + Cholesky,
                                       1 Householder TSQR, 2 not at all (rows <= active columns, msckf_vio.cpp:818-821);
                                       [1] pivots of the Gram factor below 100 lambda (-1: not computed) */
    double *pos_var_out;            /* out, optional (may be NULL), 3 doubles: P(12,12), P(13,13), P(14,14) AFTER this update, i.e.
                                       what onlineReset tests (msckf_vio.cpp:1194-1196) — they come back with the update's
                                       results, so a caller that updates every frame never needs mskf_ekf_get_pos_var and its
                                       extra wait (clone removal does not touch these entries).  -1 when no stream of the batch
                                       had features (nothing was launched).  ABI v3 */
} mskf_ekf_update_args;

/* What processModel (msckf_vio.cpp:409-469) needs to propagate the covariance over one IMU sample.
 * The host integrates the 16-dim nominal state (predictNewState, :482-531) and hands over these
 * quantities; F, Phi (3rd-order expm + observability fix-up) and Q are formed on the device. */
typedef struct mskf_imu_step {
    double dt;
    double gyro[3], acc[3];   /* bias-corrected measurements (:412-413) */
    double R_t[9];            /* R(q)^T before the step (:422-423) */
    double Phi00[9];          /* R(q_new) R(q_null)^T (:443) */
    double u[3], s[3];        /* u = R(q_null) g, s = u / (u.u) (:445-446) */
    double w1[3], w2[3];      /* [v_null - v_new]x g (:450), [dt v_null + p_null - p_new]x g (:454) */
} mskf_imu_step;

int mskf_ekf_reset(mskf_stream *s, const double *P0 /* 21x21 row-major */);
int mskf_ekf_propagate_imu(mskf_stream *s, int n_steps, const mskf_imu_step *steps);
/* One fused launch for many streams: IMU propagation over n_steps[i] samples followed (J[i] != NULL) by the
 * state augmentation.  Equivalent to mskf_ekf_propagate_imu + mskf_ekf_augment per stream. */
int mskf_ekf_predict_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const int32_t *n_steps,
                           const mskf_imu_step *const *steps, const double *const *J);
/* position variances P(12,12), P(13,13), P(14,14) for onlineReset (msckf_vio.cpp:1194-1196). Synchronises. */
int mskf_ekf_get_pos_var(mskf_stream *s, double out[3]);
int mskf_ekf_get_pos_var_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, double *out /* 3n */);
int mskf_ekf_get_pos_var_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, double *out /* 3n, filled by _end */);
int mskf_ekf_get_pos_var_batch_end(mskf_ctx *ctx);
int mskf_ekf_propagate(mskf_stream *s, int n_steps, const double *Phi /* n x 21x21 */, const double *Q /* n x 21x21 */);
int mskf_ekf_augment(mskf_stream *s, const double *J /* 6x21 */);
int mskf_ekf_update(mskf_stream *s, mskf_ekf_update_args *args);
int mskf_ekf_update_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, mskf_ekf_update_args *args);
/* begin / end halves as for mskf_fe_track_batch; `streams` and `args` must stay valid until _end */
int mskf_ekf_update_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, mskf_ekf_update_args *args);
int mskf_ekf_update_batch_end(mskf_ctx *ctx);
int mskf_ekf_remove_clone(mskf_stream *s, int clone_index);
/* Remove up to two clones per stream in one launch: idx[2*i], idx[2*i+1] are clone indices in the CURRENT
 * state order (distinct; -1 = none).  Equivalent to mskf_ekf_remove_clone calls (msckf_vio.cpp:1161-1181). */
int mskf_ekf_remove_clones_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const int32_t *idx);
/* Diagnostics for tests: raw work buffers of the stream's last update (0 Hs, 1 rowmask, 2 S, 3 T, 4 W, 5 act). */
int mskf_ekf_debug_read(mskf_stream *s, int which, void *out, size_t capacity, int *ld_out);
/* Change mskf_ekf_cfg.compression_mode of a live stream (0 auto, 1 Gram + Cholesky only, 2 Householder TSQR always, 3 the rule
 * of msckf_vio.cpp:795-821 as written); takes effect with the stream's next update.  Not while an update of the stream is pending. */
int mskf_ekf_set_compression_mode(mskf_stream *s, int mode);
int mskf_ekf_get_dim(mskf_stream *s, int *d);
int mskf_ekf_get_cov(mskf_stream *s, double *P, int capacity /* doubles */);
int mskf_ekf_set_cov(mskf_stream *s, const double *P, int d);

#ifdef __cplusplus
}
#endif
#endif /* MSKF_HIP_H */
