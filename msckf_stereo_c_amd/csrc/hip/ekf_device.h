// ekf_device.h — device-side descriptors of the MSCKF measurement-update kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/mskf_hip.h"

#define EKF_IMU_DIM 21
#define EKF_SLOTS 64          // feature workgroups per stream and launch (each loops over its features)

// Per feature of one update (device copy of mskf_ekf_feature + row offset of its block)
struct EkfFeatDev {
    int obs_start, n_obs;
    int needs_init, init_start, n_init;
    int row_off;              // first row of this feature's (4 n_obs - 3)-row block in Hs
    double position[3];
    unsigned long long colmask;   // device: bit c set = the block carries clone c's six columns and is stacked (0 = not stacked)
};

// Everything the update kernels need for one VIO stream.
struct EkfStreamDev {
    double *P;                // ld x ld, row-major, only [0,d) x [0,d) is live
    int d, ld;
    int n_clones, n_feat, n_obs;
    int dof_offset, apply_row_cap, max_stack_rows;
    int m_total;              // sum of block rows of all features
    double sigma2;            // Feature::observation_noise (variance)
    double gravity[3];
    double R_c0_c1[9], t_c0_c1[3];       // CAMState::T_cam0_cam1
    const double *chi2;                  // table[100], index = dof
    const mskf_clone_state *clones;      // n_clones
    EkfFeatDev *feats;                   // n_feat
    const int *obs_clone;                // n_obs
    const double *obs_z;                 // n_obs x 4
    // m_total x ld stacked (null-space projected) Jacobian.  A block is written only in the columns of the clones its
    // feature observed (+ column d, the residual); what a row carries is its rowmask (clone bits, 0 = the row is not
    // stacked: failed triangulation / gate, or behind the row cap).  Nothing else of a row is ever written or read.
    double *Hs;
    unsigned long long *rowmask;   // m_total, written by the feature kernels for their own rows
    // The dense update works on the active columns only (compact index i <-> column act[i]): a stacked Jacobian is
    // identically zero in the 21 IMU columns and in the columns of clones none of its features observed, and such
    // columns contribute nothing to S, K or the covariance downdate.
    double *T;                // na x (d+1) work: T = R P[act, :], then Y = L^-1 [T | Q^T r]
    double *S;                // (na+1)^2 work (compact): Gram matrix [H_act|r]^T [H_act|r], then its Cholesky factor L = R^T (+ the Q^T r row)
    double *W;                // na x na work (compact): S = T[:, act] R^T + sigma^2 I, then its Cholesky factor
    int *act;                 // active columns of this update (the 6 columns of every clone a stacked feature observed), ascending; count in rows_out[2];
                              // act + ld: indices of the stacked rows, in order, when the update is not compressed (2 ld ints in all)
    double *gate_S;           // EKF_SLOTS x (nmax x nmax)
    int nmax;                 // 4 * max_clones
    double *delta_x;          // d
    uint8_t *feat_status;     // n_feat: bit0 triangulation valid, bit1 gate passed and stacked
    double *gamma;            // n_feat
    double *pos_out;          // n_feat x 3: feature positions used (triangulated when needs_init)
    double *pos_var_out;      // 3: P(12,12), P(13,13), P(14,14) after the update (epilogue of k_ekf_gemm<PUPD>), or null
    int route;                // which kernels handle this stream's update, decided per STREAM from its own features (never from
                              // the rest of the batch, so a stream's arithmetic does not depend on its neighbours): bit 0 pair
                              // kernels (every feature has exactly the same two Jacobian clones), bit 1 wave-per-feature class
                              // (every feature <= 4 observations), bit 2 fused small update (at most 4 clones touched)
    int na_max;               // 6 x the clones any feature of this update observed: an upper bound of the active columns (from the host)
    int qr_mode;              // mskf_ekf_cfg.compression_mode: 0 auto (Gram + Cholesky, Householder TSQR when flagged), 1 Gram only, 2 TSQR always,
                              // 3 the reference's own rule (Householder when rows > columns, uncompressed otherwise)
    const int *tri_idx;       // features that need triangulation (pair path: k_ekf_triangulate), n_tri of them
    int n_tri;
    int *rows_out;            // [0] stacked rows, [1] last stacked row + 1 (K range of the Gram pass), [2] number of active columns,
                              // [3] compression diagnostics: bit 0 = Householder TSQR used, bit 1 = the lambda prior of the Gram path
                              //     would bias P by more than QR_BIAS_LIMIT (auto mode then re-does the compression as TSQR), bit 2 = no
                              //     compression (stacked rows <= active columns: the stacked rows themselves are the measurement),
                              //     bits 8.. = pivots of the Gram factor below 100 lambda (reported only)
                              // [4] nk = rows of the compressed measurement = dimension of S: na, or rows_out[0] without compression
    // propagation / augmentation
    const double *PhiQ;       // n_steps x (2 x 21 x 21): Phi then Q   (generic form)
    const mskf_imu_step *imu_steps;   // or: n_steps compact IMU records, Phi/Q formed on the device
    double qc[4];             // continuous noise variances: gyro, gyro bias, acc, acc bias (msckf_vio.cpp:174-178)
    int n_steps;
    const double *J;          // 6 x 21
    int remove_index;         // clone to delete (-1 = none)
    int remove_index2;        // second clone to delete, > remove_index (-1 = none)
    double *P_dst;            // destination of the out-of-place clone removal
};

struct EkfStreamState {       // host-side bookkeeping of the device buffers of one stream
    int max_clones = 0, ld = 0, d = EKF_IMU_DIM;
    int max_rows = 0, max_feat = 0, max_obs = 0, nmax = 0;
    bool hs_async = false;    // Hs came from the stream-ordered allocator (grown during a run)
    double *pool = nullptr;   // one allocation: P, T, S, W, P_alt, act, gate_S, chi2
    double *P = nullptr, *Hs = nullptr, *rs = nullptr, *T = nullptr, *S = nullptr, *W = nullptr, *gate_S = nullptr;
    int *act = nullptr;       // ld ints: active column list of the current update
    double *chi2 = nullptr;
    // per-update staging: one pinned+device arena, laid out by the host
    char *h_arena = nullptr, *d_arena = nullptr;
    size_t arena_bytes = 0;
    char *h_out = nullptr, *d_out = nullptr;   // results: delta_x, status, gamma, rows, positions
    size_t out_bytes = 0;
};
