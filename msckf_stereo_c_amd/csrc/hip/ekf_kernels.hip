// ekf_kernels.hip — hand-written gfx950 kernels of the MSCKF measurement update.
//
//  k_ekf_propagate      : covariance part of processModel        (msckf_vio.cpp:458-469)
//  k_ekf_augment        : covariance part of stateAugmentation   (:564-582)
//  k_ekf_remove_clone   : clone row/column deletion              (:1161-1181)
//  k_ekf_feature_blocks : Feature::initializePosition (feature.hpp:289-450), measurementJacobian
//                         (:610-677), featureJacobian null-space projection (:679-775), gatingTest (:909-935)
//  (ekf_cap.h)          : stacking order + 1500-row cap          (:1003-1010), run by the first dense kernel of the update
//  (ekf_linalg.hip)     : QR compression (:795-811) as Gram + Cholesky, gain / correction / covariance update (:831-904)
//
// All arithmetic is FP64.  The covariance P stays resident in HBM (ld x ld, row-major, exactly
// symmetric by construction); one workgroup handles one VIO stream (or one feature of one
// stream), blockIdx.y is the stream of the batch.
#include <mutex>
#include <type_traits>
#include "ekf_device.h"
#include "chol_block.h"

#define WG 256

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide sum, result broadcast to all threads (s_red: >= blockDim/64 doubles + 1)
__device__ __forceinline__ double block_sum(double v, double *s_red) {
    v = wave_sum(v);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < nw; ++i) t += s_red[i];
    return t;
}

// ------------------------------------------------------------------------------------ small math
__device__ __forceinline__ void quat_to_rot(const double *q, double *R) {
    // JPL: R = (2w^2-1) I - 2w [qv]x + 2 qv qv^T   
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double a = 2 * w * w - 1, tw = 2 * w;
    R[0] = a + 2 * x * x;        R[1] = tw * z + 2 * x * y;   R[2] = -tw * y + 2 * x * z;
    R[3] = -tw * z + 2 * y * x;  R[4] = a + 2 * y * y;        R[5] = tw * x + 2 * y * z;
    R[6] = tw * y + 2 * z * x;   R[7] = -tw * x + 2 * z * y;  R[8] = a + 2 * z * z;
}
__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_vec(const double *A, const double *v, double *o) {
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void mat3t_vec(const double *A, const double *v, double *o) {
    for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

// Phi (3rd-order expm of F dt with the observability fix-ups) and Q = Phi G Qc G^T Phi^T dt of one IMU step
// (processModel, msckf_vio.cpp:417-458), by all WG threads.  sA: scratch (F dt), sB: scratch, receives Q.
__device__ __forceinline__ void build_phi_q(const mskf_imu_step &st, const double *qc, double *sPhi, double *sA, double *sB) {
    constexpr int N = EKF_IMU_DIM;
    const int tid = threadIdx.x;
    // F dt, Phi = I + Fdt + Fdt^2/2 + Fdt^3/6, fix-ups, Q
    for (int i = tid; i < N * N; i += WG) sA[i] = 0.0;
    __syncthreads();
    if (tid < 9) {
        const int r = tid / 3, c = tid % 3;
        const double g[3] = {st.gyro[0], st.gyro[1], st.gyro[2]}, a[3] = {st.acc[0], st.acc[1], st.acc[2]};
        const double skg[9] = {0, -g[2], g[1], g[2], 0, -g[0], -g[1], g[0], 0};
        const double ska[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
        double rs = 0;
        for (int k = 0; k < 3; ++k) rs += st.R_t[3 * r + k] * ska[3 * k + c];
        sA[r * N + c] = -skg[3 * r + c] * st.dt;                     // F(0,0) = -[w]x
        sA[r * N + 3 + c] = (r == c ? -1.0 : 0.0) * st.dt;           // F(0,3) = -I
        sA[(6 + r) * N + c] = -rs * st.dt;                           // F(6,0) = -R^T [a]x
        sA[(6 + r) * N + 9 + c] = -st.R_t[3 * r + c] * st.dt;        // F(6,9) = -R^T
        sA[(12 + r) * N + 6 + c] = (r == c ? 1.0 : 0.0) * st.dt;     // F(12,6) = I
    }
    __syncthreads();
    for (int i = tid; i < N * N; i += WG) {
        const int r = i / N, c = i % N;
        double s2 = 0;
        for (int k = 0; k < N; ++k) s2 += sA[r * N + k] * sA[k * N + c];
        sB[i] = s2;
    }
    __syncthreads();
    for (int i = tid; i < N * N; i += WG) {
        const int r = i / N, c = i % N;
        double s3 = 0;
        for (int k = 0; k < N; ++k) s3 += sB[r * N + k] * sA[k * N + c];
        sPhi[i] = (r == c ? 1.0 : 0.0) + sA[i] + 0.5 * sB[i] + (1.0 / 6.0) * s3;
    }
    __syncthreads();
    if (tid < 9) {
        const int r = tid / 3, c = tid % 3;
        // A - (A u - w) s^T for the velocity and position rows (:449-455); uses the un-fixed A
        double a1u = 0, a2u = 0;
        for (int k = 0; k < 3; ++k) { a1u += sPhi[(6 + r) * N + k] * st.u[k]; a2u += sPhi[(12 + r) * N + k] * st.u[k]; }
        const double v1 = sPhi[(6 + r) * N + c] - (a1u - st.w1[r]) * st.s[c];
        const double v2 = sPhi[(12 + r) * N + c] - (a2u - st.w2[r]) * st.s[c];
        __builtin_amdgcn_wave_barrier();
        sPhi[(6 + r) * N + c] = v1;
        sPhi[(12 + r) * N + c] = v2;
        sPhi[r * N + c] = st.Phi00[3 * r + c];
    }
    __syncthreads();
    for (int i = tid; i < N * N; i += WG) {
        const int r = i / N, c = i % N;
        double q = 0;
        for (int k = 0; k < 12; ++k) q += sPhi[r * N + k] * qc[k / 3] * sPhi[c * N + k];
        sB[i] = q * st.dt;
    }
    __syncthreads();
    }

// ------------------------------------------------------------------------------------ propagate (+ augment)
// Per IMU step: Phi, Q (build_phi_q; rounds 1-3 built them in a launch of its own, one workgroup per step: one more link in the
// frame's launch chain, 360 us in the busy device for 10 us of work), P_II <- sym(Phi P_II Phi^T + Q) (in LDS); the clone cross terms are propagated once with the
// composed transition  P_IC <- (Phi_n ... Phi_1) P_IC  (one pass over P instead of one per IMU sample),
// P_CI <- P_IC^T.  With S.J set the state augmentation (rows/cols [d, d+6) = J [P_II P_IC], corner
// sym(J P_II J^T)) is fused into the same pass.
__global__ __launch_bounds__(WG) void k_ekf_propagate(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_steps <= 0 && !S.J) return;
    double *P = S.P;
    const int d = S.d, ld = S.ld, N = EKF_IMU_DIM;
    __shared__ double sPhi[N * N], sQ[N * N], sP[N * N], sT[N * N], sTot[N * N], sJ[6 * N], sC[6 * N];
    const int tid = threadIdx.x;
    for (int i = tid; i < N * N; i += WG) { sP[i] = P[(size_t)(i / N) * ld + (i % N)]; sTot[i] = (i / N == i % N) ? 1.0 : 0.0; }
    for (int step = 0; step < S.n_steps; ++step) {
        __syncthreads();
        if (S.imu_steps && !S.PhiQ) {
            build_phi_q(S.imu_steps[step], S.qc, sPhi, sT, sQ);       // scratch sT, Q lands in sQ
        } else {
            const double *PhiQ = S.PhiQ + (size_t)step * 2 * N * N;
            for (int i = tid; i < N * N; i += WG) { sPhi[i] = PhiQ[i]; sQ[i] = PhiQ[N * N + i]; }
        }
        __syncthreads();
        double t0 = 0, t1 = 0;   // Phi * Tot, elements tid and tid + WG
        for (int i = tid; i < N * N; i += WG) {
            const int r = i / N, c = i % N;
            double s = 0, st2 = 0;
            for (int k = 0; k < N; ++k) { s += sPhi[r * N + k] * sP[k * N + c]; st2 += sPhi[r * N + k] * sTot[k * N + c]; }
            sT[i] = s;
            if (i < WG) t0 = st2; else t1 = st2;
        }
        __syncthreads();
        double x0 = 0, x1 = 0;  // each thread owns elements tid and tid+WG (N*N = 441 <= 512)
        for (int e = 0; e < 2; ++e) {
            const int i = tid + e * WG;
            if (i < N * N) {
                const int r = i / N, c = i % N;
                double s = 0, st = 0;
                for (int k = 0; k < N; ++k) { s += sT[r * N + k] * sPhi[c * N + k]; st += sT[c * N + k] * sPhi[r * N + k]; }
                const double v = ((s + sQ[r * N + c]) + (st + sQ[c * N + r])) * 0.5;
                if (e == 0) x0 = v; else x1 = v;
            }
        }
        __syncthreads();
        if (tid < N * N) { sP[tid] = x0; sTot[tid] = t0; }
        if (tid + WG < N * N) { sP[tid + WG] = x1; sTot[tid + WG] = t1; }
    }
    __syncthreads();
    if (S.J) for (int i = tid; i < 6 * N; i += WG) sJ[i] = S.J[i];
    __syncthreads();
    // one pass over the clone columns: P_IC <- Tot P_IC (and its mirror), augmentation rows from the new column
    for (int c = tid; c < d; c += WG) {
        double col[N];
        if (c < N) { for (int k = 0; k < N; ++k) col[k] = sP[k * N + c]; }
        else {
            double old[N];
            for (int k = 0; k < N; ++k) old[k] = P[(size_t)k * ld + c];
            for (int r = 0; r < N; ++r) {
                double s = 0;
                for (int k = 0; k < N; ++k) s += sTot[r * N + k] * old[k];
                col[r] = s;
            }
            if (S.n_steps > 0) for (int r = 0; r < N; ++r) { P[(size_t)r * ld + c] = col[r]; P[(size_t)c * ld + r] = col[r]; }
        }
        if (S.J) {
            for (int r = 0; r < 6; ++r) {
                double s = 0;
                for (int k = 0; k < N; ++k) s += sJ[r * N + k] * col[k];
                P[(size_t)(d + r) * ld + c] = s;
                P[(size_t)c * ld + (d + r)] = s;
                if (c < N) sC[r * N + c] = s;
            }
        }
    }
    for (int i = tid; i < N * N; i += WG) P[(size_t)(i / N) * ld + (i % N)] = sP[i];
    if (S.J) {
        __syncthreads();
        if (tid < 36) {
            const int r = tid / 6, c = tid % 6;
            double s = 0, st = 0;
            for (int k = 0; k < N; ++k) { s += sC[r * N + k] * sJ[c * N + k]; st += sC[c * N + k] * sJ[r * N + k]; }
            P[(size_t)(d + r) * ld + (d + c)] = (s + st) / 2.0;
        }
    }
}

// ------------------------------------------------------------------------------------ augment
// rows/cols [d, d+6): [J P11, J P12], corner sym(J P11 J^T)
__global__ __launch_bounds__(WG) void k_ekf_augment(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (!S.J) return;
    double *P = S.P;
    const int d = S.d, ld = S.ld, N = EKF_IMU_DIM;  // d = dimension BEFORE augmentation
    __shared__ double sJ[6 * N], sC[6 * N];
    const int tid = threadIdx.x;
    for (int i = tid; i < 6 * N; i += WG) sJ[i] = S.J[i];
    __syncthreads();
    for (int c = tid; c < d; c += WG) {
        double col[N];
        for (int k = 0; k < N; ++k) col[k] = P[(size_t)k * ld + c];
        for (int r = 0; r < 6; ++r) {
            double s = 0;
            for (int k = 0; k < N; ++k) s += sJ[r * N + k] * col[k];
            P[(size_t)(d + r) * ld + c] = s;
            P[(size_t)c * ld + (d + r)] = s;
            if (c < N) sC[r * N + c] = s;
        }
    }
    __syncthreads();
    if (tid < 36) {
        const int r = tid / 6, c = tid % 6;
        double s = 0, st = 0;
        for (int k = 0; k < N; ++k) { s += sC[r * N + k] * sJ[c * N + k]; st += sC[c * N + k] * sJ[r * N + k]; }
        P[(size_t)(d + r) * ld + (d + c)] = (s + st) / 2.0;
    }
}

// ------------------------------------------------------------------------------------ remove clone
// out-of-place: P_dst <- P with the rows/cols of one or two clones removed
__global__ __launch_bounds__(WG) void k_ekf_remove_clone(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.remove_index < 0 || !S.P_dst) return;
    const int d = S.d, ld = S.ld;
    const int s0 = EKF_IMU_DIM + 6 * S.remove_index;
    const int s1 = S.remove_index2 >= 0 ? EKF_IMU_DIM + 6 * S.remove_index2 : (1 << 30);
    const int dn = d - (S.remove_index2 >= 0 ? 12 : 6);
    for (int idx = blockIdx.x * WG + threadIdx.x; idx < dn * dn; idx += gridDim.x * WG) {
        const int i = idx / dn, j = idx - i * dn;
        int si = i < s0 ? i : i + 6; if (si >= s1) si += 6;
        int sj = j < s0 ? j : j + 6; if (sj >= s1) sj += 6;
        S.P_dst[(size_t)i * ld + j] = S.P[(size_t)si * ld + sj];
    }
}

// ------------------------------------------------------------------------------------ position variances
// P(12,12), P(13,13), P(14,14) of every stream of the batch (onlineReset, msckf_vio.cpp:1194-1196)
__global__ void k_ekf_posvar(const EkfStreamDev *streams, int n, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const EkfStreamDev &S = streams[i / 3];
    const int k = 12 + i % 3;
    out[i] = S.P[(size_t)k * S.ld + k];
}

// the same for the streams of an update batch that have NO features (nothing of the update chain runs for them), into the
// stream's own result slot (mskf_ekf_update_args.pos_var_out)
__global__ void k_ekf_posvar_upd(const EkfStreamDev *streams, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const EkfStreamDev &S = streams[i / 3];
    if (!S.pos_var_out || S.n_feat > 0) return;        // (a stream with an update got them from k_ekf_gemm<PUPD>'s epilogue)
    const int k = 12 + i % 3;
    S.pos_var_out[i % 3] = S.P[(size_t)k * S.ld + k];
}

// ------------------------------------------------------------------------------------ feature blocks
#define MAX_CLONES_DEV 64          // 4*64 = 256 block rows max per feature

template <int NMEAS>
struct TriScratchT {     // poses and rays of the 2 * n_init stereo measurements of one feature
    double R[NMEAS][9];
    double t[NMEAS][3];
    double z[NMEAS][2];
};
typedef TriScratchT<2 * MAX_CLONES_DEV> TriScratch;
#define TRI_SMALL_CLONES 32       // the wave-per-feature variant triangulates over at most this many clones

// Levenberg-Marquardt inverse-depth triangulation, executed by wave 0 of the workgroup.
// feature.hpp:289-450; parameters feature.hpp:46-52.
template <class TS>
__device__ bool triangulate_wave(const EkfStreamDev &S, const EkfFeatDev &F, TS &ts, double *pos_out) {
    const int lane = threadIdx.x & 63;
    const int n_meas = 2 * F.n_init;
    // first camera pose (camera -> world): R0 = R(q)^T, t0 = p
    double R0[9], t0[3];
    {
        const mskf_clone_state &c = S.clones[S.obs_clone[F.init_start]];
        double Rw[9];
        quat_to_rot(c.q, Rw);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R0[3 * i + j] = Rw[3 * j + i];
        t0[0] = c.p[0]; t0[1] = c.p[1]; t0[2] = c.p[2];
    }
    // T_cam0_cam1^-1 = (R^T, -R^T t)
    double Rinv[9], tinv[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rinv[3 * i + j] = S.R_c0_c1[3 * j + i];
    { double tmp[3]; mat3_vec(Rinv, S.t_c0_c1, tmp); tinv[0] = -tmp[0]; tinv[1] = -tmp[1]; tinv[2] = -tmp[2]; }
    for (int m = lane; m < n_meas; m += 64) {
        const int o = F.init_start + (m >> 1);
        const mskf_clone_state &c = S.clones[S.obs_clone[o]];
        double Rw[9], Rc[9], tc[3];
        quat_to_rot(c.q, Rw);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rc[3 * i + j] = Rw[3 * j + i];  // cam0 -> world
        tc[0] = c.p[0]; tc[1] = c.p[1]; tc[2] = c.p[2];
        if (m & 1) {  // cam1_pose = cam0_pose * T_cam0_cam1^-1
            double R2[9], t2[3];
            mat3_mul(Rc, Rinv, R2);
            mat3_vec(Rc, tinv, t2);
            for (int i = 0; i < 9; ++i) Rc[i] = R2[i];
            for (int i = 0; i < 3; ++i) tc[i] = t2[i] + tc[i];
        }
        // pose <- pose^-1 * T_c0_w : R = Rc^T R0, t = Rc^T (t0 - tc)
        double Rt[9], dt[3] = {t0[0] - tc[0], t0[1] - tc[1], t0[2] - tc[2]};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = Rc[3 * j + i];
        mat3_mul(Rt, R0, ts.R[m]);
        mat3_vec(Rt, dt, ts.t[m]);
        ts.z[m][0] = S.obs_z[4 * o + ((m & 1) ? 2 : 0)];
        ts.z[m][1] = S.obs_z[4 * o + ((m & 1) ? 3 : 1)];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // initial guess (feature.hpp:231-255) from the last pose, first and last measurement
    double sol[3];
    {
        const double *Rl = ts.R[n_meas - 1], *tl = ts.t[n_meas - 1];
        const double z1[2] = {ts.z[0][0], ts.z[0][1]}, z2[2] = {ts.z[n_meas - 1][0], ts.z[n_meas - 1][1]};
        double v[3] = {z1[0], z1[1], 1.0}, m[3];
        mat3_vec(Rl, v, m);
        const double A0 = m[0] - z2[0] * m[2], A1 = m[1] - z2[1] * m[2];
        const double b0 = z2[0] * tl[2] - tl[0], b1 = z2[1] * tl[2] - tl[1];
        const double depth = (A0 * b0 + A1 * b1) / (A0 * A0 + A1 * A1);
        const double p0 = z1[0] * depth, p1 = z1[1] * depth, p2 = depth;
        sol[0] = p0 / p2; sol[1] = p1 / p2; sol[2] = 1.0 / p2;
    }
    auto cost_of = [&](const double *x) {
        double e = 0;
        for (int m = lane; m < n_meas; m += 64) {
            const double *R = ts.R[m], *t = ts.t[m];
            const double h0 = R[0] * x[0] + R[1] * x[1] + R[2] + x[2] * t[0];
            const double h1 = R[3] * x[0] + R[4] * x[1] + R[5] + x[2] * t[1];
            const double h2 = R[6] * x[0] + R[7] * x[1] + R[8] + x[2] * t[2];
            const double d0 = h0 / h2 - ts.z[m][0], d1 = h1 / h2 - ts.z[m][1];
            e += d0 * d0 + d1 * d1;
        }
        return wave_sum(e);
    };
    double lambda = 1e-3;
    int inner = 0, outer = 0;
    bool reduced = false;
    double delta_norm = 0;
    double total_cost = cost_of(sol);
    do {
        double A[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};  // A: 00 01 02 11 12 22
        for (int m = lane; m < n_meas; m += 64) {
            const double *R = ts.R[m], *t = ts.t[m];
            const double h1 = R[0] * sol[0] + R[1] * sol[1] + R[2] + sol[2] * t[0];
            const double h2 = R[3] * sol[0] + R[4] * sol[1] + R[5] + sol[2] * t[1];
            const double h3 = R[6] * sol[0] + R[7] * sol[1] + R[8] + sol[2] * t[2];
            double W[3][3];
            for (int i = 0; i < 3; ++i) { W[i][0] = R[3 * i]; W[i][1] = R[3 * i + 1]; W[i][2] = t[i]; }
            double J0[3], J1[3];
            for (int j = 0; j < 3; ++j) {
                J0[j] = 1 / h3 * W[0][j] - h1 / (h3 * h3) * W[2][j];
                J1[j] = 1 / h3 * W[1][j] - h2 / (h3 * h3) * W[2][j];
            }
            const double r0 = h1 / h3 - ts.z[m][0], r1 = h2 / h3 - ts.z[m][1];
            const double e = sqrt(r0 * r0 + r1 * r1);
            const double w = (e <= 0.01) ? 1.0 : sqrt(2.0 * 0.01 / e);
            const double ws = (w == 1) ? 1.0 : w * w;
            A[0] += ws * (J0[0] * J0[0] + J1[0] * J1[0]); A[1] += ws * (J0[0] * J0[1] + J1[0] * J1[1]);
            A[2] += ws * (J0[0] * J0[2] + J1[0] * J1[2]); A[3] += ws * (J0[1] * J0[1] + J1[1] * J1[1]);
            A[4] += ws * (J0[1] * J0[2] + J1[1] * J1[2]); A[5] += ws * (J0[2] * J0[2] + J1[2] * J1[2]);
            b[0] += ws * (J0[0] * r0 + J1[0] * r1); b[1] += ws * (J0[1] * r0 + J1[1] * r1); b[2] += ws * (J0[2] * r0 + J1[2] * r1);
        }
        for (int i = 0; i < 6; ++i) A[i] = wave_sum(A[i]);
        for (int i = 0; i < 3; ++i) b[i] = wave_sum(b[i]);
        do {
            // Cholesky solve of (A + lambda I) delta = b
            const double a00 = A[0] + lambda, a01 = A[1], a02 = A[2], a11 = A[3] + lambda, a12 = A[4], a22 = A[5] + lambda;
            double delta[3] = {0, 0, 0};
            const double l00 = sqrt(a00);
            const double l10 = a01 / l00, l20 = a02 / l00;
            const double s11 = a11 - l10 * l10;
            const double l11 = sqrt(s11);
            const double l21 = (a12 - l20 * l10) / l11;
            const double s22 = a22 - l20 * l20 - l21 * l21;
            if (a00 > 0 && s11 > 0 && s22 > 0) {
                const double l22 = sqrt(s22);
                const double y0 = b[0] / l00, y1 = (b[1] - l10 * y0) / l11, y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
                delta[2] = y2 / l22;
                delta[1] = (y1 - l21 * delta[2]) / l11;
                delta[0] = (y0 - l10 * delta[1] - l20 * delta[2]) / l00;
            }
            const double ns[3] = {sol[0] - delta[0], sol[1] - delta[1], sol[2] - delta[2]};
            delta_norm = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
            const double new_cost = cost_of(ns);
            if (new_cost < total_cost) {
                reduced = true;
                sol[0] = ns[0]; sol[1] = ns[1]; sol[2] = ns[2];
                total_cost = new_cost;
                lambda = lambda / 10 > 1e-10 ? lambda / 10 : 1e-10;
            } else {
                reduced = false;
                lambda = lambda * 10 < 1e12 ? lambda * 10 : 1e12;
            }
        } while (inner++ < 10 && !reduced);
        inner = 0;
    } while (outer++ < 10 && delta_norm > 5e-7);

    const double fp[3] = {sol[0] / sol[2], sol[1] / sol[2], 1.0 / sol[2]};
    int bad = 0;
    for (int m = lane; m < n_meas; m += 64) {
        const double *R = ts.R[m], *t = ts.t[m];
        const double pz = R[6] * fp[0] + R[7] * fp[1] + R[8] * fp[2] + t[2];
        if (pz <= 0) bad = 1;
    }
    bad = __any(bad);
    double pw[3];
    mat3_vec(R0, fp, pw);
    pos_out[0] = pw[0] + t0[0]; pos_out[1] = pw[1] + t0[1]; pos_out[2] = pw[2] + t0[2];
    return !bad;
}

// LDS: static Hf 6K + Hx 12K + r 2K + V 6K + coef 9K ~ 36K; dynamic arena = max(triangulation scratch,
// lds_rows^2 doubles for the gating matrix M) — M lives in LDS when 4 n_obs <= lds_rows (<= 120 rows = 112.5 KiB),
// otherwise in the per-slot global scratch gate_S.
// Mm is symmetric and kept in packed lower form (row i at i(i+1)/2): 120 rows = 56.7 KiB, so with the
// 32-clone instantiation (19 KiB static) two workgroups share a CU.
#define GATE_LDS_ROWS 120
__device__ __forceinline__ size_t pk(int i, int j) { return (size_t)i * (i + 1) / 2 + j; }   // i >= j
// Work group of a feature: the whole 256-thread workgroup (WAVE = false), or one wavefront (WAVE = true: four
// features per workgroup side by side, wave-level barriers only) for launches whose features all have <= MAXC = 4
// observations in the Jacobian — the pruning update hands over hundreds of 2-observation features per stream, for
// which the workgroup-wide barriers were the whole cost (~140 us per feature, measured).
// Size classes (TPB = threads of the workgroup): most lost features have few observations (84 % have <= 16 at the C2
// shape), and for those every loop below is at most one wavefront wide: a 256-thread workgroup only adds barrier
// latency and, with its LDS sized for the largest feature of the batch, leaves a CU with one or two features in flight.
// The <16, false, 64> instantiation gives such a feature one wavefront in a 64-thread workgroup with <= 30 KiB of LDS
// (five per CU, barriers at wave cost); a launch handles the features whose class max(n_obs, n_init) lies in
// [cls_lo, cls_hi].
// Work list: the grid is flat, one entry of `work` per feature group in flight = stream << 16 | slot << 8 | n_slots: the
// group handles the class members of its stream whose ordinal is congruent to slot modulo n_slots (n_slots =
// min(members, EKF_SLOTS), so the global gate scratch of a slot is never shared).  A (slots x streams) grid sized for
// the stream with the most features launched mostly empty workgroups, each holding the full LDS allocation until it
// found nothing to do: the 2 x 192 large features of a batch took 280 us that way.
#define FEAT_SMALL_CLONES 16
template <int MAXC, bool WAVE, int TPB>
__global__ __launch_bounds__(TPB, 2) void k_ekf_feature_blocks(const EkfStreamDev *streams, const int *work, int n_work, int lds_rows, int arena_doubles, int cls_lo, int cls_hi) {
    constexpr int GS = WAVE ? 64 : TPB;         // threads per feature
    constexpr int NSUB = TPB / GS;              // features side by side in a workgroup
    const int gt = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int sub = WAVE ? (int)(threadIdx.x >> 6) : 0;
    const int item = (int)blockIdx.x * NSUB + sub;
    if (item >= n_work) return;                 // (WAVE: a whole wavefront; it shares no barrier with the others)
    const int wcode = work[item];
    const int w_slot = (wcode >> 8) & 0xff, w_nslots = wcode & 0xff;
    const EkfStreamDev &S = streams[wcode >> 16];
    const int d = S.d, ld = S.ld;
    // Jacobian / reflector scratch, one block per feature in flight: it is dead once the gate matrix is reflected,
    // and the 16-wide Cholesky panel (LNB x PAN_RS doubles) of the workgroup variant reuses it
    constexpr int BLK_DOUBLES = (12 + 24 + 4 + 12) * MAXC + 3 * (6 * MAXC + 1);
    constexpr int PAN_RS = WAVE ? 16 : ((4 * MAXC + 15) / 16) * 16;      // k-major Cholesky panel stride: rows rounded up to 16
    static_assert(WAVE || BLK_DOUBLES >= LNB * PAN_RS, "panel does not fit the dead scratch");
    __shared__ double s_blk_all[NSUB][BLK_DOUBLES];
    double *s_blk = s_blk_all[sub];
    double (*sHf)[3] = reinterpret_cast<double (*)[3]>(s_blk);
    double (*sHx)[6] = reinterpret_cast<double (*)[6]>(s_blk + 12 * MAXC);
    double *sr = s_blk + 36 * MAXC;
    double (*sV)[4 * MAXC] = reinterpret_cast<double (*)[4 * MAXC]>(s_blk + 40 * MAXC);
    double (*sCoef)[3] = reinterpret_cast<double (*)[3]>(s_blk + 52 * MAXC);
    __shared__ CholBlockShared s_cb;
    __shared__ double sBeta_all[NSUB][3], sVV_all[NSUB][3];  // beta_k ; v2.v1, v3.v1, v3.v2
    __shared__ int sObsOfClone_all[NSUB][MAX_CLONES_DEV];   // indexed by clone id (any clone of the state)
    __shared__ int sCloneOfObs_all[NSUB][MAXC];
    extern __shared__ double s_arena_all[];
    double *s_arena = s_arena_all + (size_t)sub * arena_doubles;
    typedef typename std::conditional<WAVE, TriScratchT<2 * TRI_SMALL_CLONES>,
                                      typename std::conditional<(MAXC <= FEAT_SMALL_CLONES), TriScratchT<2 * MAXC>, TriScratch>::type>::type TriT;
    TriT &sTri = *reinterpret_cast<TriT *>(s_arena);
    __shared__ double sW_all[NSUB][4 * MAXC];
    __shared__ double sRo_all[NSUB][4 * MAXC];      // projected residual r_o (also written to global for the stacked system)
    __shared__ double sPos_all[NSUB][3];
    __shared__ int sValid_all[NSUB];
    __shared__ double sRed[8];
    double *sBeta = sBeta_all[sub], *sVV = sVV_all[sub], *sW = sW_all[sub], *sPos = sPos_all[sub], *sRo = sRo_all[sub];
    int *sObsOfClone = sObsOfClone_all[sub], *sCloneOfObs = sCloneOfObs_all[sub];
    int &sValid = sValid_all[sub];
    auto GSYNC = [&]() {
        // one wave: its LDS operations execute in program order; drain the LDS counter and keep the compiler from
        // reordering (no global memory is written and read back inside a feature: the residual travels through sRo)
        if (WAVE) { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f) /* lgkmcnt(0) only */; asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    auto gsum = [&](double v) { return WAVE ? wave_sum(v) : block_sum(v, sRed); };

    int ordinal = -1;
    for (int j = 0; j < S.n_feat; ++j) {
        EkfFeatDev &F = S.feats[j];
        {
            const int cls = F.needs_init && F.n_init > F.n_obs ? F.n_init : F.n_obs;
            if (cls < cls_lo || cls > cls_hi) continue;       // another launch's size class (uniform across the feature's threads)
            if (++ordinal % w_nslots != w_slot) continue;     // another group's feature
        }
        GSYNC();
#ifdef FEAT_DBG
        long long tdbg[12]; int ndbg = 0;
#define TDBG() do { if (ndbg < 12) tdbg[ndbg++] = wall_clock64(); } while (0)
#else
#define TDBG() do {} while (0)
#endif
        TDBG();
        const int M = F.n_obs, rows = 4 * M, n = rows - 3;
        double *Hrow0 = S.Hs + (size_t)F.row_off * ld;
        // ---- 1. position
        if (gt < 64) {
            bool valid = true;
            double pos[3] = {F.position[0], F.position[1], F.position[2]};
            if (F.needs_init) valid = triangulate_wave(S, F, sTri, pos);
            if (gt == 0) {
                sPos[0] = pos[0]; sPos[1] = pos[1]; sPos[2] = pos[2];
                sValid = valid ? 1 : 0;
                F.position[0] = pos[0]; F.position[1] = pos[1]; F.position[2] = pos[2];
                S.pos_out[3 * j] = pos[0]; S.pos_out[3 * j + 1] = pos[1]; S.pos_out[3 * j + 2] = pos[2];
            }
        }
        for (int c = gt; c < MAX_CLONES_DEV; c += GS) sObsOfClone[c] = -1;
        GSYNC();
        if (!sValid || M < 2) {
            if (gt == 0) { S.feat_status[j] = 0; S.gamma[j] = -1.0; F.colmask = 0ULL; }
            for (int i = gt; i < n; i += GS) S.rowmask[F.row_off + i] = 0ULL;       // its rows are not stacked
            continue;
        }
        TDBG();
        // ---- 2. per-observation Jacobians (msckf_vio.cpp:610-677)
        if (gt < M) {
            const int o = F.obs_start + gt;
            const int ci = S.obs_clone[o];
            sObsOfClone[ci] = gt;
            sCloneOfObs[gt] = ci;
            const mskf_clone_state &cam = S.clones[ci];
            double R_w_c0[9], R_w_c1[9], tmp[3];
            quat_to_rot(cam.q, R_w_c0);
            mat3_mul(S.R_c0_c1, R_w_c0, R_w_c1);
            mat3t_vec(R_w_c1, S.t_c0_c1, tmp);
            const double t_c1_w[3] = {cam.p[0] - tmp[0], cam.p[1] - tmp[1], cam.p[2] - tmp[2]};
            const double dp0[3] = {sPos[0] - cam.p[0], sPos[1] - cam.p[1], sPos[2] - cam.p[2]};
            const double dp1[3] = {sPos[0] - t_c1_w[0], sPos[1] - t_c1_w[1], sPos[2] - t_c1_w[2]};
            double p_c0[3], p_c1[3];
            mat3_vec(R_w_c0, dp0, p_c0);
            mat3_vec(R_w_c1, dp1, p_c1);
            // dz_dpc0 (rows 0,1), dz_dpc1 (rows 2,3)
            double dz[4][3] = {{1 / p_c0[2], 0, -p_c0[0] / (p_c0[2] * p_c0[2])},
                               {0, 1 / p_c0[2], -p_c0[1] / (p_c0[2] * p_c0[2])},
                               {1 / p_c1[2], 0, -p_c1[0] / (p_c1[2] * p_c1[2])},
                               {0, 1 / p_c1[2], -p_c1[1] / (p_c1[2] * p_c1[2])}};
            // dpc0_dxc = [skew(p_c0), -R_w_c0], dpc1_dxc = [R_c0_c1 skew(p_c0), -R_w_c1]
            const double sk[9] = {0, -p_c0[2], p_c0[1], p_c0[2], 0, -p_c0[0], -p_c0[1], p_c0[0], 0};
            double Rsk[9];
            mat3_mul(S.R_c0_c1, sk, Rsk);
            double Hx[4][6];
            for (int rr = 0; rr < 4; ++rr) {
                const double *L = rr < 2 ? sk : Rsk;
                const double *Rm = rr < 2 ? R_w_c0 : R_w_c1;
                for (int c = 0; c < 3; ++c) {
                    Hx[rr][c] = dz[rr][0] * L[c] + dz[rr][1] * L[3 + c] + dz[rr][2] * L[6 + c];
                    Hx[rr][3 + c] = -(dz[rr][0] * Rm[c] + dz[rr][1] * Rm[3 + c] + dz[rr][2] * Rm[6 + c]);
                }
            }
            // observability constraint: H_x <- A - A u (u^T u)^-1 u^T ; H_f <- -H_x[:, 3:6]
            double Rn[9], u[6], g[3] = {S.gravity[0], S.gravity[1], S.gravity[2]};
            quat_to_rot(cam.q_null, Rn);
            mat3_vec(Rn, g, u);
            const double dn[3] = {sPos[0] - cam.p_null[0], sPos[1] - cam.p_null[1], sPos[2] - cam.p_null[2]};
            u[3] = dn[1] * g[2] - dn[2] * g[1]; u[4] = dn[2] * g[0] - dn[0] * g[2]; u[5] = dn[0] * g[1] - dn[1] * g[0];
            double uu = 0;
            for (int k = 0; k < 6; ++k) uu += u[k] * u[k];
            for (int rr = 0; rr < 4; ++rr) {
                double Au = 0;
                for (int k = 0; k < 6; ++k) Au += Hx[rr][k] * u[k];
                for (int c = 0; c < 6; ++c) sHx[4 * gt + rr][c] = Hx[rr][c] - Au * (1.0 / uu) * u[c];
            }
            for (int rr = 0; rr < 4; ++rr) for (int c = 0; c < 3; ++c) sHf[4 * gt + rr][c] = -sHx[4 * gt + rr][3 + c];
            const double *z = S.obs_z + 4 * o;
            sr[4 * gt + 0] = z[0] - p_c0[0] / p_c0[2];
            sr[4 * gt + 1] = z[1] - p_c0[1] / p_c0[2];
            sr[4 * gt + 2] = z[2] - p_c1[0] / p_c1[2];
            sr[4 * gt + 3] = z[3] - p_c1[1] / p_c1[2];
        }
        GSYNC();
        TDBG();
        // ---- 3. three Householder reflectors of H_f (left null space, msckf_vio.cpp:757-766)
        for (int k = 0; k < 3; ++k) {
            double part = 0;
            for (int i = k + gt; i < rows; i += GS) part += sHf[i][k] * sHf[i][k];
            const double nrm2 = gsum(part);
            const double nrm = sqrt(nrm2);
            const double x0 = sHf[k][k];
            const double alpha = x0 > 0 ? -nrm : nrm;
            GSYNC();
            for (int i = gt; i < rows; i += GS) sV[k][i] = (i < k) ? 0.0 : (i == k ? x0 - alpha : sHf[i][k]);
            GSYNC();
            double pv = 0;
            for (int i = k + gt; i < rows; i += GS) pv += sV[k][i] * sV[k][i];
            const double vn = gsum(pv);
            const double beta = (nrm == 0.0 || vn == 0.0) ? 0.0 : 2.0 / vn;
            if (gt == 0) sBeta[k] = beta;
            // apply to the remaining columns of H_f
            for (int c = k + 1; c < 3; ++c) {
                double pd = 0;
                for (int i = k + gt; i < rows; i += GS) pd += sV[k][i] * sHf[i][c];
                const double sdot = gsum(pd) * beta;
                for (int i = k + gt; i < rows; i += GS) sHf[i][c] -= sdot * sV[k][i];
                GSYNC();
            }
        }
        {
            double p21 = 0, p31 = 0, p32 = 0;
            for (int i = gt; i < rows; i += GS) { p21 += sV[1][i] * sV[0][i]; p31 += sV[2][i] * sV[0][i]; p32 += sV[2][i] * sV[1][i]; }
            const double a = gsum(p21), b = gsum(p31), c = gsum(p32);
            if (gt == 0) { sVV[0] = a; sVV[1] = b; sVV[2] = c; }
        }
        GSYNC();
        TDBG();
        // ---- 4. coefficients c_k of every compact column (6M Jacobian columns + the residual)
        {
            // the residual column is a dot product over all rows: reduced across the group (one thread walking the rows was
            // 10 us of a 29-observation feature)
            double r1 = 0, r2 = 0, r3 = 0;
            for (int i = gt; i < rows; i += GS) { const double rv = sr[i]; r1 += sV[0][i] * rv; r2 += sV[1][i] * rv; r3 += sV[2][i] * rv; }
            r1 = gsum(r1); r2 = gsum(r2); r3 = gsum(r3);
            for (int cc = gt; cc <= 6 * M; cc += GS) {
                double s1 = 0, s2 = 0, s3 = 0;
                if (cc < 6 * M) {
                    const int blk = cc / 6, c6 = cc - 6 * blk;
                    for (int rr = 0; rr < 4; ++rr) {
                        const double v = sHx[4 * blk + rr][c6];
                        s1 += sV[0][4 * blk + rr] * v; s2 += sV[1][4 * blk + rr] * v; s3 += sV[2][4 * blk + rr] * v;
                    }
                } else { s1 = r1; s2 = r2; s3 = r3; }
                const double c1 = sBeta[0] * s1;
                const double c2 = sBeta[1] * (s2 - c1 * sVV[0]);
                const double c3 = sBeta[2] * (s3 - c1 * sVV[1] - c2 * sVV[2]);
                sCoef[cc][0] = c1; sCoef[cc][1] = c2; sCoef[cc][2] = c3;
            }
        }
        GSYNC();
        TDBG();
        // ---- 5. write the projected block: rows 3..4M-1 of Q^T [H_xj | r_j], only the 6 M columns of the observed
        //         clones and the residual column (the rest of a row is never read: rowmask, k_ekf_cap)
        if constexpr (MAXC <= 32) {
            // a lane's columns (cc = lane, lane + 64, ...) do not depend on the row: their coefficients, clone column and
            // block position are taken into registers once, a row then costs three broadcast reads and one store per column
            constexpr int NCH = (6 * MAXC + 63) / 64;
            double q0[NCH], q1[NCH], q2[NCH];
            int qcol[NCH], qob[NCH], qc6[NCH];
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int cc = (gt & 63) + 64 * q;
                qob[q] = -1; qcol[q] = 0; qc6[q] = 0; q0[q] = q1[q] = q2[q] = 0.0;
                if (cc < 6 * M) {
                    const int ob = cc / 6, c6 = cc - 6 * ob;
                    qob[q] = ob; qc6[q] = c6; qcol[q] = EKF_IMU_DIM + 6 * sCloneOfObs[ob] + c6;
                    q0[q] = sCoef[cc][0]; q1[q] = sCoef[cc][1]; q2[q] = sCoef[cc][2];
                }
            }
            for (int i = 3 + (gt >> 6); i < rows; i += GS / 64) {       // one output row per wave pass
                const double v0 = sV[0][i], v1 = sV[1][i], v2 = sV[2][i];
                double *out = Hrow0 + (size_t)(i - 3) * ld;
                const int ob_i = i >> 2;
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    if (qob[q] < 0) continue;
                    const double base = (ob_i == qob[q]) ? sHx[i][qc6[q]] : 0.0;
                    out[qcol[q]] = base - q0[q] * v0 - q1[q] * v1 - q2[q] * v2;
                }
            }
        } else
        for (int i = 3 + (gt >> 6); i < rows; i += GS / 64) {       // one output row per wave pass, compact columns across lanes
            const double v0 = sV[0][i], v1 = sV[1][i], v2 = sV[2][i];
            double *out = Hrow0 + (size_t)(i - 3) * ld;
            const int ob_i = i >> 2;
            for (int cc = gt & 63; cc < 6 * M; cc += 64) {
                const int ob = cc / 6, c6 = cc - 6 * ob;
                const double base = (ob_i == ob) ? sHx[i][c6] : 0.0;
                out[EKF_IMU_DIM + 6 * sCloneOfObs[ob] + c6] = base - sCoef[cc][0] * v0 - sCoef[cc][1] * v1 - sCoef[cc][2] * v2;
            }
        }
        for (int i = 3 + gt; i < rows; i += GS) {
            const double rv = sr[i] - sCoef[6 * M][0] * sV[0][i] - sCoef[6 * M][1] * sV[1][i] - sCoef[6 * M][2] * sV[2][i];
            sRo[i - 3] = rv;
            Hrow0[(size_t)(i - 3) * ld + d] = rv;   // column d of the stacked matrix carries the residual ([H | r])
        }
        GSYNC();
        TDBG();
        // ---- 6. gating test: gamma = r^T (H P H^T + sigma^2 I)^-1 r    (msckf_vio.cpp:909-935)
        // H = A^T H_xj with H_xj block diagonal (4x6 per observation), so H P H^T = A^T Mm A with
        // Mm[a][b] = H_a P_ab H_b^T (4x4 blocks from 6x6 blocks of P), and A^T . A = rows/cols 3.. of Q^T . Q.
        // Everything that touches the gate matrix is instantiated twice, once with the LDS arena and once with the global
        // scratch: through ONE pointer that may be either, every access is a flat instruction, and a flat access to LDS
        // costs several times a ds_read (the W pass, the rank-6 update and the factorisation all wait on it).
        auto gate_steps = [&](double *Mm) __attribute__((always_inline)) {
        const double *P = S.P;
        for (int pr = gt; pr < M * M; pr += GS) {
            const int a = pr / M, b = pr - a * M;
            if (b > a) continue;
            const int ca = EKF_IMU_DIM + 6 * sCloneOfObs[a], cb = EKF_IMU_DIM + 6 * sCloneOfObs[b];
            double Pab[6][6];
            for (int u = 0; u < 6; ++u) for (int v = 0; v < 6; ++v) Pab[u][v] = P[(size_t)(ca + u) * ld + cb + v];
            double HP[4][6];
            for (int i = 0; i < 4; ++i) for (int v = 0; v < 6; ++v) {
                double t = 0;
                for (int u = 0; u < 6; ++u) t += sHx[4 * a + i][u] * Pab[u][v];
                HP[i][v] = t;
            }
            for (int i = 0; i < 4; ++i) for (int jj = 0; jj < 4; ++jj) {
                if (a == b && jj > i) continue;
                double t = 0;
                for (int v = 0; v < 6; ++v) t += HP[i][v] * sHx[4 * b + jj][v];
                Mm[pk(4 * a + i, 4 * b + jj)] = t;
            }
        }
        GSYNC();
        TDBG();
        // two-sided reflectors in one pass: Q = Q_1 Q_2 Q_3 = I - V T V^T (compact WY, T 3x3 upper triangular), so
        //   Q^T Mm Q = Mm - Z V^T - V Z^T,   Z = W T - 1/2 V (T^T G T),   W = Mm V,   G = V^T W
        // (Mm symmetric, lower stored; in LDS or, for features with more rows than fit, in the stream's global scratch).
        // W and Z (rows x 3 each) live in the dead H_f / coefficient scratch.
        {
            double (*sWm)[3] = sHf;                 // rows x 3, H_f is dead after step 4
            double (*sZ)[3] = sCoef;                // rows x 3 (<= 4 MAXC), the coefficients are dead after step 5
            if (gt < rows) {
                const int i = gt;
                double w0 = 0, w1 = 0, w2 = 0;
                const double *mi = Mm + pk(i, 0);
#pragma unroll 4
                for (int c = 0; c <= i; ++c) { const double m = mi[c]; w0 += m * sV[0][c]; w1 += m * sV[1][c]; w2 += m * sV[2][c]; }
#pragma unroll 4
                for (int c = i + 1; c < rows; ++c) { const double m = Mm[pk(c, i)]; w0 += m * sV[0][c]; w1 += m * sV[1][c]; w2 += m * sV[2][c]; }
                sWm[i][0] = w0; sWm[i][1] = w1; sWm[i][2] = w2;
            }
            GSYNC();
            TDBG();
            double g[6] = {0, 0, 0, 0, 0, 0};       // G: 00 01 02 11 12 22
            for (int i = gt; i < rows; i += GS) {
                const double a0 = sV[0][i], a1 = sV[1][i], a2 = sV[2][i];
                g[0] += a0 * sWm[i][0]; g[1] += a0 * sWm[i][1]; g[2] += a0 * sWm[i][2];
                g[3] += a1 * sWm[i][1]; g[4] += a1 * sWm[i][2]; g[5] += a2 * sWm[i][2];
            }
            for (int q = 0; q < 6; ++q) g[q] = gsum(g[q]);
            // T (upper): T00 = b1, T11 = b2, T22 = b3, T01 = -b2 T00 (v1.v2), [T02; T12] = -b3 T[0:2,0:2] [v1.v3; v2.v3]
            const double b1 = sBeta[0], b2 = sBeta[1], b3 = sBeta[2];
            const double T01 = -b2 * b1 * sVV[0];
            const double T02 = -b3 * (b1 * sVV[1] + T01 * sVV[2]);
            const double T12 = -b3 * (b2 * sVV[2]);
            // U = T^T G T (symmetric 3x3): first X = G T, then U = T^T X
            const double G00 = g[0], G01 = g[1], G02 = g[2], G11 = g[3], G12 = g[4], G22 = g[5];
            const double X00 = G00 * b1, X01 = G00 * T01 + G01 * b2, X02 = G00 * T02 + G01 * T12 + G02 * b3;
            const double X10 = G01 * b1, X11 = G01 * T01 + G11 * b2, X12 = G01 * T02 + G11 * T12 + G12 * b3;
            const double X20 = G02 * b1, X21 = G02 * T01 + G12 * b2, X22 = G02 * T02 + G12 * T12 + G22 * b3;
            const double U00 = b1 * X00, U01 = b1 * X01, U02 = b1 * X02;
            const double U11 = T01 * X01 + b2 * X11, U12 = T01 * X02 + b2 * X12;
            const double U22 = T02 * X02 + T12 * X12 + b3 * X22;
            (void)X10; (void)X20; (void)X21;
            for (int i = gt; i < rows; i += GS) {
                const double a0 = sV[0][i], a1 = sV[1][i], a2 = sV[2][i];
                const double w0 = sWm[i][0], w1 = sWm[i][1], w2 = sWm[i][2];
                sZ[i][0] = w0 * b1 - 0.5 * (a0 * U00 + a1 * U01 + a2 * U02);
                sZ[i][1] = w0 * T01 + w1 * b2 - 0.5 * (a0 * U01 + a1 * U11 + a2 * U12);
                sZ[i][2] = w0 * T02 + w1 * T12 + w2 * b3 - 0.5 * (a0 * U02 + a1 * U12 + a2 * U22);
            }
            GSYNC();
            TDBG();
            // Mm -= Z V^T + V Z^T over a flat index of the packed lower triangle: every element is independent, so the
            // LDS latencies of consecutive iterations overlap (a wave walking one row per pass was 16 us at 16 observations)
            {
                const int n_el = rows * (rows + 1) / 2;
#pragma unroll 2
                for (int e = gt; e < n_el; e += GS) {
                    int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                    if ((i + 1) * (i + 2) / 2 <= e) ++i;                 // float rounding: at most one off
                    if (i * (i + 1) / 2 > e) --i;
                    const int c = e - i * (i + 1) / 2;
                    Mm[e] -= sZ[i][0] * sV[0][c] + sZ[i][1] * sV[1][c] + sZ[i][2] * sV[2][c] + sV[0][i] * sZ[c][0] + sV[1][i] * sZ[c][1] + sV[2][i] * sZ[c][2];
                }
            }
            GSYNC();
        }
        // S = Mm[3:,3:] + sigma^2 I ; in-place right-looking Cholesky (lower) on the sub-matrix view.  The residual
        // r_o rides along as an extra row (sW): after step k it holds y_k = (L^-1 r_o)_k, so gamma = y . y
        // needs no separate triangular solve.
#define SG(i, j) Mm[pk((i) + 3, (j) + 3)]
        TDBG();
        bool pd_ok = true;
        // workgroup variant: blocked factorisation (chol_block.h) whether the gate matrix sits in LDS or, for features
        // with more rows than fit, in the stream's global scratch (same code through a generic pointer)
        const bool blocked = !WAVE;
        if (blocked) {
            // packed row `rows` (columns 3 ..) carries r_o: the blocked factorisation leaves L^-1 r_o there
            for (int i = gt; i < n; i += GS) { SG(i, i) += S.sigma2; SG(n, i) = sRo[i]; }
            for (int i = gt; i < LNB * PAN_RS; i += GS) s_blk[i] = 0.0;            // panel buffer (scratch is dead)
            if (gt < 16) Mm[pk(rows, 3 + n) + gt] = 0.0;                          // readable slack behind the last row
            GSYNC();
            chol_blocked_lds<TPB / 64>(Mm, [](int i, int j) { return (int)pk(i + 3, j + 3); }, n, n + 1, 0.0, s_blk, PAN_RS, s_cb);
            int bad = 0;
            for (int i = gt; i < n; i += GS) { bad |= !(SG(i, i) > 0.0); sW[i] = SG(n, i); }
            pd_ok = !__syncthreads_or(bad);
        } else {
            for (int i = gt; i < n; i += GS) { SG(i, i) += S.sigma2; sW[i] = sRo[i]; }
            GSYNC();
            for (int k = 0; k < n; ++k) {
                const double dk = SG(k, k);
                if (!(dk > 0)) { pd_ok = false; break; }
                const double inv = 1.0 / sqrt(dk);
                GSYNC();
                for (int i = k + gt; i < n; i += GS) SG(i, k) *= inv;   // column k: l_kk = sqrt(dk), l_ik = a_ik / l_kk
                if (gt == 0) sW[k] *= inv;                               // y_k
                GSYNC();
                const int rem = n - k - 1;
                const double yk = sW[k];
                for (int idx = gt; idx < rem * rem; idx += GS) {
                    const int a = idx / rem + k + 1, b = idx % rem + k + 1;
                    if (b <= a) SG(a, b) -= SG(a, k) * SG(b, k);
                }
                GSYNC();
                for (int i = k + 1 + gt; i < n; i += GS) sW[i] -= SG(i, k) * yk;
            }
        }
#undef SG
        TDBG();
#ifdef FEAT_DBG
        if (item == 0 && gt == 0) printf("feat M=%d TPB=%d: tri %lld jac %lld hh %lld coef %lld write %lld gateM %lld reflect W %lld GZ %lld upd %lld chol %lld\n", M, TPB,
            (tdbg[1]-tdbg[0])*10, (tdbg[2]-tdbg[1])*10, (tdbg[3]-tdbg[2])*10, (tdbg[4]-tdbg[3])*10, (tdbg[5]-tdbg[4])*10, (tdbg[6]-tdbg[5])*10, (tdbg[7]-tdbg[6])*10,
            (tdbg[8]-tdbg[7])*10, (tdbg[9]-tdbg[8])*10, (tdbg[10]-tdbg[9])*10);
#endif
        double gamma = 1e300;
        if (pd_ok) {
            GSYNC();
            double pg = 0;
            for (int i = gt; i < n; i += GS) pg += sW[i] * sW[i];
            gamma = gsum(pg);
        }
        const int dof = M + S.dof_offset;
        const bool pass = pd_ok && dof >= 1 && dof < 100 && gamma < S.chi2[dof];
        if (gt < 64) {
            // what the block's rows carry: the clone bits of its observations if it passed the gate, 0 otherwise (every row of the
            // block: the Gram pass reads a row only where its mask says so; k_ekf_cap wrote these masks in rounds 1-3)
            unsigned long long cm = 0ULL;
            if (pass) for (int o = gt; o < M; o += 64) cm |= 1ULL << sCloneOfObs[o];
            for (int off = 32; off > 0; off >>= 1) cm |= __shfl_xor(cm, off);
            for (int i = gt; i < n; i += 64) S.rowmask[F.row_off + i] = cm;
            if (gt == 0) { S.feat_status[j] = (uint8_t)(1 | (pass ? 2 : 0)); S.gamma[j] = gamma; F.colmask = cm; }
        }
        };
        if (rows <= lds_rows) gate_steps(s_arena);
        else gate_steps(S.gate_S + (size_t)w_slot * S.nmax * S.nmax);       // packed lower, global
    }
}

// ------------------------------------------------------------------------------------ pair features (pruning update)
// The pruning update (msckf_vio.cpp:1073-1153) hands over hundreds of features per stream that all have exactly TWO
// Jacobian observations: the two clones being removed.  A workgroup (or even a wavefront) per feature is almost all
// latency there, so this path gives every feature ONE THREAD: triangulation of the not yet initialised ones first
// (k_ekf_triangulate, one wavefront per such feature, the same Levenberg-Marquardt routine), then k_ekf_pair_blocks with
// the feature's 8 x 13 block [H_x | r] and its reflectors in a private LDS slab.  Same algebra as k_ekf_feature_blocks:
// per-observation Jacobians with the observability projection, three Householder reflectors of H_f, rows 3..7 of
// Q^T [H_x | r] written in the twelve columns of the two clones + the residual column, gate on
// gamma = r_o^T (H_o P_cc H_o^T + sigma^2 I)^-1 r_o against chi2[2 + dof_offset].
__global__ __launch_bounds__(64) void k_ekf_triangulate(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (!(S.route & 1)) return;        // not a pair-route stream (the route is a property of the stream, ekf_device.h)
    __shared__ TriScratch sTri;
    for (int t = blockIdx.x; t < S.n_tri; t += gridDim.x) {
        const int j = S.tri_idx[t];
        EkfFeatDev &F = S.feats[j];
        double pos[3] = {F.position[0], F.position[1], F.position[2]};
        __syncthreads();
        const bool valid = triangulate_wave(S, F, sTri, pos);
        if (threadIdx.x == 0) {
            F.position[0] = pos[0]; F.position[1] = pos[1]; F.position[2] = pos[2];
            S.feat_status[j] = valid ? 1 : 0;
        }
    }
}

#define PAIR_SLAB 153          // doubles per thread: X 8 x 13 (104) + H_f 8 x 3 (24) + V 3 x 8 (24) + 1 (odd stride: no bank conflicts)
__global__ __launch_bounds__(64) void k_ekf_pair_blocks(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (!(S.route & 1)) return;
    const int d = S.d, ld = S.ld;
    extern __shared__ double s_pair[];
    __shared__ double sPcc[12 * 12];
    __shared__ double sRw[2][2][9], sTc1[2][3], sRn[2][9];     // per clone of the pair: R_w_c0, R_w_c1, t_c1_w, R(q_null)
    __shared__ int sPair[2];
    double *X = s_pair + (size_t)threadIdx.x * PAIR_SLAB;       // [8][13]: H_x of obs 0 in columns 0..5, obs 1 in 6..11, r in 12
    double *Hf = X + 104;                                        // [8][3]
    double *V = Hf + 24;                                         // [3][8]
    for (int base = blockIdx.x * 64; base < S.n_feat; base += gridDim.x * 64) {
        const int j = base + threadIdx.x;
        const bool have = j < S.n_feat;
        // the clone pair of the first feature of this batch; P_cc and the clone rotations are shared by every feature with that pair
        __syncthreads();
        if (threadIdx.x == 0) {
            const EkfFeatDev &F0 = S.feats[base];
            sPair[0] = S.obs_clone[F0.obs_start]; sPair[1] = S.obs_clone[F0.obs_start + 1];
        }
        __syncthreads();
        const int pa = sPair[0], pb = sPair[1];
        for (int e = threadIdx.x; e < 144; e += 64) {
            const int u = e / 12, v = e - 12 * u;
            const int cu = EKF_IMU_DIM + 6 * (u < 6 ? pa : pb) + (u % 6), cv = EKF_IMU_DIM + 6 * (v < 6 ? pa : pb) + (v % 6);
            sPcc[e] = S.P[(size_t)cu * ld + cv];
        }
        if (threadIdx.x < 2) {
            const mskf_clone_state &cam = S.clones[threadIdx.x == 0 ? pa : pb];
            double R0[9], R1[9], tmp[3];
            quat_to_rot(cam.q, R0);
            mat3_mul(S.R_c0_c1, R0, R1);
            mat3t_vec(R1, S.t_c0_c1, tmp);
            for (int i = 0; i < 9; ++i) { sRw[threadIdx.x][0][i] = R0[i]; sRw[threadIdx.x][1][i] = R1[i]; }
            for (int i = 0; i < 3; ++i) sTc1[threadIdx.x][i] = cam.p[i] - tmp[i];
            quat_to_rot(cam.q_null, R0);
            for (int i = 0; i < 9; ++i) sRn[threadIdx.x][i] = R0[i];
        }
        __syncthreads();
        if (!have) continue;
        EkfFeatDev &F = S.feats[j];
        const int c0 = S.obs_clone[F.obs_start], c1 = S.obs_clone[F.obs_start + 1];
        const bool valid = F.needs_init ? (S.feat_status[j] & 1) != 0 : true;
        const double pos[3] = {F.position[0], F.position[1], F.position[2]};
        S.pos_out[3 * j] = pos[0]; S.pos_out[3 * j + 1] = pos[1]; S.pos_out[3 * j + 2] = pos[2];
        if (!valid || F.n_obs != 2 || c0 != pa || c1 != pb) {      // (a foreign pair cannot happen: the host checks the batch)
            S.feat_status[j] = 0; S.gamma[j] = -1.0; F.colmask = 0ULL;
            for (int i = 0; i < 4 * F.n_obs - 3; ++i) S.rowmask[F.row_off + i] = 0ULL;
            continue;
        }
        // ---- per-observation Jacobians (msckf_vio.cpp:610-677)
        for (int i = 0; i < 104; ++i) X[i] = 0.0;
        const double g[3] = {S.gravity[0], S.gravity[1], S.gravity[2]};
#pragma unroll 1
        for (int ob = 0; ob < 2; ++ob) {
            const mskf_clone_state &cam = S.clones[ob == 0 ? pa : pb];
            const double *R_w_c0 = sRw[ob][0], *R_w_c1 = sRw[ob][1];
            const double dp0[3] = {pos[0] - cam.p[0], pos[1] - cam.p[1], pos[2] - cam.p[2]};
            const double dp1[3] = {pos[0] - sTc1[ob][0], pos[1] - sTc1[ob][1], pos[2] - sTc1[ob][2]};
            double p_c0[3], p_c1[3];
            mat3_vec(R_w_c0, dp0, p_c0);
            mat3_vec(R_w_c1, dp1, p_c1);
            const double dz[4][3] = {{1 / p_c0[2], 0, -p_c0[0] / (p_c0[2] * p_c0[2])},
                                     {0, 1 / p_c0[2], -p_c0[1] / (p_c0[2] * p_c0[2])},
                                     {1 / p_c1[2], 0, -p_c1[0] / (p_c1[2] * p_c1[2])},
                                     {0, 1 / p_c1[2], -p_c1[1] / (p_c1[2] * p_c1[2])}};
            const double sk[9] = {0, -p_c0[2], p_c0[1], p_c0[2], 0, -p_c0[0], -p_c0[1], p_c0[0], 0};
            double Rsk[9];
            mat3_mul(S.R_c0_c1, sk, Rsk);
            double *Xb = X + (4 * ob) * 13 + 6 * ob;          // this observation's 4 x 6 block inside X (row stride 13)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double l0 = rr < 2 ? sk[c] : Rsk[c], l1 = rr < 2 ? sk[3 + c] : Rsk[3 + c], l2 = rr < 2 ? sk[6 + c] : Rsk[6 + c];
                    const double m0 = rr < 2 ? R_w_c0[c] : R_w_c1[c], m1 = rr < 2 ? R_w_c0[3 + c] : R_w_c1[3 + c], m2 = rr < 2 ? R_w_c0[6 + c] : R_w_c1[6 + c];
                    Xb[rr * 13 + c] = dz[rr][0] * l0 + dz[rr][1] * l1 + dz[rr][2] * l2;
                    Xb[rr * 13 + 3 + c] = -(dz[rr][0] * m0 + dz[rr][1] * m1 + dz[rr][2] * m2);
                }
            }
            double u[6];
            mat3_vec(sRn[ob], g, u);
            const double dn[3] = {pos[0] - cam.p_null[0], pos[1] - cam.p_null[1], pos[2] - cam.p_null[2]};
            u[3] = dn[1] * g[2] - dn[2] * g[1]; u[4] = dn[2] * g[0] - dn[0] * g[2]; u[5] = dn[0] * g[1] - dn[1] * g[0];
            double uu = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) uu += u[k] * u[k];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                double Au = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) Au += Xb[rr * 13 + k] * u[k];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double h = Xb[rr * 13 + c] - Au * (1.0 / uu) * u[c];
                    Xb[rr * 13 + c] = h;
                    if (c >= 3) Hf[(4 * ob + rr) * 3 + (c - 3)] = -h;
                }
            }
            const double *z = S.obs_z + 4 * (F.obs_start + ob);
            X[(4 * ob + 0) * 13 + 12] = z[0] - p_c0[0] / p_c0[2];
            X[(4 * ob + 1) * 13 + 12] = z[1] - p_c0[1] / p_c0[2];
            X[(4 * ob + 2) * 13 + 12] = z[2] - p_c1[0] / p_c1[2];
            X[(4 * ob + 3) * 13 + 12] = z[3] - p_c1[1] / p_c1[2];
        }
        // ---- three Householder reflectors of H_f, applied to H_f's remaining columns and to [H_x | r] (:757-766)
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            double nrm2 = 0;
            for (int i = k; i < 8; ++i) nrm2 += Hf[i * 3 + k] * Hf[i * 3 + k];
            const double nrm = sqrt(nrm2);
            const double x0 = Hf[k * 3 + k];
            const double alpha = x0 > 0 ? -nrm : nrm;
            double vn = 0;
            for (int i = 0; i < 8; ++i) {
                const double v = (i < k) ? 0.0 : (i == k ? x0 - alpha : Hf[i * 3 + k]);
                V[k * 8 + i] = v;
                vn += v * v;
            }
            const double beta = (nrm == 0.0 || vn == 0.0) ? 0.0 : 2.0 / vn;
            for (int c = k + 1; c < 3; ++c) {
                double sdot = 0;
                for (int i = k; i < 8; ++i) sdot += V[k * 8 + i] * Hf[i * 3 + c];
                sdot *= beta;
                for (int i = k; i < 8; ++i) Hf[i * 3 + c] -= sdot * V[k * 8 + i];
            }
#pragma unroll 1
            for (int c = 0; c < 13; ++c) {
                double sdot = 0;
                for (int i = k; i < 8; ++i) sdot += V[k * 8 + i] * X[i * 13 + c];
                sdot *= beta;
                for (int i = k; i < 8; ++i) X[i * 13 + c] -= sdot * V[k * 8 + i];
            }
        }
        // ---- gate: S_g = H_o P_cc H_o^T + sigma^2 I on rows 3..7, gamma = r_o^T S_g^-1 r_o (:909-935)
        // H_o P_cc (5 x 12) goes into the dead part of the slab: rows 0..2 of X and H_f
        double Sg[5][5];
        {
#pragma unroll 1
            for (int i = 0; i < 5; ++i) {
                double *hp = i < 3 ? X + 12 * i : Hf + 12 * (i - 3);
#pragma unroll 1
                for (int v = 0; v < 12; ++v) {
                    double t = 0;
                    for (int u2 = 0; u2 < 12; ++u2) t += X[(3 + i) * 13 + u2] * sPcc[u2 * 12 + v];
                    hp[v] = t;
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double *hp = i < 3 ? X + 12 * i : Hf + 12 * (i - 3);
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) {
                    if (jj > i) continue;
                    double t = 0;
#pragma unroll 1
                    for (int v = 0; v < 12; ++v) t += hp[v] * X[(3 + jj) * 13 + v];
                    Sg[i][jj] = t + (i == jj ? S.sigma2 : 0.0);
                }
            }
        }
        bool pd_ok = true;
        double y[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) y[i] = X[(3 + i) * 13 + 12];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double dk = Sg[k][k];
            pd_ok = pd_ok && (dk > 0);
            const double inv = 1.0 / sqrt(dk > 0 ? dk : 1.0);
#pragma unroll
            for (int i = k; i < 5; ++i) Sg[i][k] *= inv;
            y[k] *= inv;
#pragma unroll
            for (int a = k + 1; a < 5; ++a) {
#pragma unroll
                for (int b = k + 1; b <= a; ++b) Sg[a][b] -= Sg[a][k] * Sg[b][k];
                y[a] -= Sg[a][k] * y[k];
            }
        }
        double gamma = 1e300;
        if (pd_ok) { gamma = 0; for (int i = 0; i < 5; ++i) gamma += y[i] * y[i]; }
        const int dof = 2 + S.dof_offset;
        const bool pass = pd_ok && dof >= 1 && dof < 100 && gamma < S.chi2[dof];
        // ---- the projected block: rows 3..7, the twelve columns of the pair + the residual column
        double *Hrow0 = S.Hs + (size_t)F.row_off * ld;
        for (int i = 0; i < 5; ++i) {
            double *out = Hrow0 + (size_t)i * ld;
            for (int c = 0; c < 6; ++c) {
                out[EKF_IMU_DIM + 6 * pa + c] = X[(3 + i) * 13 + c];
                out[EKF_IMU_DIM + 6 * pb + c] = X[(3 + i) * 13 + 6 + c];
            }
            out[d] = X[(3 + i) * 13 + 12];
        }
        S.feat_status[j] = (uint8_t)(1 | (pass ? 2 : 0));
        S.gamma[j] = gamma;
        F.colmask = pass ? ((1ULL << pa) | (1ULL << pb)) : 0ULL;
        for (int i = 0; i < 5; ++i) S.rowmask[F.row_off + i] = F.colmask;
    }
}

// ------------------------------------------------------------------------------------ launchers
extern "C" {
void ekf_launch_propagate(const EkfStreamDev *d, int n, hipStream_t st) { hipLaunchKernelGGL(k_ekf_propagate, dim3(1, n), dim3(WG), 0, st, d); }
void ekf_launch_augment(const EkfStreamDev *d, int n, hipStream_t st) { hipLaunchKernelGGL(k_ekf_augment, dim3(1, n), dim3(WG), 0, st, d); }
void ekf_launch_remove_clone(const EkfStreamDev *d, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_ekf_remove_clone, dim3(32, n), dim3(WG), 0, st, d);
}
// work_wave / work_small / work_big: the work lists of the three size classes (see the kernel); work_wave holds the
// features of the streams whose features all have <= 4 Jacobian observations (route bit 1)
void ekf_launch_features(const EkfStreamDev *d, const int *work_wave, int n_wave, const int *work_small, int n_small, const int *work_big, int n_big,
                         int max_rows, int max_rows_small, int big_clones, hipStream_t st) {
    const int packed_max = ((GATE_LDS_ROWS + 1) * (GATE_LDS_ROWS + 2) / 2 + 16) * (int)sizeof(double);   // + the r_o row + slack
    const int tri_doubles = (int)((sizeof(TriScratchT<2 * TRI_SMALL_CLONES>) + 7) / 8);
    static std::once_flag attr_once;
    std::call_once(attr_once, [=]() {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_feature_blocks<32, false, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, packed_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_feature_blocks<MAX_CLONES_DEV, false, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, packed_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_feature_blocks<4, true, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * tri_doubles * 8);
    });
    const int ALL = 1 << 30;
    // streams whose features ALL have <= 4 Jacobian observations (and triangulations that fit the small scratch): one
    // wavefront per feature, four per workgroup.  The class is a property of the stream (route bit 1), so its work list
    // holds whole streams and the other two lists none of their features.
    if (n_wave > 0)
        hipLaunchKernelGGL((k_ekf_feature_blocks<4, true, WG>), dim3((n_wave + 3) / 4), dim3(WG), (size_t)4 * tri_doubles * 8, st, d, work_wave, n_wave, 16, tri_doubles, 0, ALL);
    // small class: features of at most FEAT_SMALL_CLONES observations (Jacobian and triangulation), one wavefront each.
    // (A third class of <= 8 observations, 12 KiB of LDS and eight features per CU, was measured and dropped: every class
    // launch is a single latency-bound round, so a further split only adds another round to the chain.)
    if (n_small > 0) {
        const int lr = max_rows_small;                                   // <= 4 * FEAT_SMALL_CLONES: always in LDS
        size_t lds = ((size_t)(lr + 1) * (lr + 2) / 2 + 16) * sizeof(double);
        if (lds < sizeof(TriScratchT<2 * FEAT_SMALL_CLONES>)) lds = sizeof(TriScratchT<2 * FEAT_SMALL_CLONES>);
        hipLaunchKernelGGL((k_ekf_feature_blocks<FEAT_SMALL_CLONES, false, 64>), dim3(n_small), dim3(64), lds, st, d, work_small, n_small, lr, 0, 0, FEAT_SMALL_CLONES);
    }
    if (n_big <= 0) return;
    const int lds_rows = max_rows <= GATE_LDS_ROWS ? max_rows : GATE_LDS_ROWS;
    size_t lds = ((size_t)(lds_rows + 1) * (lds_rows + 2) / 2 + 16) * sizeof(double);
    if (lds < sizeof(TriScratch)) lds = sizeof(TriScratch);
    // (The two class launches are independent, but an any-order launch of the second one, hipExtAnyOrderLaunch, is not
    // honoured on gfx9: the trace shows it starting when the first ends.)
    // (instantiation by the streams' configured window, not by the largest feature that happens to be in the batch)
    if (big_clones <= 32) hipLaunchKernelGGL((k_ekf_feature_blocks<32, false, WG>), dim3(n_big), dim3(WG), lds, st, d, work_big, n_big, lds_rows, 0, FEAT_SMALL_CLONES + 1, ALL);
    else hipLaunchKernelGGL((k_ekf_feature_blocks<MAX_CLONES_DEV, false, WG>), dim3(n_big), dim3(WG), lds, st, d, work_big, n_big, lds_rows, 0, FEAT_SMALL_CLONES + 1, ALL);
}
// pruning update: every feature of every stream has exactly two Jacobian observations (the host checked)
void ekf_launch_pair_features(const EkfStreamDev *d, int n, int max_feat, int max_tri, hipStream_t st) {
    if (max_tri > 0) {
        const int slots = max_tri < 256 ? max_tri : 256;
        hipLaunchKernelGGL(k_ekf_triangulate, dim3(slots, n), dim3(64), 0, st, d);
    }
    static std::once_flag attr_once;
    std::call_once(attr_once, []() {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_pair_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * PAIR_SLAB * 8);
    });
    const int blocks = (max_feat + 63) / 64;
    hipLaunchKernelGGL(k_ekf_pair_blocks, dim3(blocks, n), dim3(64), (size_t)64 * PAIR_SLAB * sizeof(double), st, d);
}
void ekf_launch_posvar(const EkfStreamDev *d, int n, double *out, hipStream_t st) {
    hipLaunchKernelGGL(k_ekf_posvar, dim3((3 * n + 63) / 64), dim3(64), 0, st, d, n, out);
}
void ekf_launch_posvar_upd(const EkfStreamDev *d, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_ekf_posvar_upd, dim3((3 * n + 63) / 64), dim3(64), 0, st, d, n);
}
}
