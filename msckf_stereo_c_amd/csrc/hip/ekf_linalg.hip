// ekf_linalg.hip — batched dense FP64 building blocks of the Kalman update (msckf_vio.cpp:795-904),
// one job per VIO stream, blockIdx.y = stream of the batch:
//
//  k_ekf_gemm   : 32x32 output tile per workgroup, v_mfma_f64_16x16x4_f64 (one 16x16 sub-tile per wave,
//                 K staged through LDS 32 at a time, next stage prefetched into registers).  Modes:
//                   GRAM  G  = [Hs|rs]^T [Hs|rs]          ((d+1)x(d+1), replaces the QR: G = R^T R, last row = (Q^T r)^T R)
//                   T     T  = R P,  R = L^T upper        (skips the zero half of R)
//                   S2    S  = T R^T + sigma^2 I          (symmetric, lower computed and mirrored)
//                   PUPD  P <- P - Y^T Y                  (symmetric)
//  k_ekf_chol   : blocked (NB = 16) right-looking Cholesky in place, one workgroup per stream; the Gram
//                 variant factors G + lambda I (lambda = 1e-14 d max diag: G is rank deficient and an unpivoted
//                 factorisation of it is unstable) and carries the extra Q^T r row through the panel solves.
//  k_ekf_trsm   : Y = L^-1 [T | Q^T r] in place, one workgroup per 32-column strip and stream.
//  (delta_x = Y^T w, w = column d of Y, is computed by extra workgroups of the PUPD GEMM launch)
//
//  All of them work on the ACTIVE columns of the update only (compact index i <-> column act[i], count rows_out[2],
//  built by k_ekf_cap): the stacked Jacobian is zero in the 21 IMU columns and in the columns of clones that no
//  stacked feature observed, and those columns add nothing to S, K or the downdate.  The pruning update (hundreds of
//  features, but only the two clones being removed) thus factors 12 x 12 systems instead of 180 x 180.
//
// Why Gram + Cholesky instead of Householder QR: the update only needs R^T R = H^T H and R^T (Q^T r) =
// H^T r; forming them is one GEMM-shaped pass over the stacked Jacobian (MFMA-friendly, fully parallel
// over output tiles) instead of d sequential reflector applications.  H^T H is rank deficient (the 21
// IMU columns of H are zero, unobserved clones, the gauge): the factorisation is regularised, see k_ekf_chol_lds.
#include <mutex>
#include <type_traits>
#include "ekf_device.h"
#include "ekf_cap.h"
#include "chol_block.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

// Gram + regularised Cholesky adds a prior lambda I to the stacked information H^T H / sigma^2: the posterior covariance
// moves by about lambda max(P_aa) / sigma^2 relative.  Above this limit (or when the stack has no more rows than active
// columns) the compression runs as Householder TSQR instead (tsqr_wide), which has no such term.
#define QR_BIAS_LIMIT 1e-6
// How a stream's stack is compressed is decided from ONE pair of numbers everywhere (the GRAM pass, the factorisation
// kernels, the fused small update): st = rows_out[0], the rows actually stacked, and na = rows_out[2], the active columns.
//   * st <= na (auto mode): the reference compresses nothing there (msckf_vio.cpp:818-821) and H^T H would be singular by
//     construction, so neither the Gram pass nor any factorisation of the stack runs: the st stacked rows themselves are the
//     measurement ("direct": R = H_act, st x na).  Blocks that were gated out leave gaps between the stacked rows (the last
//     stacked row, rows_out[1], may lie far beyond na), so the rows are addressed through a compact list of their indices
//     (rowidx, built by ekf_compress_entry behind the active-column list).
//   * otherwise Gram + regularised Cholesky, re-done as Householder TSQR when the factorisation raises the bias flag (bit 1).
// qr_mode (mskf_ekf_cfg.compression_mode): 0 auto, 1 Gram only, 2 Householder always, 3 = the reference literally
// (msckf_vio.cpp:795-821): Householder QR when the stack has more rows than columns, nothing otherwise.
__device__ __forceinline__ bool ekf_mode_direct(int qr_mode, int stacked, int na) { return (qr_mode == 0 || qr_mode == 3) && stacked <= na; }
__device__ __forceinline__ bool ekf_mode_householder(int qr_mode) { return qr_mode == 2 || qr_mode == 3; }
__device__ __forceinline__ bool ekf_direct_wanted(const EkfStreamDev &S) { return ekf_mode_direct(S.qr_mode, S.rows_out[0], S.rows_out[2]); }
__device__ __forceinline__ bool ekf_skip_gram(const EkfStreamDev &S) { return ekf_mode_householder(S.qr_mode) || ekf_direct_wanted(S); }
// the stream's stacked rows are used uncompressed (set by the factorisation kernel): R = H_act (rows_out[1] x na, dense, rowmask-gathered)
__device__ __forceinline__ bool ekf_direct(const EkfStreamDev &S) { return (S.rows_out[3] & 4) != 0; }
// streams whose whole update runs in k_ekf_small_update (route bit, set per STREAM by the host: at most SU_MAX_NA active
// columns possible); the general kernels leave them alone and vice versa
#define EKF_ROUTE_PAIRS 1
#define EKF_ROUTE_WAVE 2
#define EKF_ROUTE_SMALL 4

enum { GM_GRAM = 0, GM_T = 1, GM_S2 = 2, GM_PUPD = 3 };

// Per-mode compile-time shape of op(A) op(B): TA: op(A)(i,k) = A[k*ld+i] (else A[i*ld+k]); B is always B[k*ld+j].
// KMIN_I: op(A)(i,k) == 0 for k < i (R = L^T is upper triangular); KMIN_J: op(B)(k,j) == 0 for k < j.
template <int MODE> struct GemmTraits;
template <> struct GemmTraits<GM_GRAM> { static constexpr bool TA = true,  SYM = true,  KMIN_I = false, KMIN_J = false; };
template <> struct GemmTraits<GM_T>    { static constexpr bool TA = true,  SYM = false, KMIN_I = true,  KMIN_J = false; };
template <> struct GemmTraits<GM_S2>   { static constexpr bool TA = false, SYM = true,  KMIN_I = false, KMIN_J = true;  };
template <> struct GemmTraits<GM_PUPD> { static constexpr bool TA = true,  SYM = true,  KMIN_I = false, KMIN_J = false; };

#define GT 32      // output tile edge
#ifndef GK
#define GK 32      // K per LDS stage (64 measured slower: 49 vs 41 us per launch, the LDS footprint halves the workgroups per CU)
#endif
#define GNE (GK / 8)   // elements of each operand per thread and stage

// One 32x32 output tile per workgroup (one 16x16 MFMA sub-tile per wave).  (Tried in round 2: 64x64 tiles with a 2x2 block
// of sub-tiles per wave, i.e. half the LDS reads per MFMA.  2.6 x SLOWER at the C2 shapes, 109 vs 42 us per launch: with
// na = 100..175 a stream has 3..6 such tiles, the launch no longer fills the CUs, and a workgroup is bound by the latency of
// its ~22 K stages, not by feeding the matrix pipe.  Tried in round 3: the 32x32 tile per WAVE in a 64-thread workgroup,
// four independent accumulators, no barriers, one LDS read per MFMA — same tile count, bit-identical results, and 2 x
// slower: 83 vs 41 us per launch alone at 192 streams.  A v_mfma_f64_16x16x4 occupies the matrix pipe for 64 cycles, so
// neither dependent issue nor LDS reads per MFMA are what leaves the pipe idle; with one wave per tile a tile's 32
// global loads per stage are issued by 64 lanes instead of 256 and only 4 tiles' worth of loads are in flight per SIMD.)
// The K loop is software pipelined:
// the global loads of stage s+1 are issued into registers before the MFMAs of stage s, so a stage costs an LDS
// round trip instead of an HBM/L2 round trip.
template <int MODE>
__global__ __launch_bounds__(256) void k_ekf_gemm(const EkfStreamDev *streams) {
    using TR = GemmTraits<MODE>;
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0) return;
    if (MODE != GM_PUPD && (S.route & EKF_ROUTE_SMALL)) return;
    const int d = S.d, ld = S.ld;
    // GRAM is the first dense kernel of the update: every tile works out for itself which blocks are stacked (ekf_cap.h; tile 0
    // writes the result down for the kernels that follow).  Tiles that cannot exist whatever is stacked - above the diagonal,
    // or beyond the clones any feature of the stream observed (na_max, from the host) - leave before that.
    int gram_tile_i = 0, gram_tile_j = 0;
    EkfCapResult cap = {0, 0, 0, 0ULL};
    if (MODE == GM_GRAM) {
        int tg = 1;
        while ((tg + 1) * (tg + 1) <= (int)gridDim.x) ++tg;      // the launch is tg x tg tiles per stream
        gram_tile_i = (int)blockIdx.x / tg; gram_tile_j = (int)blockIdx.x - gram_tile_i * tg;
        if (gram_tile_j > gram_tile_i || gram_tile_i * GT > S.na_max) return;
        cap = ekf_cap_local(S, blockIdx.x == 0);
        if (ekf_mode_householder(S.qr_mode) || ekf_mode_direct(S.qr_mode, cap.stacked, cap.na)) return;       // ekf_skip_gram, on this update's own numbers
    }
    const int na = MODE == GM_GRAM ? cap.na : S.rows_out[2];                  // active columns (compact index i <-> column act[i])
    const int *__restrict__ act = S.act;
    const double *__restrict__ A; const double *__restrict__ B; double *C;
    int M, N, K;
    double alpha = 1.0, beta = 0.0, diag_add = 0.0;
    if (MODE == GM_GRAM)      { A = S.Hs; B = S.Hs; C = S.S; M = N = na + 1; K = cap.end; }       // G_c = [H_act|r]^T [H_act|r]
    const int nk = MODE == GM_GRAM ? 0 : S.rows_out[4];                                               // rows of the compressed measurement
    const bool direct = MODE != GM_GRAM && MODE != GM_PUPD && ekf_direct(S);                          // R = H_act itself (no compression)
    if (MODE == GM_T)         { A = direct ? S.Hs : S.S;  B = S.P;  C = S.T; M = nk; N = d; K = na; }   // T = R P[act, :]
    else if (MODE == GM_S2)   { A = S.T;  B = direct ? S.Hs : S.S;  C = S.W; M = N = nk; K = na; diag_add = S.sigma2; } // S = T[:, act] R^T + sigma^2 I
    else if (MODE == GM_PUPD) { A = S.T;  B = S.T;  C = S.P; M = N = d; K = nk; alpha = -1.0; beta = 1.0; }   // P -= Y^T Y
    const int tiles_n = (N + GT - 1) / GT, tiles_m = (M + GT - 1) / GT;
    const int tile = blockIdx.x;
    if (MODE == GM_PUPD && tile >= tiles_m * tiles_n) {
        // tiles_n extra workgroups: delta_x = Y^T w, w = column d of Y (msckf_vio.cpp:860); 32 columns each,
        // the K range split over the 8 thread rows and reduced through LDS
        const int c0 = (tile - tiles_m * tiles_n) * GT;
        if (c0 >= d) return;
        const double *Y = S.T;
        __shared__ double s_part[8][GT + 1];
        const int cl = threadIdx.x & 31, ks = threadIdx.x >> 5;
        const int c = c0 + cl;
        double s2 = 0;
        if (c < d)
            for (int k = ks; k < nk; k += 8) s2 += Y[(size_t)k * ld + c] * Y[(size_t)k * ld + d];
        s_part[ks][cl] = s2;
        __syncthreads();
        if (ks == 0 && c < d) {
            double t = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += s_part[q][cl];
            S.delta_x[c] = t;
        }
        return;
    }
    if (MODE != GM_GRAM && tile >= tiles_m * tiles_n) return;
    const int ti = MODE == GM_GRAM ? gram_tile_i : tile / tiles_n, tj = MODE == GM_GRAM ? gram_tile_j : tile - ti * tiles_n;
    if (MODE == GM_GRAM && ti >= tiles_m) return;
    if (TR::SYM && tj > ti) return;
    const int i0 = ti * GT, j0 = tj * GT;
    __shared__ double sA[GK][GT + 1];   // sA[k][i]
    __shared__ double sB[GK][GT + 1];   // sB[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    const int lo = tid & 31, hi = tid >> 5;          // hi in [0,8)
    int k_begin = 0;
    if (TR::KMIN_I && !direct) k_begin = (i0 / GK) * GK;
    if (TR::KMIN_J && !direct) k_begin = (j0 / GK) * GK;
    const int k_end = K;
    // gathered source columns of this thread's fixed (i = lo / j = lo) operand lanes
    int colA = i0 + lo, colB = j0 + lo;
    if (MODE == GM_GRAM) {
        colA = (i0 + lo < na) ? ekf_act_column(cap.clones, i0 + lo) : d;       // compact index na is the residual column d of [H | r]
        colB = (j0 + lo < na) ? ekf_act_column(cap.clones, j0 + lo) : d;
    }
    // GRAM: what a stacked row carries is its rowmask (bit c = the six columns of clone c, any bit = the residual column);
    // everything else of the row was never written and is not read
    const int cloneA = MODE == GM_GRAM ? (colA < d ? (colA - EKF_IMU_DIM) / 6 : -1) : 0;
    const int cloneB = MODE == GM_GRAM ? (colB < d ? (colB - EKF_IMU_DIM) / 6 : -1) : 0;
    const unsigned long long *__restrict__ rowmask = S.rowmask;
    const int *__restrict__ rowidx = S.act + S.ld;       // uncompressed update: the stacked rows, in order (ekf_compress_entry)
    double ra[GNE], rb[GNE];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < GNE; ++e) {
            if (MODE == GM_GRAM) {
                // both operands are (column = lo, row = hi + 8 e) of [H | r]
                const int gk = k0 + hi + 8 * e;
                double v = 0.0, w = 0.0;
                if (gk < K) {
                    const unsigned long long rm = rowmask[gk];
                    const bool onA = cloneA < 0 ? rm != 0ULL : ((rm >> cloneA) & 1ULL) != 0ULL;
                    const bool onB = cloneB < 0 ? rm != 0ULL : ((rm >> cloneB) & 1ULL) != 0ULL;
                    if (onA && i0 + lo < M) v = A[(size_t)gk * ld + colA];
                    if (onB && j0 + lo < N) w = B[(size_t)gk * ld + colB];
                }
                ra[e] = v; rb[e] = w;
                continue;
            }
            // A: TA -> (i = lo, k = hi + 8e) reads A[k*ld + i] coalesced in i; else (k = lo, i = hi + 8e) reads A[i*ld + k]
            const int ii = TR::TA ? lo : hi + 8 * (e / (GK / 32)), kk = TR::TA ? hi + 8 * e : lo + 32 * (e % (GK / 32));
            const int gi = i0 + ii, gk = k0 + kk;
            double v = 0.0;
            if (gi < M && gk < K && !(TR::KMIN_I && !direct && gk < gi)) {
                if (MODE == GM_S2) v = A[(size_t)gi * ld + act[gk]];              // T[:, act]
                else if (MODE == GM_T && direct) {                                // H_act row gi = stacked row rowidx[gi], rowmask-gathered
                    const int col = act[gk], ri = rowidx[gi];
                    if ((rowmask[ri] >> ((col - EKF_IMU_DIM) / 6)) & 1ULL) v = A[(size_t)ri * ld + col];
                }
                else v = A[(size_t)gk * ld + (MODE == GM_GRAM ? colA : gi)];
            }
            ra[e] = v;
            const int gj = j0 + lo, gkb = k0 + hi + 8 * e;
            double w = 0.0;
            if (gj < N && gkb < K && !(TR::KMIN_J && !direct && gkb < gj)) {
                if (MODE == GM_T) w = B[(size_t)act[gkb] * ld + gj];             // P[act, :]
                else if (MODE == GM_S2 && direct) {                               // (H_act)^T: stacked row rowidx[gj] of H, column act[gkb]
                    const int col = act[gkb], rj = rowidx[gj];
                    if ((rowmask[rj] >> ((col - EKF_IMU_DIM) / 6)) & 1ULL) w = B[(size_t)rj * ld + col];
                }
                else w = B[(size_t)gkb * ld + (MODE == GM_GRAM ? colB : gj)];
            }
            rb[e] = w;
        }
    };
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += GK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < GNE; ++e) {
            if (TR::TA) sA[hi + 8 * e][lo] = ra[e]; else sA[lo + 32 * (e % (GK / 32))][hi + 8 * (e / (GK / 32))] = ra[e];
            sB[hi + 8 * e][lo] = rb[e];
        }
        __syncthreads();
        if (k0 + GK < k_end) fetch(k0 + GK);
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
            const int kk = 4 * s + (lane >> 4);
            const double a = sA[kk][wi + (lane & 15)];
            const double b = sB[kk][wj + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
    // epilogue: lane holds D[row = (lane>>4) + 4 r][col = lane & 15]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi + (lane >> 4) + 4 * r;
        const int j = j0 + wj + (lane & 15);
        if (i >= M || j >= N) continue;
        if (TR::SYM && j > i) continue;             // diagonal tiles: lower part only, mirrored below
        double v = alpha * acc[r];
        if (MODE == GM_PUPD) v += beta * C[(size_t)i * ld + j];
        if (i == j) v += diag_add;
        C[(size_t)i * ld + j] = v;
        if (TR::SYM && i != j) C[(size_t)j * ld + i] = v;
        // the position variances P(12..14, 12..14) after the update ride home with the results (onlineReset, msckf_vio.cpp:1194-1196;
        // round 3 read them with a launch of its own behind this one)
        if (MODE == GM_PUPD && i == j && i >= 12 && i < 15 && S.pos_var_out) S.pos_var_out[i - 12] = v;
    }
}

// ------------------------------------------------------------------------------------ Householder TSQR
// The reference compresses the stacked Jacobian with a Householder QR (SPQR / Eigen HouseholderQR, msckf_vio.cpp:795-817).
// The default here is the Gram matrix + a regularised Cholesky (one MFMA pass, fully parallel), which squares the
// condition number of H and adds the prior lambda I.  The Kalman update itself is regularised by P, so what that costs is
// bounded by lambda max(P_aa) / sigma^2 whatever cond(H) is (measured: tests/test_gpu_kernels.py, condition sweep); the
// factorisation evaluates that bound and this path takes over when it exceeds QR_BIAS_LIMIT, when the stack has no more
// rows than active columns (the reference's m <= d case: nothing is compressed there, :818-821; H^T H would be singular
// by construction), or always with compression_mode = 2.
// Row-block TSQR: the upper-triangular R of [H_act | r] (n1 = na + 1 columns, the residual rides along as the last one)
// stays resident (LDS, packed by rows, when it fits; the stream's W buffer otherwise); the stacked rows are
// streamed through LDS sixteen at a time and annihilated column by column against R's diagonal with Householder
// reflectors of length 17.  Called by every thread of the workgroup; thread j owns column j of the block during a step.
// colOf(c, col, clone): source column of compact column c in [H | r] and the clone whose rowmask bit guards it (-1: residual).
// (Round 2 ran this as a kernel of its own, launched after the Gram factorisation of EVERY update just to find that nothing
// was to do: 1.6 % of the kernel time of the bench and one more link in the update's dependency chain.  The factorisation
// kernels call it now, with their own LDS: the packed factor's space holds R, the panel's the row block.)
// Round 4: WIDE row blocks.  Round 3 streamed sixteen rows at a time with one thread per column: a step (one reflector) is a
// workgroup barrier plus 16 multiply-adds per thread, ceil(K / 16) * n1 steps in all (7500 at C2: 50 row blocks x 150
// columns), most of the workgroup idle (150 of 512 threads have a column) - 3 to 4.5 ms per update in the busy device, which
// made this kernel the dominant one with the literal Householder compression (profiles/r03_bench_tsqr.json).  Now a column of
// the block is shared by Q neighbouring lanes (Q = 4 or 8: the partial dot products meet in DPP quad / half-row butterflies),
// each owning RL rows, so a block is Q * RL rows (48 at C2: what the LDS left beside the resident R allows; 128 in the fused
// small update) and the number of barrier-separated steps drops by that factor while every thread has work.
template <int Q> __device__ __forceinline__ double qr_lane_sum(double v) {
    // sum over the Q lanes of a group (Q = 4: a DPP quad, Q = 8: half a DPP row); every lane of the group gets the total
    auto bfly = [](double x, auto ctrl) {
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, decltype(ctrl)::value, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), decltype(ctrl)::value, 0xf, 0xf, true);
        return x + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    v = bfly(v, std::integral_constant<int, 0xB1>());        // quad_perm [1,0,3,2]
    v = bfly(v, std::integral_constant<int, 0x4E>());        // quad_perm [2,3,0,1]
    if (Q == 8) v = bfly(v, std::integral_constant<int, 0x141>());   // row_half_mirror: lane i <-> 7 - i of each half row
    return v;
}
template <int Q, int RL, class ColOf, class RAt>
__device__ __forceinline__ void tsqr_wide(const EkfStreamDev &S, int n1, int K, ColOf colOf, RAt Rat, double *sB) {
    constexpr int BR = Q * RL;
    const int tid = threadIdx.x, nth = blockDim.x, ld = S.ld;
    const int q = tid & (Q - 1), g = tid / Q, ng = nth / Q;
    for (int k0 = 0; k0 < K; k0 += BR) {
        __syncthreads();
        for (int e = tid; e < BR * n1; e += nth) {
            const int r = e / n1, c = e - r * n1, gk = k0 + r;
            double v = 0.0;
            if (gk < K) {
                int col, clone;
                colOf(c, col, clone);
                const unsigned long long rm = S.rowmask[gk];
                const bool on = clone < 0 ? rm != 0ULL : ((rm >> clone) & 1ULL) != 0ULL;
                if (on) v = S.Hs[(size_t)gk * ld + col];
            }
            sB[e] = v;
        }
        __syncthreads();
        double *own = sB + (size_t)q * RL * n1;          // this lane's RL rows of the block
        for (int k = 0; k < n1; ++k) {
            double bk[RL];
            double ss = 0.0;
#pragma unroll
            for (int i = 0; i < RL; ++i) { bk[i] = own[i * n1 + k]; ss += bk[i] * bk[i]; }
            ss = qr_lane_sum<Q>(ss);
            // Nothing (left) in this column of the block: structurally zero, or what earlier reflectors of this block left
            // of it.  Rounding residue shrinks by ~1e-16 per generation of reflectors; once its square leaves the normal range
            // 2 / (v0^2 + ss) overflows, and far above that it is already meaningless: below 1e-40 of R_kk^2 (or 1e-200
            // absolute) the column counts as annihilated.  Uniform: every group forms the same sum in the same order.
            const double x0 = Rat(k, k);
            if (ss < 1e-200 || ss < 1e-40 * (x0 * x0)) continue;
            const double nrm = sqrt(x0 * x0 + ss);
            const double alpha = x0 > 0.0 ? -nrm : nrm;
            const double v0 = x0 - alpha;
            const double beta = 2.0 / (v0 * v0 + ss);
            for (int j = k + 1 + g; j < n1; j += ng) {
                double bj[RL];
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RL; ++i) { bj[i] = own[i * n1 + j]; dot += bk[i] * bj[i]; }
                const double rkj = Rat(k, j);
                const double t = beta * (qr_lane_sum<Q>(dot) + v0 * rkj);
#pragma unroll
                for (int i = 0; i < RL; ++i) own[i * n1 + j] = bj[i] - t * bk[i];
                if (q == 0) Rat(k, j) = rkj - t * v0;
            }
            __syncthreads();
            if (tid == 0) Rat(k, k) = alpha;
        }
    }
    __syncthreads();
}
// rows per lane of a quad for a row block of `room` doubles beside the resident R: 16, 32 or 48 rows per block
template <class ColOf, class RAt>
__device__ __forceinline__ void tsqr_quads(const EkfStreamDev &S, int n1, int K, ColOf colOf, RAt Rat, double *sB, int room) {
    if (room >= 48 * n1) tsqr_wide<4, 12>(S, n1, K, colOf, Rat, sB);
    else if (room >= 32 * n1) tsqr_wide<4, 8>(S, n1, K, colOf, Rat, sB);
    else tsqr_wide<4, 4>(S, n1, K, colOf, Rat, sB);
}

// What the Gram factorisation kernels do for a stream BEFORE factoring (which == 0 only).  Returns 0: factor the Gram
// matrix; 1: the stack is used uncompressed and everything is set up (the kernel returns); 2: no Gram matrix was formed,
// go straight to the TSQR.
__device__ __forceinline__ int ekf_compress_entry(const EkfStreamDev &S, int *s_w /* shared, >= blockDim / 64 ints */) {
    const int na = S.rows_out[2], me = S.rows_out[1], d = S.d, ld = S.ld;
    if (na <= 0) { if (threadIdx.x == 0) S.rows_out[3] = 0; return 1; }
    if (ekf_direct_wanted(S)) {
        // The reference's m <= d case (msckf_vio.cpp:818-821): no more stacked rows than active columns, nothing to
        // compress.  The stacked rows themselves are the measurement: R = H_act (st x na, read through rowidx and the
        // rowmask by the T and S GEMMs), Q^T r = r, S is st x st.  rowidx = indices of the stacked rows in order: ranks
        // from wave ballots + a scan over the waves' counts, 512 rows per pass.
        int *rowidx = S.act + ld;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)blockDim.x >> 6;
        int base = 0;
        for (int r0 = 0; r0 < me; r0 += (int)blockDim.x) {
            const int r = r0 + tid;
            const bool on = r < me && S.rowmask[r] != 0ULL;
            const unsigned long long b = __ballot(on);
            if (lane == 0) s_w[wave] = __popcll(b);
            __syncthreads();
            int off = base, tot = 0;
            for (int w = 0; w < nw; ++w) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
            if (on && off + __popcll(b & ((1ULL << lane) - 1ULL)) < ld) rowidx[off + __popcll(b & ((1ULL << lane) - 1ULL))] = r;
            base += tot;
            __syncthreads();
        }
        // (base == rows_out[0] <= na <= ld - 21: the list fits the ld ints behind act)
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < base; i += (int)blockDim.x) S.T[(size_t)i * ld + d] = S.Hs[(size_t)rowidx[i] * ld + d];
        if (tid == 0) { S.rows_out[3] = 4; S.rows_out[4] = base; }
        return 1;
    }
    return ekf_skip_gram(S) ? 2 : 0;
}
// ... and AFTER it: Householder TSQR instead, for the streams that need it (forced, no Gram matrix formed, or the bias flag
// of the factorisation just done), handed over in the layout the update expects: S.S = L = R^T (lower, ld-wide rows),
// column d of T = Q^T r.  Rat(k, j): the resident R (zeroed here); sB: the row block, `room` doubles (>= 16 x n1).
template <class RAt>
__device__ __forceinline__ void ekf_compress_exit(const EkfStreamDev &S, bool gram_done, int diag, RAt Rat, double *sB, int room) {
    const int na = S.rows_out[2], n1 = na + 1, K = S.rows_out[1], d = S.d, ld = S.ld;
    const int tid = threadIdx.x, nth = blockDim.x;
    const bool need = ekf_mode_householder(S.qr_mode) || (S.qr_mode == 0 && (!gram_done || (diag & 2)));
    if (!need) { if (tid == 0) S.rows_out[3] = diag; return; }
    __syncthreads();
    for (int e = tid; e < n1 * n1; e += nth) { const int k = e / n1, j = e - k * n1; if (j >= k) Rat(k, j) = 0.0; }
    const int *act = S.act;
    tsqr_quads(S, n1, K, [=](int c, int &col, int &clone) { col = c < na ? act[c] : d; clone = c < na ? (col - EKF_IMU_DIM) / 6 : -1; }, Rat, sB, room);
    for (int e = tid; e < na * na; e += nth) {
        const int i = e / na, j = e - i * na;
        if (j <= i) S.S[(size_t)i * ld + j] = Rat(j, i);
    }
    for (int k = tid; k < na; k += nth) S.T[(size_t)k * ld + d] = Rat(k, na);
    if (tid == 0) S.rows_out[3] = diag | 1;
}

// ------------------------------------------------------------------------------------ Householder TSQR, row block in registers
// k_ekf_tsqr: the compression of the streams whose mode asks for the Householder path (compression_mode 2 / 3), a kernel of its
// own since the second half of round 4.  The row-block TSQR above (tsqr_wide) lives inside the factorisation kernels and keeps
// R and the row block in LDS: with mode 3 as the default EVERY update took that path, 192 workgroups of 150 KiB of LDS each for
// ~1 ms, twice per update - a CU with one of them takes no workgroup of any kernel that uses LDS (every front-end kernel does),
// and a 150 KiB request itself waits until a CU has drained completely: the front-end launches were starved for milliseconds
// (k_fe_book 20 ms, k_track4 11 ms per launch in the worst run), the rate of the default bench fell from ~100 k to 45-90 k
// frames/s from run to run.  The kernel was bound by LDS traffic (every element of the block read and written once per
// reflector), not by arithmetic.  Here
//   * a column of the block belongs to a DPP quad, lane q of the quad holding rows [q RL, (q + 1) RL) of it IN REGISTERS, a quad
//     owning the columns g, g + 128, .. (NC of them): a block is 4 RL = 128 rows (64 for the wide windows), so the dependent
//     chain of reflector steps is 2.7 x shorter than with 48-row blocks;
//   * R lives in the stream's W buffer in global memory (L2): per step a quad touches one element per column it owns, read two
//     steps ahead of its use;
//   * LDS holds two copies of one reflector (1 KiB each): the quad that owns column k + 1 applies reflector k to that column
//     first, forms reflector k + 1 at once and publishes it in the other copy while the rest of the workgroup is still applying
//     reflector k - one barrier per step.
// No LDS to speak of, so the launch shares its CUs with whatever else is running.  Arithmetic: explicit fused multiply-adds
// (this is a factorisation, not one of the decision-bearing sequences of DESIGN section 3); same skipping rule for annihilated
// columns as tsqr_wide.
#define TQ_THREADS 512
// sum over the four DPP rows of a wavefront (lanes c, c + 16, c + 32, c + 48): every one of the four gets the same bits
// (v_permlane16_swap / v_permlane32_swap of gfx950 - a VALU operation each, where a ds_bpermute pays the LDS crossbar's latency
//  twice per sum on the dependent chain of every reflector step: swapping a value with a copy of itself leaves row pairs (halves)
//  side by side in the two registers.  All four lanes of a column must be active.)
__device__ __forceinline__ double tq_rows_sum(double v) {
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false), hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        v = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));   // rows 0, 1: x0 + x1; rows 2, 3: x2 + x3
    }
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false), hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        v = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    }
    return v;
}
// 1 / x to double precision without the division sequence: hardware estimate + two Newton steps (as rsqrt_nr, chol_block.h)
__device__ __forceinline__ double tq_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// acc += x[lane C of this lane's DPP row] * a in ONE instruction (DP-ALU DPP, row_newbcast only).  Every lane of the wavefront must
// be active: a source lane that EXEC disables counts as invalid and the lanes that read it would skip the operation.
template <int C, bool FIRST> __device__ __forceinline__ void tq_fmac_bcast(double &acc, double x, double a) {
    // (FIRST: the wait states a DPP operation needs after a VALU write of EXEC - the compiler does not see a DPP instruction here)
    if (FIRST) asm("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(a), "n"(C));
    else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(a), "n"(C));
}
template <int RL, int I = 0> struct TqDot {
    static __device__ __forceinline__ void run(double (&p)[4], const double (&vr)[RL / 16], const double (&bc)[RL]) {
        tq_fmac_bcast<I % 16, I == 0>(p[I & 3], vr[I / 16], bc[I]);
        TqDot<RL, I + 1>::run(p, vr, bc);
    }
};
template <int RL> struct TqDot<RL, RL> { static __device__ __forceinline__ void run(double (&)[4], const double (&)[RL / 16], const double (&)[RL]) {} };
template <int RL, int I = 0> struct TqAxpy {
    static __device__ __forceinline__ void run(double (&bc)[RL], const double (&vr)[RL / 16], double ntt) {
        tq_fmac_bcast<I % 16, I == 0>(bc[I], vr[I / 16], ntt);
        TqAxpy<RL, I + 1>::run(bc, vr, ntt);
    }
};
template <int RL> struct TqAxpy<RL, RL> { static __device__ __forceinline__ void run(double (&)[RL], const double (&)[RL / 16], double) {} };
// the four lanes that hold column k (one per DPP row, rows [r RL, (r + 1) RL) of the block each) form reflector k and publish it
template <int RL>
__device__ __forceinline__ void tq_form_reflector(const double (&bc)[RL], double x0, double *sVs, int *sFlag_s, double *Rkk, int r) {
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int i = 0; i < RL; i += 4) { p0 = __builtin_fma(bc[i], bc[i], p0); p1 = __builtin_fma(bc[i + 1], bc[i + 1], p1); p2 = __builtin_fma(bc[i + 2], bc[i + 2], p2); p3 = __builtin_fma(bc[i + 3], bc[i + 3], p3); }
    const double ss = tq_rows_sum((p0 + p1) + (p2 + p3));
    const bool skip = ss < 1e-200 || ss < 1e-40 * (x0 * x0);           // (see tsqr_wide)
    if (!skip) {
        // (no sqrt / division sequences on the dependent chain: estimate + Newton steps, a few ulp)
        const double n2 = __builtin_fma(x0, x0, ss);
        const double nrm = n2 * rsqrt_nr(n2);
        const double alpha = x0 > 0.0 ? -nrm : nrm;
        const double v0 = x0 - alpha;
        const double beta = 2.0 * tq_rcp(__builtin_fma(v0, v0, ss));
#pragma unroll
        for (int i = 0; i < RL; i += 2) *reinterpret_cast<double2 *>(sVs + r * RL + i) = make_double2(bc[i], bc[i + 1]);
        if (r == 0) { sVs[4 * RL] = v0; sVs[4 * RL + 1] = beta; *Rkk = alpha; }
    }
    if (r == 0) *sFlag_s = skip ? 0 : 1;
}
// Lane (r, c) of wavefront w (r = DPP row 0..3, c = lane of the row): rows [r RL, (r + 1) RL) of the block, columns 16 w + c + 128 t.
template <int RL, int NC>
__device__ __forceinline__ void tsqr_regs(const EkfStreamDev &S, int na, int K, double *Rg, double *sV, int *sFlag) {
    static_assert(RL % 16 == 0, "a DPP row holds the reflector's entries of its rows, 16 per register");
    constexpr int BR = 4 * RL, NW = TQ_THREADS / 64, SLAB = 16 * NW, VS = BR + 2, VR = RL / 16;
    const int n1 = na + 1, d = S.d, ld = S.ld;
    const int tid = threadIdx.x, lane = tid & 63, r = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int col[NC], clone[NC];
    bool has[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        const int j = 16 * wave + c + SLAB * t;
        has[t] = j < n1;
        col[t] = j < na ? S.act[j] : d;
        clone[t] = j < na ? (col[t] - EKF_IMU_DIM) / 6 : -1;
    }
    for (int e = tid; e < n1 * n1; e += TQ_THREADS) { const int k = e / n1, j = e - k * n1; if (j >= k) Rg[(size_t)k * ld + j] = 0.0; }
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += BR) {
        double b[NC][RL];
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const int gk = k0 + r * RL + i;
            const unsigned long long rm = gk < K ? S.rowmask[gk] : 0ULL;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const bool on = has[t] && (clone[t] < 0 ? rm != 0ULL : ((rm >> clone[t]) & 1ULL) != 0ULL);
                b[t][i] = on ? S.Hs[(size_t)gk * ld + col[t]] : 0.0;
            }
        }
        // R at this lane's columns: row k (ra) and row k + 1 (rb) at the top of step k, row k + 2 requested during the step
        double ra[NC], rb[NC], rc[NC];
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const int j = 16 * wave + c + SLAB * t;
            ra[t] = has[t] ? Rg[j] : 0.0;
            rb[t] = has[t] && n1 > 1 ? Rg[(size_t)ld + j] : 0.0;
        }
        if (wave == 0 && c == 0) tq_form_reflector<RL>(b[0], ra[0], sV, sFlag, Rg, r);      // reflector 0
        __syncthreads();
        for (int k = 0; k < n1; ++k) {
            const int s = k & 1;
            const double *vs = sV + s * VS;
            // (the reflector's copy is read whether it is live or not - a skipped step leaves stale numbers there that nothing
            //  uses - so that the flag and the entries come back from LDS together instead of one round trip after the other)
            double vr[VR];
#pragma unroll
            for (int m = 0; m < VR; ++m) vr[m] = vs[r * RL + 16 * m + c];
            const double v0 = vs[BR], beta = vs[BR + 1];
            const bool live = __builtin_amdgcn_readfirstlane(sFlag[s]) != 0;       // (uniform: the whole workgroup reads the same word)
#pragma unroll
            for (int t = 0; t < NC; ++t) { const int j = 16 * wave + c + SLAB * t; rc[t] = (has[t] && k + 2 < n1) ? Rg[(size_t)(k + 2) * ld + j] : 0.0; }
            // reflector k onto the 16 columns of slab t of this wavefront (all 64 lanes: the DPP broadcasts need their source lanes)
            auto apply = [&](double (&bc)[RL], double rkj, int j, bool on) {
                double p[4] = {0.0, 0.0, 0.0, 0.0};
                TqDot<RL>::run(p, vr, bc);
                const double dot = tq_rows_sum((p[0] + p[1]) + (p[2] + p[3]));
                const double tt = on ? beta * __builtin_fma(v0, rkj, dot) : 0.0;
                TqAxpy<RL>::run(bc, vr, -tt);
                if (on && r == 0) Rg[(size_t)k * ld + j] = __builtin_fma(-tt, v0, rkj);
            };
            // the wavefront that holds column k + 1 takes that slab first, forms reflector k + 1 at once and publishes it in the
            // other copy while the rest of the workgroup is still applying reflector k
            const int k1 = k + 1;
            const int w1 = (k1 & (SLAB - 1)) >> 4, c1 = k1 & 15, t1 = k1 / SLAB;
            const bool own_wave = k1 < n1 && wave == w1;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                if (own_wave && t1 == t) {
                    const int j = 16 * wave + c + SLAB * t;
                    if (live) apply(b[t], ra[t], j, has[t] && j > k);
                    if (c == c1) tq_form_reflector<RL>(b[t], rb[t], sV + (s ^ 1) * VS, sFlag + (s ^ 1), Rg + (size_t)k1 * ld + k1, r);
                }
            }
            if (live) {
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    const int j0 = 16 * wave + SLAB * t;                // first column of the slab: all of it done (<= k) or beyond n1 -> nothing to do
                    if (!(own_wave && t1 == t) && j0 + 15 > k && j0 < n1) apply(b[t], ra[t], j0 + c, has[t] && j0 + c > k);
                }
            }
#pragma unroll
            for (int t = 0; t < NC; ++t) { ra[t] = rb[t]; rb[t] = rc[t]; }
            __syncthreads();
        }
    }
}
// The same scheme for ONE wavefront and a narrow stack (n1 <= 16 NC columns: the fused small update, n1 <= 25): lane (r, c) holds
// rows [r RL, (r + 1) RL) of a 4 RL-row block for the columns c + 16 t; R (n1 x n1, row-major with stride ldr) and the reflector
// (4 RL + 2 doubles, sVw) live in LDS of the wavefront's own - no workgroup barrier anywhere: the LDS operations of one wavefront
// execute in program order.  Reduces the blocks first, first + stride, .. < n_blocks into R (which the caller has zeroed or
// filled); load(block, row in block, column) delivers the entries (0 outside the stack).  Every lane of the wavefront calls it.
template <int RL, int NC, class Load>
__device__ __forceinline__ void tsqr_wave(int n1, int n_blocks, int first, int stride, Load load, double *R, int ldr, double *sVw, int *sFlagw) {
    constexpr int BR = 4 * RL, VR = RL / 16;
    const int lane = threadIdx.x & 63, r = lane >> 4, c = lane & 15;
    for (int blk = first; blk < n_blocks; blk += stride) {
        double b[NC][RL];
#pragma unroll
        for (int i = 0; i < RL; ++i) {
#pragma unroll
            for (int t = 0; t < NC; ++t) b[t][i] = load(blk, r * RL + i, c + 16 * t);
        }
        for (int k = 0; k < n1; ++k) {
            const int tk = k >> 4, ck = k & 15;
#pragma unroll
            for (int t = 0; t < NC; ++t)
                if (tk == t && c == ck) tq_form_reflector<RL>(b[t], R[k * ldr + k], sVw, sFlagw, R + k * ldr + k, r);
            __builtin_amdgcn_wave_barrier();
            double vr[VR];
#pragma unroll
            for (int m = 0; m < VR; ++m) vr[m] = sVw[r * RL + 16 * m + c];
            const double v0 = sVw[BR], beta = sVw[BR + 1];
            const bool live = __builtin_amdgcn_readfirstlane(*sFlagw) != 0;
            if (live) {
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    if (16 * t + 15 > k && 16 * t < n1) {           // (uniform: the slab still has columns right of k)
                        const int j = c + 16 * t;
                        const bool on = j < n1 && j > k;
                        const double rkj = on ? R[k * ldr + j] : 0.0;
                        double p[4] = {0.0, 0.0, 0.0, 0.0};
                        TqDot<RL>::run(p, vr, b[t]);
                        const double dot = tq_rows_sum((p[0] + p[1]) + (p[2] + p[3]));
                        const double tt = on ? beta * __builtin_fma(v0, rkj, dot) : 0.0;
                        TqAxpy<RL>::run(b[t], vr, -tt);
                        if (on && r == 0) R[k * ldr + j] = __builtin_fma(-tt, v0, rkj);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
// do_cap: this is the first dense kernel of the route (no stream of the batch needs a Gram matrix, so k_ekf_gemm<GRAM> was not
// launched): the stacking decision (ekf_cap.h) is worked out and written down here.
template <int RL, int NC, int WPE>
__global__ __launch_bounds__(TQ_THREADS, WPE) void k_ekf_tsqr(const EkfStreamDev *streams, int do_cap) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0 || (S.route & EKF_ROUTE_SMALL)) return;
    if (!ekf_mode_householder(S.qr_mode)) return;                 // the Gram route's streams: k_ekf_chol_lds / k_ekf_chol (which = 0)
    if (do_cap) { (void)ekf_cap_local<TQ_THREADS>(S, true); __syncthreads(); }
    __shared__ int s_w[TQ_THREADS / 64];
    __shared__ __attribute__((aligned(16))) double sV[2 * (4 * RL + 2)];
    __shared__ int sFlag[2];
    if (ekf_compress_entry(S, s_w) != 2) return;                  // nothing active, or the stack is used uncompressed (all set up)
    const int na = S.rows_out[2], K = S.rows_out[1], d = S.d, ld = S.ld;
    const int tid = threadIdx.x;
    if (na + 1 > (TQ_THREADS / 4) * NC) { if (tid == 0) S.rows_out[3] = 0x80; return; }      // (cannot happen: the launcher picks NC from the window)
    double *Rg = S.W;
    tsqr_regs<RL, NC>(S, na, K, Rg, sV, sFlag);
    // hand over in the layout the update expects: S.S = L = R^T (lower, ld-wide rows), column d of T = Q^T r
    for (int e = tid; e < na * na; e += TQ_THREADS) {
        const int i = e / na, j = e - i * na;
        if (j <= i) S.S[(size_t)i * ld + j] = Rg[(size_t)j * ld + i];
    }
    for (int k = tid; k < na; k += TQ_THREADS) S.T[(size_t)k * ld + d] = Rg[(size_t)k * ld + na];
    if (tid == 0) S.rows_out[3] = 1;
}

// ------------------------------------------------------------------------------------ Cholesky
// Fallback for active blocks that do not fit LDS (more than CHOL_LDS_MAX_ROWS rows: 50- and 60-clone windows): the same
// blocked factorisation (chol_block.h) run in place on the row-major global matrix through a generic pointer — diagonal
// block in registers, panel solve per row, trailing update on the matrix cores — with only the 16-wide panel in LDS.
// A row is read up to 15 entries right of its diagonal as filler: those are the mirrored upper triangle and, for the
// last rows, columns [n, n+16) of the ld-wide row (ld - n >= 21: the IMU columns are not part of the compact block).
#define CHOLG_THREADS 512
__global__ __launch_bounds__(CHOLG_THREADS) void k_ekf_chol(const EkfStreamDev *streams, int which, int pan_rs) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0 || (S.route & EKF_ROUTE_SMALL)) return;
    int entry = 0;
    __shared__ int s_w[CHOLG_THREADS / 64];
    if (which == 0) {
        if (ekf_mode_householder(S.qr_mode)) return;      // compressed by k_ekf_tsqr (launched beside this kernel when the batch has such streams)
        entry = ekf_compress_entry(S, s_w);
        if (entry == 1) return;
    }
    const int n = which == 0 ? S.rows_out[2] : S.rows_out[4];   // active columns (Gram) / rows of the compressed measurement (S)
    const int nt = n + (which == 0 ? 1 : 0);              // + the extra Q^T r row of the Gram factorisation
    const int lda = S.ld;
    double *A = which == 0 ? S.S : S.W;                   // G_c = [H_act|r]^T [H_act|r]  /  S = T[:, act] R^T + sigma^2 I
    extern __shared__ double s_dyn[];
    double *sPanT = s_dyn;                                // [LNB][pan_rs]
    __shared__ double s_tol, s_mx[CHOLG_THREADS / 64];
    __shared__ int s_diag;
    __shared__ CholBlockShared s_cb;
    const int tid = threadIdx.x;
    if (tid == 0) s_diag = 0;
    for (int e = tid; e < LNB * pan_rs; e += CHOLG_THREADS) sPanT[e] = 0.0;
    if (which == 0) {
        // regularised factorisation G + lambda I, lambda = 1e-14 d max(diag G) (see k_ekf_chol_lds)
        double mx = 0;
        for (int i = tid; i < n; i += CHOLG_THREADS) mx = fmax(mx, A[(size_t)i * lda + i]);
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) { double m = 0; for (int i = 0; i < CHOLG_THREADS / 64; ++i) m = fmax(m, s_mx[i]); s_tol = m * (double)S.d * 1e-14; }
        __syncthreads();
        const double lam = s_tol;
        for (int i = tid; i < n; i += CHOLG_THREADS) A[(size_t)i * lda + i] += lam;
    }
    __syncthreads();
    chol_blocked_lds<CHOLG_THREADS / 64>(A, [lda](int i, int j) { return i * lda + j; }, n, nt, 0.0, sPanT, pan_rs, s_cb);
    if (which == 0) {
        // pivots (L_kk^2) that end within 100 lambda of the regularisation floor (reported in the diagnostics only)
        int tiny = 0;
        for (int i = tid; i < n; i += CHOLG_THREADS) { const double l = A[(size_t)i * lda + i]; tiny += (l * l < 100.0 * s_tol) ? 1 : 0; }
        for (int o = 32; o > 0; o >>= 1) tiny += __shfl_xor(tiny, o);
        __shared__ int s_tiny[CHOLG_THREADS / 64];
        if ((tid & 63) == 0) s_tiny[tid >> 6] = tiny;
        __syncthreads();
        // predicted relative bias of the posterior covariance from the lambda prior: lambda max(P_aa) / sigma^2
        double pm = 0;
        for (int i = tid; i < n; i += CHOLG_THREADS) pm = fmax(pm, S.P[(size_t)S.act[i] * lda + S.act[i]]);
        for (int o = 32; o > 0; o >>= 1) pm = fmax(pm, __shfl_xor(pm, o));
        __syncthreads();
        if ((tid & 63) == 0) s_mx[tid >> 6] = pm;
        __syncthreads();
        if (tid == 0) {
            int t = 0; for (int i = 0; i < CHOLG_THREADS / 64; ++i) t += s_tiny[i];
            double p = 0; for (int i = 0; i < CHOLG_THREADS / 64; ++i) p = fmax(p, s_mx[i]);
            s_diag = (t << 8) | ((s_tol * p > QR_BIAS_LIMIT * S.sigma2) ? 2 : 0);
        }
        // column d of T <- (Q^T r) = the extra row of L, so the TRSM carries w = L2^-1 Q^T r along
        for (int k = tid; k < n; k += CHOLG_THREADS) S.T[(size_t)k * lda + S.d] = A[(size_t)n * lda + k];
        __syncthreads();
        double *Rg = S.W;
        ekf_compress_exit(S, true, s_diag, [=](int k, int j) -> double & { return Rg[(size_t)k * lda + j]; }, sPanT, LNB * pan_rs);
    }
}

// ------------------------------------------------------------------------------------ LDS-resident Cholesky
// Same factorisation with the active block held in LDS in packed lower form (row i at i(i+1)/2).
// Structure used: the 21 IMU columns of every stacked Jacobian are zero (H_x only touches clone columns,
// msckf_vio.cpp:698,713), so rows/cols [0,21) of G = [H|r]^T[H|r] are exactly zero (pivots skipped, L = 0)
// and rows/cols [0,21) of S = T R^T + sigma^2 I are sigma^2 I (L = sigma I).  Only the trailing
// (d-21) x (d-21) block is factorised: <= 180 rows for 30 clones = 129 KiB packed + a 16-wide panel.
//
// Right-looking, 16 columns per panel, one 512-thread workgroup (8 waves) per stream:
//   1. wave 0 factors the 16x16 diagonal block in registers: lane r owns row r, the pivot and the column
//      entries travel through v_mov_b64_dpp row_newbcast (no LDS round trips, no barriers inside the block); 1/sqrt(pivot)
//      comes from v_rsq_f64 + two Newton steps, so the serial chain has no division;
//   2. one thread per row below solves x L11^T = a against L11 broadcast the same way and leaves the panel k-major
//      (sPanT[c][row]) for the MFMA operands;
//   3. the trailing update A22 -= X X^T runs on v_mfma_f64_16x16x4_f64, one 16x16 tile of the lower triangle at
//      a time per wave (4 MFMAs per tile), read-modify-write on the packed matrix.
#define CHOL_LDS_MAX_ROWS 181     // active rows incl. the extra Q^T r row
#define CHOL_THREADS 512          // 8 waves: 256 VGPRs per lane, the unrolled 16-column eliminations stay in registers
#define CHOL_WAVES (CHOL_THREADS / 64)
#define CHOL_PAN_RS 192           // row stride of the k-major panel (rows padded to a multiple of 16)

__device__ __forceinline__ int pk(int i, int j) { return i * (i + 1) / 2 + j; }

__global__ __launch_bounds__(CHOL_THREADS) void k_ekf_chol_lds(const EkfStreamDev *streams, int which, int lds_doubles) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0 || (S.route & EKF_ROUTE_SMALL)) return;
    int entry = 0;
    __shared__ int s_w[CHOL_WAVES];
    if (which == 0) {
        if (ekf_mode_householder(S.qr_mode)) return;      // compressed by k_ekf_tsqr (launched beside this kernel when the batch has such streams)
        entry = ekf_compress_entry(S, s_w);
        if (entry == 1) return;
    }
    double *A = which == 0 ? S.S : S.W;
    const int off = 0, lda = S.ld;                 // compact storage: index i <-> column act[i]
    const int n = which == 0 ? S.rows_out[2] : S.rows_out[4];   // active columns (Gram) / rows of the compressed measurement (S)
    const int nt = n + (which == 0 ? 1 : 0);       // rows incl. the extra row
    const bool semidef = which == 0;
    extern __shared__ double s_dyn[];
    double *sM = s_dyn;                            // packed lower, nt rows
    double *sPanT = s_dyn + nt * (nt + 1) / 2;     // [LNB][CHOL_PAN_RS]
    __shared__ double s_tol, s_mx[CHOL_WAVES];
    __shared__ int s_diag;
    __shared__ CholBlockShared s_cb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_diag = 0;
    // R of the TSQR lives where the packed factor does (packed upper by rows: the same n1 (n1 + 1) / 2 doubles), the row block in the panel's space
    auto Rl = [=](int k, int j) -> double & { return sM[k * nt - k * (k - 1) / 2 + (j - k)]; };
    const int room = lds_doubles - nt * (nt + 1) / 2;       // what the launch's LDS leaves beside the packed factor / the resident R: the TSQR's row block
    // load: one matrix row per wave pass, coalesced along j
    // (8 rows x 3 column chunks = up to 24 loads in flight per lane: the copy is latency bound otherwise)
    for (int i0 = wave * 8; i0 < nt; i0 += CHOL_WAVES * 8) {
        double v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u;
            const double *src = A + (size_t)(off + i) * lda + off;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int j = lane + 64 * k; v[u][k] = (i < nt && j <= i) ? src[j] : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u;
            if (i >= nt) continue;
            double *dst = sM + pk(i, 0);
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int j = lane + 64 * k; if (j <= i) dst[j] = v[u][k]; }
        }
    }
    for (int e = tid; e < LNB * CHOL_PAN_RS; e += CHOL_THREADS) sPanT[e] = 0.0;
    __syncthreads();
    if (semidef) {
        // G = H^T H is rank deficient (unobserved clones, the gauge) and its small pivots are rounding noise:
        // an unpivoted Cholesky that skips or keeps them by a threshold is unstable (measured: up to 7e-5
        // relative error in P on few-feature updates, tools/dev/update_truth.py).  Factor G + lambda I instead,
        // lambda = 1e-14 d max(diag G): every pivot stays above the noise, nothing is skipped, and the only effect
        // is a prior of weight lambda / sigma^2 on the clone states (bias ~1e-11 relative, same tool).
        double mx = 0;
        for (int i = tid; i < n; i += CHOL_THREADS) mx = fmax(mx, sM[pk(i, i)]);
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if (lane == 0) s_mx[wave] = mx;
        __syncthreads();
        if (tid == 0) { double m = 0; for (int i = 0; i < CHOL_WAVES; ++i) m = fmax(m, s_mx[i]); s_tol = m * (double)S.d * 1e-14; }
        __syncthreads();
        const double lam = s_tol;
        for (int i = tid; i < n; i += CHOL_THREADS) sM[pk(i, i)] += lam;
        __syncthreads();
    }
    const double tol = 0.0;      // a pivot <= 0 (cannot happen for G + lambda I or for S >= sigma^2 I) zeroes its column
    chol_blocked_lds<CHOL_WAVES>(sM, [](int i, int j) { return pk(i, j); }, n, nt, tol, sPanT, CHOL_PAN_RS, s_cb);
    if (which == 0) {
        // pivots (L_kk^2) that end within 100 lambda of the regularisation floor (reported in the diagnostics only)
        int tiny = 0;
        for (int i = tid; i < n; i += CHOL_THREADS) { const double l = sM[pk(i, i)]; tiny += (l * l < 100.0 * s_tol) ? 1 : 0; }
        for (int o = 32; o > 0; o >>= 1) tiny += __shfl_xor(tiny, o);
        __shared__ int s_tiny[CHOL_WAVES];
        if (lane == 0) s_tiny[wave] = tiny;
        __syncthreads();
        // predicted relative bias of the posterior covariance from the lambda prior: lambda max(P_aa) / sigma^2
        double pm = 0;
        for (int i = tid; i < n; i += CHOL_THREADS) pm = fmax(pm, S.P[(size_t)S.act[i] * lda + S.act[i]]);
        for (int o = 32; o > 0; o >>= 1) pm = fmax(pm, __shfl_xor(pm, o));
        __syncthreads();
        if (lane == 0) s_mx[wave] = pm;
        __syncthreads();
        if (tid == 0) {
            int t = 0; for (int i = 0; i < CHOL_WAVES; ++i) t += s_tiny[i];
            double p = 0; for (int i = 0; i < CHOL_WAVES; ++i) p = fmax(p, s_mx[i]);
            s_diag = (t << 8) | ((s_tol * p > QR_BIAS_LIMIT * S.sigma2) ? 2 : 0);
        }
    }
    // store back
    for (int i = wave; i < nt; i += CHOL_WAVES) {
        double *dst = A + (size_t)(off + i) * lda + off;
        const double *src = sM + pk(i, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int j = lane + 64 * k; if (j <= i && j < n) dst[j] = src[j]; }
    }
    if (which == 0) {
        // column d of T <- (Q^T r) = the extra row of L, so the TRSM carries w = L2^-1 Q^T r along
        for (int k = tid; k < n; k += CHOL_THREADS) S.T[(size_t)k * lda + S.d] = sM[pk(n, k)];
        __syncthreads();
        ekf_compress_exit(S, true, s_diag, Rl, sPanT, room);
    }
}

// ------------------------------------------------------------------------------------ TRSM
// Y = L^-1 B in place, L = lower Cholesky factor in S.W (na x na), B = S.T (na x (d+1)).  One workgroup per
// 32-column strip; the strip (na x 32 doubles) stays in LDS for the whole solve, L is staged 16 rows at a time.
#define QR_MAX_N1 (6 * 64 + 1)     // active columns of a 64-clone window + the residual
#define TS_COLS 32
#define TS_RB 16
__global__ __launch_bounds__(256) void k_ekf_trsm(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0 || (S.route & EKF_ROUTE_SMALL)) return;
    const int n = S.rows_out[4], ld = S.ld, ncols = S.d + 1;      // rows of the compressed measurement, all d+1 columns
    const int c0 = blockIdx.x * TS_COLS;
    if (c0 >= ncols) return;
    const double *L = S.W;
    double *B = S.T;
    extern __shared__ double s_dyn[];
    double *sL = s_dyn;                              // [TS_RB][n + 1] row block of L
    double *sY = s_dyn + (size_t)TS_RB * (n + 1);    // [n][TS_COLS] the strip
    const int tid = threadIdx.x, c = tid & 31, r8 = tid >> 5;
    for (int e = tid; e < n * TS_COLS; e += 256) {
        const int i = e >> 5, cc = e & 31;
        sY[e] = (c0 + cc < ncols) ? B[(size_t)i * ld + c0 + cc] : 0.0;
    }
    // L row blocks are prefetched one block ahead into registers: 16 threads per row, columns lr, lr + 16, ...
    constexpr int LPF = 14;                       // ceil((max d + 16) / 16) for d <= 208
    const int lrow = tid >> 4, lr = tid & 15;
    double lpf[LPF];
    auto fetch_L = [&](int ib) {
        const int nb = min(TS_RB, n - ib), wcols = ib + nb;
#pragma unroll
        for (int q = 0; q < LPF; ++q) {
            const int p = lr + 16 * q;
            lpf[q] = (lrow < nb && p < wcols) ? L[(size_t)(ib + lrow) * ld + p] : 0.0;
        }
    };
    const bool l_fits = n + TS_RB <= 16 * LPF;    // else fall back to direct staging
    if (l_fits) fetch_L(0);
    for (int ib = 0; ib < n; ib += TS_RB) {
        const int nb = min(TS_RB, n - ib);
        __syncthreads();
        if (l_fits) {
#pragma unroll
            for (int q = 0; q < LPF; ++q) {
                const int p = lr + 16 * q;
                if (lrow < nb && p < ib + nb) sL[(size_t)lrow * (n + 1) + p] = lpf[q];
            }
            if (ib + TS_RB < n) fetch_L(ib + TS_RB);
        } else {
            for (int e = tid; e < nb * (ib + nb); e += 256) {
                const int i = e / (ib + nb), p = e - i * (ib + nb);
                sL[(size_t)i * (n + 1) + p] = L[(size_t)(ib + i) * ld + p];
            }
        }
        __syncthreads();
        // GEMM part: rows r8 and r8 + 8 of the block against all previous rows of the strip
        double acc0 = 0.0, acc1 = 0.0;
        {
            const double *l0 = sL + (size_t)r8 * (n + 1), *l1 = sL + (size_t)(r8 + 8) * (n + 1);
            for (int p = 0; p < ib; ++p) {
                const double y = sY[p * TS_COLS + c];
                acc0 += l0[p] * y;
                acc1 += l1[p] * y;
            }
        }
        if (r8 < nb) sY[(ib + r8) * TS_COLS + c] -= acc0;
        if (r8 + 8 < nb) sY[(ib + r8 + 8) * TS_COLS + c] -= acc1;
        __syncthreads();
        // in-block forward substitution, one thread per column
        if (tid < TS_COLS) {
            double x[TS_RB];
#pragma unroll
            for (int j = 0; j < TS_RB; ++j) {
                x[j] = 0.0;
                if (j < nb) {
                    double s2 = sY[(ib + j) * TS_COLS + tid];
                    const double *lj = sL + (size_t)j * (n + 1) + ib;
#pragma unroll
                    for (int cc = 0; cc < j; ++cc) s2 -= lj[cc] * x[cc];
                    x[j] = (lj[j] != 0.0) ? s2 / lj[j] : 0.0;
                    sY[(ib + j) * TS_COLS + tid] = x[j];
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < n * TS_COLS; e += 256) {
        const int i = e >> 5, cc = e & 31;
        if (c0 + cc < ncols) B[(size_t)i * ld + c0 + cc] = sY[e];
    }
}

// ------------------------------------------------------------------------------------ small update, fused
// The whole measurement update of a stream in ONE workgroup when the stack touches few clones (na <= SU_MAX_NA active
// columns): the pruning update (msckf_vio.cpp:1073-1184) stacks hundreds of 2-observation features but only the two
// clones being removed, i.e. na = 12.  The general path spends seven launches (Gram, factor, T, S, factor, solve,
// downdate) on 13 x 13 and 12 x 12 matrices there; here the compression, the gain and the covariance downdate run
// back to back out of LDS.  Same algebra as the general path: G = [H_act|r]^T [H_act|r] + lambda I = L L^T,
// T = L^T[0:na,0:na] P[act,:], S = T[:,act] L[0:na,0:na] + sigma^2 I = L2 L2^T, Y = L2^-1 [T | Q^T r],
// delta_x = Y^T w, P -= Y^T Y (symmetric by construction).  All sums run in a fixed order (no atomics).
#define SU_MAX_NA 24
#define SU_CH 128         // stacked rows per Gram chunk
__global__ __launch_bounds__(256) void k_ekf_small_update(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0 || !(S.route & EKF_ROUTE_SMALL)) return;
    const EkfCapResult cap = ekf_cap_local(S, true);         // the first dense kernel of this route: which blocks are stacked (ekf_cap.h)
    const int d = S.d, ld = S.ld, na = cap.na, n1 = na + 1, K = cap.end;
    const int tid = threadIdx.x;
    if (na <= 0 || na > SU_MAX_NA) {            // nothing stacked (or a caller error): no correction, P unchanged
        for (int c = tid; c < d; c += 256) S.delta_x[c] = 0.0;
        return;
    }
    extern __shared__ double s_dyn[];
    double *sT = s_dyn;                          // na x (d + 1): T, then Y
    double *sG = sT + (size_t)na * (d + 1);      // n1 x n1: Gram matrix, then its factor L (lower)
    double *sS = sG + n1 * n1;                   // na x na: S, then its factor L2 (lower)
    double *sC = sS + na * na;                   // SU_CH x n1 chunk of [H_act | r]
    __shared__ int s_col[SU_MAX_NA + 1], s_clone[SU_MAX_NA + 1];
    __shared__ double s_lam;
    if (tid < n1) {
        const int col = tid < na ? ekf_act_column(cap.clones, tid) : d;
        s_col[tid] = col;
        s_clone[tid] = tid < na ? (col - EKF_IMU_DIM) / 6 : -1;
    }
    __syncthreads();
    // ---- 1. Gram matrix: pairs (i >= j) over the threads; with few pairs (the pruning update: 91) the threads also split
    //         the rows of a chunk into `groups` interleaved sets, summed in a fixed order at the end
    // (the Householder modes never use the Gram matrix: steps 1 and 2 are skipped for them and the TSQR of step 2b runs at once)
    const bool hh_only = ekf_mode_householder(S.qr_mode);
    if (tid == 0) s_lam = 0.0;
    if (!hh_only) {
    const int np = n1 * (n1 + 1) / 2;
    const int groups = np <= 128 ? 256 / np : 1;
    int pi[2], pj[2], pg[2];
    double acc[2] = {0.0, 0.0};
    for (int q = 0; q < 2; ++q) {
        int p = tid + 256 * q, grp = 0;
        if (groups > 1) { grp = q == 0 ? tid / np : groups; p = tid - grp * np; }
        pg[q] = (groups > 1) ? (grp < groups ? grp : -1) : (p < np ? 0 : -1);
        if (pg[q] < 0) p = 0;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= p) ++i;
        pi[q] = i; pj[q] = p - i * (i + 1) / 2;
    }
    for (int k0 = 0; k0 < K; k0 += SU_CH) {
        __syncthreads();
        for (int e = tid; e < SU_CH * n1; e += 256) {
            const int r = e / n1, c = e - r * n1, gk = k0 + r;
            double v = 0.0;
            if (gk < K) {
                const unsigned long long rm = S.rowmask[gk];
                const bool on = s_clone[c] < 0 ? rm != 0ULL : ((rm >> s_clone[c]) & 1ULL) != 0ULL;
                if (on) v = S.Hs[(size_t)gk * ld + s_col[c]];
            }
            sC[e] = v;
        }
        __syncthreads();
        for (int q = 0; q < 2; ++q) {
            if (pg[q] < 0) continue;
            double a = acc[q];
            for (int r = pg[q]; r < SU_CH; r += groups) a += sC[r * n1 + pi[q]] * sC[r * n1 + pj[q]];
            acc[q] = a;
        }
    }
    __syncthreads();
    if (groups > 1) {
        // sC is free now: partial sums [group][pair], reduced by group 0's threads in group order
        if (pg[0] >= 0) sC[pg[0] * np + (pi[0] * (pi[0] + 1) / 2 + pj[0])] = acc[0];
        __syncthreads();
        if (pg[0] == 0) {
            const int p = pi[0] * (pi[0] + 1) / 2 + pj[0];
            double t = 0.0;
            for (int g2 = 0; g2 < groups; ++g2) t += sC[g2 * np + p];
            sG[pi[0] * n1 + pj[0]] = t; sG[pj[0] * n1 + pi[0]] = t;
        }
    } else {
        for (int q = 0; q < 2; ++q)
            if (pg[q] >= 0) { sG[pi[q] * n1 + pj[q]] = acc[q]; sG[pj[q] * n1 + pi[q]] = acc[q]; }
    }
    __syncthreads();
    // ---- 2. G + lambda I = L L^T (lambda as in k_ekf_chol_lds), the Q^T r row rides along as row na
    if (tid == 0) { double mx = 0; for (int i = 0; i < na; ++i) mx = fmax(mx, sG[i * n1 + i]); s_lam = mx * (double)d * 1e-14; }
    __syncthreads();
    if (tid < na) sG[tid * n1 + tid] += s_lam;
    __syncthreads();
    for (int k = 0; k < na; ++k) {
        const double pv = sG[k * n1 + k];
        const double inv = pv > 0.0 ? 1.0 / sqrt(pv) : 0.0;
        __syncthreads();
        if (tid >= k && tid < n1) sG[tid * n1 + k] = (tid == k) ? (pv > 0.0 ? sqrt(pv) : 0.0) : sG[tid * n1 + k] * inv;
        __syncthreads();
        for (int e = tid; e < (n1 - k - 1) * (n1 - k - 1); e += 256) {
            const int i = k + 1 + e / (n1 - k - 1), j = k + 1 + e % (n1 - k - 1);
            if (j <= i) sG[i * n1 + j] -= sG[i * n1 + k] * sG[j * n1 + k];
        }
        __syncthreads();
    }
    }   // !hh_only
    // ---- 2b. the same decision as ekf_compress_exit: bias flag, no more stacked rows than columns, or forced ->
    //          Householder TSQR of the stacked rows, R in sG (full rows), then L = R^T back in place
    {
        __shared__ int s_tiny, s_bias;
        if (tid == 0 && hh_only) { s_tiny = 0; s_bias = 0; }
        if (tid == 0 && !hh_only) {
            int t = 0;
            double pm = 0;
            for (int i = 0; i < na; ++i) {
                const double l = sG[i * n1 + i];
                t += (l * l < 100.0 * s_lam) ? 1 : 0;
                pm = fmax(pm, S.P[(size_t)s_col[i] * ld + s_col[i]]);
            }
            s_tiny = t;
            s_bias = (s_lam * pm > QR_BIAS_LIMIT * S.sigma2) ? 1 : 0;
        }
        __syncthreads();
        const int tiny = s_tiny;
        const bool need_qr = ekf_mode_householder(S.qr_mode) || (S.qr_mode == 0 && (cap.stacked <= na || s_bias));
        if (need_qr) {
            for (int e = tid; e < n1 * n1; e += 256) sG[e] = 0.0;
            __syncthreads();
            // Round 4, second half: a TSQR TREE over the four wavefronts.  The stack is tall and narrow here (the pruning update: up to
            // 1500 rows x 13 columns): every wavefront reduces its own 128-row blocks (w, w + 4, ..) into an R of its own with the
            // row block in registers (tsqr_wave: no workgroup barrier), then wavefront 0 annihilates the other three R's (3 n1 rows:
            // one more block) against its own, which is sG.  (Before: tsqr_wide<8, 16>, all 256 threads on one block at a time with a
            // workgroup barrier per reflector and the block in LDS: 250-320 us of the kernel's 310-470.)
            static_assert(SU_CH == 128 && SU_MAX_NA + 1 <= 32, "tsqr_wave<32, 2>: 128-row blocks, two 16-column slabs");
            {
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                double *sRw = sC;                                   // [3][n1][n1]: the R's of wavefronts 1..3
                double *sVq = (double *)(((uintptr_t)(sC + 3 * n1 * n1) + 15) & ~(uintptr_t)15);    // [4][130]: the wavefronts' reflector buffers (16-byte accesses)
                int *sFq = (int *)(sVq + 4 * 130);                  // [4]
                for (int e = tid; e < 3 * n1 * n1; e += 256) sRw[e] = 0.0;
                __syncthreads();
                const unsigned long long *rowmask = S.rowmask;
                const double *Hs = S.Hs;
                auto loadH = [&](int blk, int row, int col) -> double {
                    const int gk = blk * 128 + row;
                    if (gk >= K || col >= n1) return 0.0;
                    const unsigned long long rm = rowmask[gk];
                    const int cl = s_clone[col];
                    const bool on = cl < 0 ? rm != 0ULL : ((rm >> cl) & 1ULL) != 0ULL;
                    return on ? Hs[(size_t)gk * ld + s_col[col]] : 0.0;
                };
                double *Rmine = wv == 0 ? sG : sRw + (wv - 1) * n1 * n1;
                tsqr_wave<32, 2>(n1, (K + 127) / 128, wv, 4, loadH, Rmine, n1, sVq + wv * 130, sFq + wv);
                __syncthreads();
                if (wv == 0) {
                    auto loadR = [&](int, int row, int col) -> double {
                        if (row >= 3 * n1 || col >= n1) return 0.0;
                        const int w2 = row / n1, rr = row - w2 * n1;
                        return col >= rr ? sRw[(w2 * n1 + rr) * n1 + col] : 0.0;
                    };
                    tsqr_wave<32, 2>(n1, 1, 0, 1, loadR, sG, n1, sVq, sFq);
                }
                __syncthreads();
            }
            for (int e = tid; e < n1 * n1; e += 256) { const int i = e / n1, j = e - i * n1; if (j < i) sG[e] = sG[j * n1 + i]; }
            __syncthreads();
        }
        if (tid == 0) S.rows_out[3] = (tiny << 8) | (s_bias ? 2 : 0) | (need_qr ? 1 : 0);
    }
    // ---- 3. T = R P[act, :] with R = L^T (upper): T[i][c] = sum_{k >= i} L[k][i] P[act[k]][c];  T[i][d] = (Q^T r)_i = L[na][i]
    const double *P = S.P;
    for (int c = tid; c < d; c += 256) {
        double pc[SU_MAX_NA];
#pragma unroll
        for (int k = 0; k < SU_MAX_NA; ++k) pc[k] = k < na ? P[(size_t)s_col[k] * ld + c] : 0.0;
#pragma unroll
        for (int i = 0; i < SU_MAX_NA; ++i) {
            if (i < na) {
                double t = 0.0;
#pragma unroll
                for (int k = i; k < SU_MAX_NA; ++k) if (k < na) t += sG[k * n1 + i] * pc[k];
                sT[i * (d + 1) + c] = t;
            }
        }
    }
    if (tid < na) sT[tid * (d + 1) + d] = sG[na * n1 + tid];
    __syncthreads();
    // ---- 4. S = T[:, act] R^T + sigma^2 I (lower, mirrored): S[i][j] = sum_{k >= j} T[i][act[k]] L[k][j]
    for (int e = tid; e < na * na; e += 256) {
        const int i = e / na, j = e - i * na;
        if (j > i) continue;
        double t = 0.0;
        for (int k = j; k < na; ++k) t += sT[i * (d + 1) + s_col[k]] * sG[k * n1 + j];
        if (i == j) t += S.sigma2;
        sS[i * na + j] = t; sS[j * na + i] = t;
    }
    __syncthreads();
    // ---- 5. S = L2 L2^T
    for (int k = 0; k < na; ++k) {
        const double pv = sS[k * na + k];
        const double inv = pv > 0.0 ? 1.0 / sqrt(pv) : 0.0;
        __syncthreads();
        if (tid >= k && tid < na) sS[tid * na + k] = (tid == k) ? (pv > 0.0 ? sqrt(pv) : 0.0) : sS[tid * na + k] * inv;
        __syncthreads();
        for (int e = tid; e < (na - k - 1) * (na - k - 1); e += 256) {
            const int i = k + 1 + e / (na - k - 1), j = k + 1 + e % (na - k - 1);
            if (j <= i) sS[i * na + j] -= sS[i * na + k] * sS[j * na + k];
        }
        __syncthreads();
    }
    // ---- 6. Y = L2^-1 [T | Q^T r], one thread per column
    for (int c = tid; c <= d; c += 256) {
        double y[SU_MAX_NA];
#pragma unroll
        for (int i = 0; i < SU_MAX_NA; ++i) {
            y[i] = 0.0;
            if (i >= na) continue;
            double t = sT[i * (d + 1) + c];
#pragma unroll
            for (int k = 0; k < i; ++k) t -= sS[i * na + k] * y[k];
            const double l = sS[i * na + i];
            y[i] = l != 0.0 ? t / l : 0.0;
            sT[i * (d + 1) + c] = y[i];
        }
    }
    __syncthreads();
    // ---- 7. Y goes to the stream's T buffer (ld-wide rows): delta_x = Y^T w (msckf_vio.cpp:860) and P <- P - Y^T Y
    //         (:897-904) are the tile-parallel GM_PUPD launch that follows
    for (int e = tid; e < na * (d + 1); e += 256) {
        const int k = e / (d + 1), c = e - k * (d + 1);
        S.T[(size_t)k * ld + c] = sT[e];
    }
}

extern "C" {
void ekf_launch_gemm(const EkfStreamDev *d, int n, int mode, int max_mn, hipStream_t st) {
    const int t = (max_mn + GT - 1) / GT;
    const dim3 grid(t * t + (mode == GM_PUPD ? t : 0), n);
    switch (mode) {
        case GM_GRAM: hipLaunchKernelGGL(k_ekf_gemm<GM_GRAM>, grid, dim3(256), 0, st, d); break;
        case GM_T:    hipLaunchKernelGGL(k_ekf_gemm<GM_T>, grid, dim3(256), 0, st, d); break;
        case GM_S2:   hipLaunchKernelGGL(k_ekf_gemm<GM_S2>, grid, dim3(256), 0, st, d); break;
        default:      hipLaunchKernelGGL(k_ekf_gemm<GM_PUPD>, grid, dim3(256), 0, st, d); break;
    }
}
// Householder compression of the streams in compression_mode 2 / 3 (the others leave at once)
void ekf_launch_tsqr(const EkfStreamDev *d, int n, int max_d, int do_cap, hipStream_t st) {
    const int n1 = max_d - EKF_IMU_DIM + 1;
    // (a 64-row block at 128 VGPRs - four waves per SIMD left for others - measured 1270 us alone against 810 and the same bench rate)
    if (n1 <= 2 * (TQ_THREADS / 4)) hipLaunchKernelGGL((k_ekf_tsqr<32, 2, 2>), dim3(1, n), dim3(TQ_THREADS), 0, st, d, do_cap);
    else hipLaunchKernelGGL((k_ekf_tsqr<16, 4, 2>), dim3(1, n), dim3(TQ_THREADS), 0, st, d, do_cap);
}
void ekf_launch_chol(const EkfStreamDev *d, int n, int which, int max_d, hipStream_t st) {
    const int nt = max_d - EKF_IMU_DIM + 1;
    if (nt <= CHOL_LDS_MAX_ROWS) {
        static std::once_flag attr_once;
    std::call_once(attr_once, []() { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_chol_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(((size_t)CHOL_LDS_MAX_ROWS * (CHOL_LDS_MAX_ROWS + 1) / 2 + (size_t)CHOL_PAN_RS * LNB) * sizeof(double))); });
        const size_t lds = ((size_t)nt * (nt + 1) / 2 + (size_t)CHOL_PAN_RS * LNB) * sizeof(double);
        hipLaunchKernelGGL(k_ekf_chol_lds, dim3(1, n), dim3(CHOL_THREADS), lds, st, d, which, (int)(lds / sizeof(double)));
        return;
    }
    const int pan_rs = (nt + 15) / 16 * 16;
    hipLaunchKernelGGL(k_ekf_chol, dim3(1, n), dim3(CHOLG_THREADS), (size_t)LNB * pan_rs * sizeof(double), st, d, which, pan_rs);
}
void ekf_launch_small_update(const EkfStreamDev *d, int n, int max_d, hipStream_t st) {
    const size_t lds = ((size_t)SU_MAX_NA * (max_d + 1) + (SU_MAX_NA + 1) * (SU_MAX_NA + 1) + SU_MAX_NA * SU_MAX_NA + SU_CH * (SU_MAX_NA + 1)) * sizeof(double);
    static std::once_flag attr_once;
    std::call_once(attr_once, []() {
        // the whole CU's LDS less what the kernel declares statically (a request beyond that is refused, and the refusal would
        // surface as the "last error" of a later, innocent launch)
        hipFuncAttributes fa;
        size_t stat = 4096;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_ekf_small_update)) == hipSuccess) stat = fa.sharedSizeBytes;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_small_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - stat)) != hipSuccess) (void)hipGetLastError();
    });
    hipLaunchKernelGGL(k_ekf_small_update, dim3(1, n), dim3(256), lds, st, d);
}
int ekf_small_update_max_na(void) { return SU_MAX_NA; }
void ekf_launch_trsm(const EkfStreamDev *d, int n, int max_d, hipStream_t st) {
    const int strips = (max_d + 1 + TS_COLS - 1) / TS_COLS;
    // the solve runs over the active rows only: n <= max_d - 21 (the IMU columns are never active).  A full 64-clone
    // window (d = 405, n <= 384) needs 144 KiB; the limit is raised to the whole 160 KiB of a gfx950 CU.
    const int max_n = max_d > EKF_IMU_DIM ? max_d - EKF_IMU_DIM : 1;
    const size_t lds = (size_t)(TS_RB * (max_n + 1) + (size_t)max_n * TS_COLS) * sizeof(double);
    static std::once_flag attr_once;
    std::call_once(attr_once, []() { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(k_ekf_trsm, dim3(strips, n), dim3(256), lds, st, d);
}
}
