// ekf_linalg.hip — batched dense FP64 building blocks of the Kalman update (msckf_vio.cpp:795-904),
// one job per VIO stream, blockIdx.y = stream of the batch:
//
//  k_ekf_gemm   : 32x32 output tile per workgroup, v_mfma_f64_16x16x4_f64 (one 16x16 sub-tile per wave,
//                 K staged through LDS 16 at a time).  Modes:
//                   GRAM  G  = [Hs|rs]^T [Hs|rs]          ((d+1)x(d+1), replaces the QR: G = R^T R, last row = (Q^T r)^T R)
//                   T     T  = R P,  R = L^T upper        (skips the zero half of R)
//                   S2    S  = T R^T + sigma^2 I          (symmetric, lower computed and mirrored)
//                   PUPD  P <- P - Y^T Y                  (symmetric)
//  k_ekf_chol   : blocked (NB = 16) right-looking Cholesky in place, one workgroup per stream; the
//                 semidefinite variant skips pivots <= tol (zero IMU columns / gauge directions of H^T H)
//                 and carries extra rows (the Q^T r row) through the panel solves only.
//  k_ekf_trsm   : Y = L^-1 [T | Q^T r] in place, one workgroup per 32-column strip and stream.
//  k_ekf_dx     : delta_x = Y^T w (w = column d of Y).
//
// Why Gram + Cholesky instead of Householder QR: the update only needs R^T R = H^T H and R^T (Q^T r) =
// H^T r; forming them is one GEMM-shaped pass over the stacked Jacobian (MFMA-friendly, fully parallel
// over output tiles) instead of d sequential reflector applications.  H^T H is rank deficient (the 21
// IMU columns of H are zero, the gauge is unobservable): skipped pivots drop exactly those directions.
#include <mutex>
#include "ekf_device.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

enum { GM_GRAM = 0, GM_T = 1, GM_S2 = 2, GM_PUPD = 3 };

struct GemmArgs {
    const double *A, *B;
    double *C;
    int M, N, K, lda, ldb, ldc;
    int transA, transB;     // op(A)(i,k) = transA ? A[k*lda+i] : A[i*lda+k] ; op(B)(k,j) = transB ? B[j*ldb+k] : B[k*ldb+j]
    int sym;                // compute tiles with j0 <= i0 only, mirror
    int kmin_i, kmin_j;     // op(A)(i,k) == 0 for k < i  /  op(B)(k,j) == 0 for k < j
    double alpha, beta, diag_add;
};

__device__ __forceinline__ bool gemm_setup(const EkfStreamDev &S, int mode, GemmArgs &g) {
    const int d = S.d, ld = S.ld;
    g.sym = 0; g.kmin_i = 0; g.kmin_j = 0; g.alpha = 1.0; g.beta = 0.0; g.diag_add = 0.0;
    g.lda = g.ldb = g.ldc = ld;
    switch (mode) {
        case GM_GRAM:
            g.A = S.Hs; g.B = S.Hs; g.C = S.S; g.M = g.N = d + 1; g.K = S.rows_out[1]; g.transA = 1; g.transB = 0; g.sym = 1;
            return true;
        case GM_T:
            g.A = S.S; g.B = S.P; g.C = S.T; g.M = d; g.N = d; g.K = d; g.transA = 1; g.transB = 0; g.kmin_i = 1;
            return true;
        case GM_S2:
            g.A = S.T; g.B = S.S; g.C = S.W; g.M = d; g.N = d; g.K = d; g.transA = 0; g.transB = 0; g.sym = 1; g.kmin_j = 1;
            g.diag_add = S.sigma2;
            return true;
        case GM_PUPD:
            g.A = S.T; g.B = S.T; g.C = S.P; g.M = d; g.N = d; g.K = d; g.transA = 1; g.transB = 0; g.sym = 1;
            g.alpha = -1.0; g.beta = 1.0;
            return true;
    }
    return false;
}

#define GT 32      // output tile edge
#define GK 16      // K per LDS stage

__global__ __launch_bounds__(256) void k_ekf_gemm(const EkfStreamDev *streams, int mode) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0) return;
    GemmArgs g;
    gemm_setup(S, mode, g);
    const int tiles_n = (g.N + GT - 1) / GT, tiles_m = (g.M + GT - 1) / GT;
    const int tile = blockIdx.x;
    if (mode == GM_PUPD && tile == tiles_m * tiles_n) {
        // one extra workgroup: delta_x = Y^T w, w = column d of Y (msckf_vio.cpp:860)
        const int d = S.d, ld = S.ld;
        for (int c = threadIdx.x; c < d; c += 256) {
            double s2 = 0;
            for (int k = 0; k < d; ++k) s2 += S.T[(size_t)k * ld + c] * S.T[(size_t)k * ld + d];
            S.delta_x[c] = s2;
        }
        return;
    }
    if (tile >= tiles_m * tiles_n) return;
    const int ti = tile / tiles_n, tj = tile - ti * tiles_n;
    if (g.sym && tj > ti) return;
    const int i0 = ti * GT, j0 = tj * GT;
    __shared__ double sA[GK][GT + 1];   // sA[k][i]
    __shared__ double sB[GK][GT + 1];   // sB[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    int k_begin = 0;
    if (g.kmin_i) k_begin = (i0 / GK) * GK;
    if (g.kmin_j) { const int kb = (j0 / GK) * GK; k_begin = k_begin > kb ? k_begin : kb; }
    if (g.kmin_i && g.kmin_j) { /* both given: the max above is already right */ }
    for (int k0 = k_begin; k0 < g.K; k0 += GK) {
        __syncthreads();
        // stage op(A)[i0.., k0..] and op(B)[k0.., j0..]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int ii, kk;
            if (g.transA) { ii = tid & 31; kk = (tid >> 5) + 8 * e; }
            else { kk = tid & 15; ii = (tid >> 4) + 16 * e; }
            const int gi = i0 + ii, gk = k0 + kk;
            double v = 0.0;
            if (gi < g.M && gk < g.K && !(g.kmin_i && gk < gi)) v = g.transA ? g.A[(size_t)gk * g.lda + gi] : g.A[(size_t)gi * g.lda + gk];
            sA[kk][ii] = v;
            int jj, kb;
            if (g.transB) { kb = tid & 15; jj = (tid >> 4) + 16 * e; }
            else { jj = tid & 31; kb = (tid >> 5) + 8 * e; }
            const int gj = j0 + jj, gkb = k0 + kb;
            double w = 0.0;
            if (gj < g.N && gkb < g.K && !(g.kmin_j && gkb < gj)) w = g.transB ? g.B[(size_t)gj * g.ldb + gkb] : g.B[(size_t)gkb * g.ldb + gj];
            sB[kb][jj] = w;
        }
        __syncthreads();
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
        if (i >= g.M || j >= g.N) continue;
        if (g.sym && j > i) continue;             // diagonal tiles: lower part only, mirrored below
        double v = g.alpha * acc[r];
        if (g.beta != 0.0) v += g.beta * g.C[(size_t)i * g.ldc + j];
        if (i == j) v += g.diag_add;
        g.C[(size_t)i * g.ldc + j] = v;
        if (g.sym && i != j) g.C[(size_t)j * g.ldc + i] = v;
    }
}

// ------------------------------------------------------------------------------------ Cholesky
#define CNB 16
struct CholArgs { double *A; int n, n_extra, lda, semidef; };

// in-place lower Cholesky of the leading n x n block; rows [n, n+n_extra) only take part in the panel solves
__global__ __launch_bounds__(256) void k_ekf_chol(const EkfStreamDev *streams, int which) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0) return;
    CholArgs c;
    if (which == 0) { c.A = S.S; c.n = S.d; c.n_extra = 1; c.lda = S.ld; c.semidef = 1; }   // G = [H|r]^T [H|r]
    else { c.A = S.W; c.n = S.d; c.n_extra = 0; c.lda = S.ld; c.semidef = 0; }               // S = T R^T + sigma^2 I
    const int n = c.n, nt = c.n + c.n_extra, lda = c.lda;
    double *A = c.A;
    extern __shared__ double s_dyn[];
    double *sD = s_dyn;                  // [CNB][CNB+1]
    double *sPanel = s_dyn + CNB * (CNB + 1);   // [nt][CNB]
    __shared__ double s_tol;
    const int tid = threadIdx.x;
    if (c.semidef) {
        // tolerance relative to the largest diagonal entry
        double mx = 0;
        for (int i = tid; i < n; i += 256) mx = fmax(mx, A[(size_t)i * lda + i]);
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        __shared__ double s_mx[4];
        if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) s_tol = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3])) * (double)n * 2.220446049250313e-16;
    } else if (tid == 0) s_tol = 0.0;
    __syncthreads();
    const double tol = s_tol;
    for (int kb = 0; kb < n; kb += CNB) {
        const int nb = min(CNB, n - kb);
        // 1. diagonal block
        __syncthreads();
        for (int e = tid; e < CNB * CNB; e += 256) {
            const int i = e / CNB, j = e % CNB;
            sD[i * (CNB + 1) + j] = (i < nb && j <= i) ? A[(size_t)(kb + i) * lda + kb + j] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {
            for (int j = 0; j < nb; ++j) {
                const double piv = sD[j * (CNB + 1) + j];
                const bool skip = !(piv > tol);
                const double l = skip ? 0.0 : sqrt(piv);
                const double inv = skip ? 0.0 : 1.0 / l;
                __builtin_amdgcn_wave_barrier();
                if (tid >= j && tid < nb) sD[tid * (CNB + 1) + j] = (tid == j) ? l : sD[tid * (CNB + 1) + j] * inv;
                __builtin_amdgcn_wave_barrier();
                // trailing update inside the block: (i, c) with j < c <= i < nb
                for (int e = tid; e < CNB * CNB; e += 64) {
                    const int i = e / CNB, cc = e % CNB;
                    if (cc > j && cc <= i && i < nb) sD[i * (CNB + 1) + cc] -= sD[i * (CNB + 1) + j] * sD[cc * (CNB + 1) + j];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int e = tid; e < CNB * CNB; e += 256) {
            const int i = e / CNB, j = e % CNB;
            if (i < nb && j <= i) A[(size_t)(kb + i) * lda + kb + j] = sD[i * (CNB + 1) + j];
        }
        // 2. panel below: x L11^T = a, one thread per row
        const int r0 = kb + nb;
        for (int i = r0 + tid; i < nt; i += 256) {
            double x[CNB];
#pragma unroll
            for (int j = 0; j < CNB; ++j) {
                double s = (j < nb) ? A[(size_t)i * lda + kb + j] : 0.0;
#pragma unroll
                for (int cc = 0; cc < j; ++cc) s -= x[cc] * sD[j * (CNB + 1) + cc];
                const double ljj = sD[j * (CNB + 1) + j];
                x[j] = (j < nb && ljj != 0.0) ? s / ljj : 0.0;
            }
#pragma unroll
            for (int j = 0; j < CNB; ++j) {
                if (j < nb) A[(size_t)i * lda + kb + j] = x[j];
                sPanel[(size_t)(i - r0) * CNB + j] = x[j];
            }
        }
        __syncthreads();
        // 3. trailing update A22 -= L21 L21^T (lower part; extra rows against all columns < n)
        const int rem = nt - r0;
        const int remc = n - r0;           // columns that still get factored
        for (int e = tid; e < rem * remc; e += 256) {
            const int a = e / remc, b = e - a * remc;
            if (b > a) continue;
            const double *pa = sPanel + (size_t)a * CNB, *pb = sPanel + (size_t)b * CNB;
            double s = 0;
#pragma unroll
            for (int cc = 0; cc < CNB; ++cc) s += pa[cc] * pb[cc];
            A[(size_t)(r0 + a) * lda + r0 + b] -= s;
        }
    }
    if (which == 0) {
        __syncthreads();
        for (int k = tid; k < S.d; k += 256) S.T[(size_t)k * S.ld + S.d] = A[(size_t)S.d * lda + k];
    }
}

// ------------------------------------------------------------------------------------ LDS-resident Cholesky
// Same factorisation with the active block held in LDS in packed lower form (row i at i(i+1)/2).
// Structure used: the 21 IMU columns of every stacked Jacobian are zero (H_x only touches clone columns,
// msckf_vio.cpp:698,713), so rows/cols [0,21) of G = [H|r]^T[H|r] are exactly zero (pivots skipped, L = 0)
// and rows/cols [0,21) of S = T R^T + sigma^2 I are sigma^2 I (L = sigma I).  Only the trailing
// (d-21) x (d-21) block is factorised: <= 180 rows for 30 clones = 127 KiB packed + an 8-wide panel.
#define LNB 8
#define CHOL_LDS_MAX_ROWS 181     // active rows incl. the extra Q^T r row
__global__ __launch_bounds__(1024) void k_ekf_chol_lds(const EkfStreamDev *streams, int which) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0) return;
    double *A = which == 0 ? S.S : S.W;
    const int off = EKF_IMU_DIM, lda = S.ld;
    const int n = S.d - off;                       // active columns
    const int nt = n + (which == 0 ? 1 : 0);       // rows incl. the extra row
    const bool semidef = which == 0;
    extern __shared__ double s_dyn[];
    double *sM = s_dyn;                            // packed lower, nt rows
    double *sPan = s_dyn + (size_t)nt * (nt + 1) / 2;   // [nt][LNB]
    __shared__ double s_tol, s_mx[16];
    const int tid = threadIdx.x;
    // load (row-wise, coalesced along j)
    for (int e = tid; e < nt * nt; e += 1024) {
        const int i = e / nt, j = e - i * nt;
        if (j <= i) sM[(size_t)i * (i + 1) / 2 + j] = A[(size_t)(off + i) * lda + off + j];
    }
    __syncthreads();
    if (semidef) {
        double mx = 0;
        for (int i = tid; i < n; i += 1024) mx = fmax(mx, sM[(size_t)i * (i + 1) / 2 + i]);
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) { double m = 0; for (int i = 0; i < 16; ++i) m = fmax(m, s_mx[i]); s_tol = m * (double)S.d * 2.220446049250313e-16; }
    } else if (tid == 0) s_tol = 0.0;
    __syncthreads();
    const double tol = s_tol;
    for (int kb = 0; kb < n; kb += LNB) {
        const int nb = min(LNB, n - kb);
        // 1. diagonal block, in place in sM, by wave 0
        if (tid < 64) {
            for (int j = 0; j < nb; ++j) {
                const double piv = sM[(size_t)(kb + j) * (kb + j + 1) / 2 + kb + j];
                const bool skip = !(piv > tol);
                const double l = skip ? 0.0 : sqrt(piv);
                const double inv = skip ? 0.0 : 1.0 / l;
                __builtin_amdgcn_wave_barrier();
                if (tid >= j && tid < nb) {
                    double *p = &sM[(size_t)(kb + tid) * (kb + tid + 1) / 2 + kb + j];
                    *p = (tid == j) ? l : (*p) * inv;
                }
                __builtin_amdgcn_wave_barrier();
                // (i, c) with j < c <= i < nb
                if (tid < LNB * LNB) {
                    const int i = tid / LNB, cc = tid % LNB;
                    if (cc > j && cc <= i && i < nb) {
                        const size_t ri = (size_t)(kb + i) * (kb + i + 1) / 2 + kb, rc = (size_t)(kb + cc) * (kb + cc + 1) / 2 + kb;
                        sM[ri + cc] -= sM[ri + j] * sM[rc + j];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        // 2. panel rows below: x L11^T = a
        const int r0 = kb + nb;
        for (int i = r0 + tid; i < nt; i += 1024) {
            double *row = &sM[(size_t)i * (i + 1) / 2 + kb];
            double x[LNB];
#pragma unroll
            for (int j = 0; j < LNB; ++j) {
                double s2 = (j < nb) ? row[j] : 0.0;
                const size_t rj = (size_t)(kb + j) * (kb + j + 1) / 2 + kb;
#pragma unroll
                for (int cc = 0; cc < j; ++cc) s2 -= x[cc] * ((j < nb) ? sM[rj + cc] : 0.0);
                const double ljj = (j < nb) ? sM[rj + j] : 0.0;
                x[j] = (j < nb && ljj != 0.0) ? s2 / ljj : 0.0;
            }
#pragma unroll
            for (int j = 0; j < LNB; ++j) { if (j < nb) row[j] = x[j]; sPan[(size_t)j * nt + (i - r0)] = x[j]; }
        }
        __syncthreads();
        // 3. trailing update (lower part; the extra row only against columns < n)
        const int rem = nt - r0, remc = n - r0;
        for (int e = tid; e < rem * remc; e += 1024) {
            const int a = e / remc, b = e - a * remc;
            if (b > a) continue;
            double s2 = 0;
#pragma unroll
            for (int cc = 0; cc < LNB; ++cc) s2 += sPan[(size_t)cc * nt + a] * sPan[(size_t)cc * nt + b];
            sM[(size_t)(r0 + a) * (r0 + a + 1) / 2 + r0 + b] -= s2;
        }
        __syncthreads();
    }
    // store back; the trivial IMU block: L = 0 (Gram) or sigma I (S)
    for (int e = tid; e < nt * nt; e += 1024) {
        const int i = e / nt, j = e - i * nt;
        if (j <= i && j < n) A[(size_t)(off + i) * lda + off + j] = sM[(size_t)i * (i + 1) / 2 + j];
    }
    if (which == 0) {
        // column d of T <- (Q^T r) = row d of L, so the TRSM carries w = L2^-1 Q^T r along (IMU part is zero)
        for (int k = tid; k < S.d; k += 1024) S.T[(size_t)k * lda + S.d] = (k < off) ? 0.0 : sM[(size_t)n * (n + 1) / 2 + (k - off)];
    }
    const double l0 = semidef ? 0.0 : sqrt(S.sigma2);
    const int rows_all = S.d + (which == 0 ? 1 : 0);
    for (int e = tid; e < rows_all * off; e += 1024) {
        const int i = e / off, j = e - i * off;
        if (j <= i) A[(size_t)i * lda + j] = (i == j) ? l0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------ r_thin column
// column d of T <- row d of L (= (Q^T r)^T), so the TRSM carries w = L2^-1 Q^T r along
__global__ __launch_bounds__(256) void k_ekf_rthin(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0) return;
    for (int k = threadIdx.x; k < S.d; k += 256) S.T[(size_t)k * S.ld + S.d] = S.S[(size_t)S.d * S.ld + k];
}

// ------------------------------------------------------------------------------------ TRSM
// Y = L^-1 B in place, L = lower Cholesky factor in S.W (d x d), B = S.T (d x (d+1)).  One workgroup per
// 32-column strip; the strip (d x 32 doubles) stays in LDS for the whole solve, L is staged 16 rows at a time.
#define TS_COLS 32
#define TS_RB 16
__global__ __launch_bounds__(256) void k_ekf_trsm(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    if (S.n_feat <= 0) return;
    const int n = S.d, ld = S.ld, ncols = S.d + 1;
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
    for (int ib = 0; ib < n; ib += TS_RB) {
        const int nb = min(TS_RB, n - ib);
        __syncthreads();
        for (int e = tid; e < nb * (ib + nb); e += 256) {
            const int i = e / (ib + nb), p = e - i * (ib + nb);
            sL[(size_t)i * (n + 1) + p] = L[(size_t)(ib + i) * ld + p];
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

// ------------------------------------------------------------------------------------ delta_x
__global__ __launch_bounds__(256) void k_ekf_dx(const EkfStreamDev *streams) {
    const EkfStreamDev &S = streams[blockIdx.y];
    const int d = S.d, ld = S.ld;
    if (S.n_feat <= 0) return;
    for (int c = threadIdx.x; c < d; c += 256) {
        double s = 0;
        for (int k = 0; k < d; ++k) s += S.T[(size_t)k * ld + c] * S.T[(size_t)k * ld + d];
        S.delta_x[c] = s;
    }
}

extern "C" {
void ekf_launch_gemm(const EkfStreamDev *d, int n, int mode, int max_mn, hipStream_t st) {
    const int t = (max_mn + GT - 1) / GT;
    hipLaunchKernelGGL(k_ekf_gemm, dim3(t * t + (mode == GM_PUPD ? 1 : 0), n), dim3(256), 0, st, d, mode);
}
void ekf_launch_chol(const EkfStreamDev *d, int n, int which, int max_d, hipStream_t st) {
    const int nt = max_d - EKF_IMU_DIM + 1;
    if (nt <= CHOL_LDS_MAX_ROWS) {
        static std::once_flag attr_once;
    std::call_once(attr_once, []() { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_chol_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(((size_t)CHOL_LDS_MAX_ROWS * (CHOL_LDS_MAX_ROWS + 1) / 2 + (size_t)CHOL_LDS_MAX_ROWS * LNB) * sizeof(double))); });
        const size_t lds = ((size_t)nt * (nt + 1) / 2 + (size_t)nt * LNB) * sizeof(double);
        hipLaunchKernelGGL(k_ekf_chol_lds, dim3(1, n), dim3(1024), lds, st, d, which);
        return;
    }
    const size_t lds = (size_t)(CNB * (CNB + 1) + (size_t)(max_d + 2) * CNB) * sizeof(double);
    hipLaunchKernelGGL(k_ekf_chol, dim3(1, n), dim3(256), lds, st, d, which);
}
void ekf_launch_rthin(const EkfStreamDev *d, int n, hipStream_t st) { hipLaunchKernelGGL(k_ekf_rthin, dim3(1, n), dim3(256), 0, st, d); }
void ekf_launch_trsm(const EkfStreamDev *d, int n, int max_d, hipStream_t st) {
    const int strips = (max_d + 1 + TS_COLS - 1) / TS_COLS;
    const size_t lds = (size_t)(TS_RB * (max_d + 1) + (size_t)max_d * TS_COLS) * sizeof(double);
    static std::once_flag attr_once;
    std::call_once(attr_once, []() { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ekf_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
    hipLaunchKernelGGL(k_ekf_trsm, dim3(strips, n), dim3(256), lds, st, d);
}
void ekf_launch_dx(const EkfStreamDev *d, int n, hipStream_t st) { hipLaunchKernelGGL(k_ekf_dx, dim3(1, n), dim3(256), 0, st, d); }
}
