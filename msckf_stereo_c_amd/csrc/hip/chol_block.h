// chol_block.h — blocked right-looking Cholesky of a symmetric matrix held in LDS (lower triangle, rows
// contiguous), shared by k_ekf_chol_lds (Gram / innovation covariance of the update, ekf_linalg.hip) and the
// chi-square gate of k_ekf_feature_blocks (ekf_kernels.hip).  16 columns per panel:
//   1. wave 0 factors the 16x16 diagonal block in registers: lane r owns row r, the pivot and the column
//      entries travel through v_mov_b64_dpp row_newbcast (no LDS round trips, no barriers inside the block);
//      1/sqrt(pivot) comes from v_rsq_f64 + two Newton steps, so the serial chain has no division;
//   2. one thread per row below solves x L11^T = a against L11 broadcast the same way and leaves the panel
//      k-major (sPanT[c][row]) for the MFMA operands;
//   3. the trailing update A22 -= X X^T runs on v_mfma_f64_16x16x4_f64, one 16x16 tile of the lower triangle
//      at a time per wave (4 MFMAs per tile), read-modify-write on the LDS matrix.
// Rows [n, nt) are extra rows that ride along (panel solves and trailing updates, never pivots): after the
// factorisation such a row r holds (L^-1 r)^T.
#pragma once
#include <hip/hip_runtime.h>

typedef double cb_v4f64 __attribute__((ext_vector_type(4)));

#define LNB 16

// value of lane C of each 16-lane row, in every lane of that row: one v_mov_b64_dpp row_newbcast
template <int C> __device__ __forceinline__ double row_bcast_f64(double v) {
    long long b = __double_as_longlong(v);
    b = __builtin_amdgcn_update_dpp(b, b, 0x150 + C, 0xf, 0xf, false);
    return __longlong_as_double(b);
}

// 1/sqrt(x) to double precision without a division: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// Column-by-column elimination with the matrix rows spread over the lanes of a 16-lane row (lane r = row r, a[c]
// = entry (r, c)); unrolled by template recursion because the DPP lane selectors are immediates.
template <int J, int C> __device__ __forceinline__ void chol_rank1(double (&a)[LNB]) {
    if constexpr (C < LNB) {
        a[C] = fma(-a[J], row_bcast_f64<C>(a[J]), a[C]);      // rows r < C carry unused upper-triangle values
        chol_rank1<J, C + 1>(a);
    }
}
template <int J> __device__ __forceinline__ void chol_diag_cols(double (&a)[LNB], int r, double tol, double &invd) {
    if constexpr (J < LNB) {
        const double piv = row_bcast_f64<J>(a[J]);
        const bool skip = !(piv > tol);
        const double y = skip ? 0.0 : rsqrt_nr(piv);
        double l = piv * y;
        l = skip ? 0.0 : fma(0.5 * y, fma(-l, l, piv), l);
        if (r == J) { a[J] = l; invd = y; } else a[J] *= y;
        chol_rank1<J, J + 1>(a);
        chol_diag_cols<J + 1>(a, r, tol, invd);
    }
}
// x L11^T = a for one matrix row per lane; lane c of every 16-lane row holds row c of L11 (Lr) and 1/L11[c][c]
template <int J, int C> __device__ __forceinline__ void panel_elim(double (&x)[LNB], double lj) {
    if constexpr (C < LNB) {
        x[C] = fma(-x[J], row_bcast_f64<C>(lj), x[C]);
        panel_elim<J, C + 1>(x, lj);
    }
}
template <int J> __device__ __forceinline__ void panel_cols(double (&x)[LNB], const double (&Lr)[LNB], double invr) {
    if constexpr (J < LNB) {
        x[J] *= row_bcast_f64<J>(invr);
        // L11 does not depend on x, so the optimiser would materialise all 120 broadcasts up front (240 VGPRs, spills):
        // tie column J's source to x[J] so its broadcasts are formed when they are consumed
        double lj = Lr[J];
        asm volatile("" : "+v"(lj) : "v"(x[J]));
        panel_elim<J, J + 1>(x, lj);
        panel_cols<J + 1>(x, Lr, invr);
    }
}


struct CholBlockShared {
    double L11[LNB][LNB + 1];    // L11[j][c] = L11(c, j)
    double inv[LNB];
};

// AT(i, j): index of element (i, j), j <= i, in sM; a row must be contiguous in j and readable (finite filler)
// for 15 entries past its diagonal (the last row needs 16 doubles of slack behind it).  sPanT: LNB x pan_rs doubles, pan_rs >= nt rounded up to 16, zero-initialised
// by the caller.  All NWAVES*64 threads of the workgroup must call; pivots <= tol zero their column.
template <int NWAVES, class AT>
__device__ __forceinline__ void chol_blocked_lds(double *sM, AT at, int n, int nt, double tol, double *sPanT, int pan_rs,
                                                 CholBlockShared &sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int kb = 0; kb < n; kb += LNB) {
        const int nb = min(LNB, n - kb);
        // ---- 1. diagonal block in the registers of wave 0 (rows >= nb are padded with whatever follows in LDS)
        if (wave == 0) {
            int r = lane & 15;
            asm volatile("" : "+v"(r));      // per-panel value: keeps 16 x 16 lane predicates from being hoisted and spilled
            // Entries right of the diagonal (and rows past nb in the last panel) only ever feed other
            // upper-triangle / padded entries, never the factor, and are masked on the way out.
            double a[LNB];
            {
                const double *src = sM + at(min(kb + r, nt - 1), kb);     // padded lanes re-read the last row
#pragma unroll
                for (int c = 0; c < LNB; ++c) a[c] = src[c];
            }
            double invd = 0.0;     // lane r keeps 1/L[r][r]
            chol_diag_cols<0>(a, r, tol, invd);
            if (lane < LNB) {
#pragma unroll
                for (int c = 0; c < LNB; ++c) {
                    const bool in = c <= r && r < nb;
                    if (in) sM[at(kb + r, kb + c)] = a[c];
                    sh.L11[c][r] = in ? a[c] : 0.0;
                }
                sh.inv[r] = r < nb ? invd : 0.0;
            }
        }
        __syncthreads();
        // ---- 2. panel rows below: x L11^T = a, one thread per row
        const int r0 = kb + nb;
        const int rem = nt - r0;
        if (wave * 64 < rem) {           // wave-uniform: all 64 lanes take part in the broadcasts
            const int rr = lane & 15;
            double Lr[LNB];
#pragma unroll
            for (int c = 0; c < LNB; ++c) Lr[c] = sh.L11[c][rr];
            const double invr = sh.inv[rr];
            const bool has_row = tid < rem;
            const int i = r0 + (has_row ? tid : 0);
            double *row = sM + at(i, kb);
            double x[LNB];
#pragma unroll
            for (int c = 0; c < LNB; ++c) x[c] = row[c];        // columns >= nb: finite filler, multiplied by 0 below
            panel_cols<0>(x, Lr, invr);
            if (has_row) {
#pragma unroll
                for (int c = 0; c < LNB; ++c) {
                    if (c < nb) row[c] = x[c];
                    sPanT[c * pan_rs + tid] = x[c];
                }
            }
        }
        __syncthreads();
        // ---- 3. trailing update on the matrix cores: 16x16 tiles (ta, tb <= ta) of rows/cols r0 + ...
        const int ntr = (rem + 15) >> 4;
        const int n_tiles = ntr * (ntr + 1) / 2;
        {
            // tiles are numbered row-major over the lower triangle; a wave takes tiles wave, wave + NWAVES, ...
            int ta = 0, tb = wave;
            while (tb > ta) { tb -= ta + 1; ++ta; }
            for (int t = wave; t < n_tiles; t += NWAVES) {
                cb_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s4 = 0; s4 < LNB / 4; ++s4) {
                    const double *pp = sPanT + (4 * s4 + (lane >> 4)) * pan_rs + (lane & 15);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pp[16 * ta], pp[16 * tb], acc, 0, 0, 0);
                }
                const int j = r0 + 16 * tb + (lane & 15);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = r0 + 16 * ta + (lane >> 4) + 4 * q;
                    if (i < nt && j <= i) sM[at(i, j)] -= acc[q];
                }
                tb += NWAVES;
                while (tb > ta) { tb -= ta + 1; ++ta; }
            }
        }
        __syncthreads();
    }
}
