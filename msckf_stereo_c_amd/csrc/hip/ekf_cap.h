// ekf_cap.h — which blocks of an update are stacked (msckf_vio.cpp:1002-1010), as a routine the first dense kernel of an
// update runs itself.
//
// Rounds 1-3 ran this as a kernel of its own (k_ekf_cap, one workgroup per stream) between the feature kernels and the
// Gram pass: 17 us of work that cost 300 us as one more launch of the update's chain in the busy device.  Now
//   * the feature kernels write the rowmask of their own rows (what a row carries: the clone bits of a block that passed
//     the gate, 0 otherwise), and
//   * every workgroup of the FIRST dense kernel of the stream's route - k_ekf_gemm<GRAM> (its tiles), k_ekf_small_update, or
//     k_ekf_tsqr when no stream of the batch needs a Gram matrix (then the Gram launch does not happen at all) -
//     computes the stacking decision for itself from the features' gate results: the passing blocks are stacked in feature
//     order until the stacked rows exceed the cap (the block that crosses it is still stacked, :1006-1009), which gives the
//     stacked row count, the end of the last stacked block (the K range of the Gram pass), the clones the stack touches
//     (active columns) and the first feature behind the cap.  Integer work on a few hundred features: every workgroup gets
//     the same answer, no workgroup waits for another.  ONE workgroup per stream (publish = true) also writes it down for
//     the kernels that follow: rows_out[0..4], the active column list, and the gate bits / masks of the capped features.
// All NT threads of the workgroup call it (256; 512 in k_ekf_tsqr).
#pragma once
#include "ekf_device.h"

struct EkfCapResult {
    int stacked;                  // rows_out[0]: rows of the stacked blocks
    int end;                      // rows_out[1]: end of the last stacked block (rows beyond it carry nothing)
    int na;                       // rows_out[2]: active columns = 6 x clones the stack touches
    unsigned long long clones;    // those clones (bit c)
};

// column of compact index i: the (i % 6)-th column of the (i / 6)-th clone of `clones`
__device__ __forceinline__ int ekf_act_column(unsigned long long clones, int i) {
    int q = i / 6;
    unsigned long long m = clones;
    while (q-- > 0) m &= m - 1;                    // drop the q lowest set bits
    return EKF_IMU_DIM + 6 * (int)__builtin_ctzll(m) + i % 6;
}

template <int NT = 256>          // threads of the calling workgroup
__device__ __forceinline__ EkfCapResult ekf_cap_local(const EkfStreamDev &S, bool publish) {
    __shared__ int s_sum[NT], s_cross[NT];
    __shared__ unsigned long long s_or[NT / 64];
    __shared__ int s_cap_from, s_tot[2];
    __shared__ unsigned long long s_mask;
    const int tid = threadIdx.x, nf = S.n_feat;
    const int per = (nf + NT - 1) / NT;
    const int j0 = tid * per, j1 = min(nf, j0 + per);
    __syncthreads();                                // (the scratch may still be read from an earlier call)
    int local = 0;
    for (int j = j0; j < j1; ++j) if (S.feat_status[j] & 2) local += 4 * S.feats[j].n_obs - 3;
    s_sum[tid] = local;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {        // inclusive scan of the per-thread sums
        const int v = tid >= off ? s_sum[tid - off] : 0;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    const int before = s_sum[tid] - local;
    int cross = nf;                                 // first feature of this thread's run whose inclusive stacked-row count exceeds the cap
    if (S.apply_row_cap) {
        int run = before;
        for (int j = j0; j < j1; ++j)
            if (S.feat_status[j] & 2) { run += 4 * S.feats[j].n_obs - 3; if (run > S.max_stack_rows) { cross = j; break; } }
    }
    s_cross[tid] = cross;
    __syncthreads();
    if (tid == 0) {
        int c = nf;
        for (int t = 0; t < NT; ++t) if (s_cross[t] < c) c = s_cross[t];
        s_cap_from = c < nf ? c + 1 : nf;           // features [cap_from, nf) are not stacked
    }
    __syncthreads();
    const int cap_from = s_cap_from;
    int stack = 0, meff = 0;
    unsigned long long orm = 0ULL;
    for (int j = j0; j < j1; ++j) {
        EkfFeatDev &F = S.feats[j];
        const int n = 4 * F.n_obs - 3;
        const bool pass = (S.feat_status[j] & 2) != 0;
        if (pass && j < cap_from) { stack += n; meff = F.row_off + n; orm |= F.colmask; }
        else if (pass && publish) {
            // behind the cap: the block loses its gate bit and its rows their masks (nothing reads them: they lie beyond the
            // last stacked block; cleared for the record the host and the diagnostics read)
            F.colmask = 0ULL; S.feat_status[j] &= (uint8_t)~2;
            for (int i = 0; i < n; ++i) S.rowmask[F.row_off + i] = 0ULL;
        }
    }
    __syncthreads();
    s_sum[tid] = stack; s_cross[tid] = meff;
    for (int off = 32; off > 0; off >>= 1) orm |= __shfl_xor(orm, off);
    if ((tid & 63) == 0) s_or[tid >> 6] = orm;
    __syncthreads();
    if (tid == 0) {
        int st = 0, me = 0;
        for (int t = 0; t < NT; ++t) { st += s_sum[t]; if (s_cross[t] > me) me = s_cross[t]; }
        unsigned long long m = 0ULL;
        for (int w = 0; w < NT / 64; ++w) m |= s_or[w];
        s_tot[0] = st; s_tot[1] = me; s_mask = m;
    }
    __syncthreads();
    EkfCapResult R;
    R.stacked = s_tot[0]; R.end = s_tot[1]; R.clones = s_mask;
    R.na = 6 * (int)__popcll(R.clones);
    if (publish) {
        for (int i = tid; i < R.na; i += NT) S.act[i] = ekf_act_column(R.clones, i);
        if (tid == 0) {
            S.rows_out[0] = R.stacked;
            S.rows_out[1] = R.end;       // rows beyond the last stacked block carry nothing: the Gram pass stops there
            S.rows_out[2] = R.na;
            S.rows_out[3] = 0;
            S.rows_out[4] = R.na;        // rows of the compressed measurement (the factorisation kernel lowers it when nothing is compressed)
        }
    }
    return R;
}
