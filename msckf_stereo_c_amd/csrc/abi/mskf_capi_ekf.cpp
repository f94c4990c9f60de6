// mskf_capi_ekf.cpp — C-ABI: EKF entry points (include/mskf_hip.h).  The covariance lives in HBM;
// the host passes small per-frame descriptors (Phi/Q sequence, J, clone states, observation lists).
#include <algorithm>
#include <chrono>
#include <cstring>
#include "../../../include/mskf_chi2_table.h"
#include "mskf_internal.h"

extern "C" {
void ekf_launch_propagate(const EkfStreamDev *d, int n, hipStream_t st);
void ekf_launch_augment(const EkfStreamDev *d, int n, hipStream_t st);
void ekf_launch_remove_clone(const EkfStreamDev *d, int n, hipStream_t st);
void ekf_launch_features(const EkfStreamDev *d, const int *work_wave, int n_wave, const int *work_small, int n_small, const int *work_big, int n_big,
                         int max_rows, int max_rows_small, int big_clones, hipStream_t st);
void ekf_launch_pair_features(const EkfStreamDev *d, int n, int max_feat, int max_tri, hipStream_t st);
void ekf_launch_posvar(const EkfStreamDev *d, int n, double *out, hipStream_t st);
void ekf_launch_posvar_upd(const EkfStreamDev *d, int n, hipStream_t st);
void ekf_launch_gemm(const EkfStreamDev *d, int n, int mode, int max_mn, hipStream_t st);
void ekf_launch_chol(const EkfStreamDev *d, int n, int which, int max_d, hipStream_t st);
void ekf_launch_tsqr(const EkfStreamDev *d, int n, int max_d, int do_cap, hipStream_t st);
void ekf_launch_trsm(const EkfStreamDev *d, int n, int max_d, hipStream_t st);
void ekf_launch_small_update(const EkfStreamDev *d, int n, int max_d, hipStream_t st);
int ekf_small_update_max_na(void);
}

namespace {
constexpr int kMaxClonesDev = 64;   // MAX_CLONES_DEV in ekf_kernels.hip
constexpr int kMaxRows = 65536;     // stacked-Jacobian row capacity per stream and update

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Zero-filled device buffer.  The fill is enqueued on the stream the buffer is going to be used on: the context
// streams are hipStreamNonBlocking, so a hipMemset on the null stream is not ordered against their kernels (it is
// asynchronous for device memory) and could land on top of data a kernel had already written.
int dev_alloc(double **p, size_t n, hipStream_t st) {
    MSKF_HIPCHK(hipMalloc((void **)p, n * sizeof(double)));
    MSKF_HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(double), st));
    return MSKF_OK;
}

// a pinned host arena with a device twin and an event guarding reuse of the host side
int arena_ensure(char **h, char **d, size_t *cap, size_t need) {
    if (need <= *cap) return MSKF_OK;
    // (single-stream entry points only — mskf_ekf_propagate / _augment —, not on the batched path: see PinnedDev::ensure for why
    // a running pipeline must not free)
    if (*h) (void)hipHostFree(*h);
    if (*d) (void)hipFree(*d);
    *h = *d = nullptr; *cap = 0;
    const size_t c = align_up(need + need / 2, 4096);
    MSKF_HIPCHK(hipHostMalloc((void **)h, c, hipHostMallocDefault));
    MSKF_HIPCHK(hipMalloc((void **)d, c));
    *cap = c;
    return MSKF_OK;
}
}  // namespace

struct EkfExtra {  // lives behind EkfStreamState via the stream (kept out of the device header)
    EkfStreamDev desc_static;         // constant part of the stream's descriptors (base_desc)
    bool desc_valid = false;
    double *P_alt = nullptr;          // ping-pong target of clone removal
    char *h_small = nullptr, *d_small = nullptr;   // Phi/Q sequence, J
    size_t small_cap = 0;
    hipEvent_t small_done = nullptr;
    bool small_pending = false;
    std::vector<double *> retired_hs; // stacked-Jacobian blocks from hipMalloc that were outgrown: freed with the stream (hipFree inside a run waits for the whole device)
};
static EkfExtra *extra_of(mskf_stream *s) { return (EkfExtra *)s->ekf_extra; }

int mskf_ekf_stream_init(mskf_stream *s) {
    EkfStreamState &E = s->ekf_state;
    E.max_clones = s->ekf.max_cam_state_size;
    if (E.max_clones < 4 || E.max_clones > kMaxClonesDev) {
        mskf_set_error("max_cam_state_size must be in [4, 64]");
        return MSKF_ERR_UNSUPPORTED;
    }
    if (s->ekf.compression_mode < 0 || s->ekf.compression_mode > 3) {
        // (the field took the place of padding in ABI v1: a caller that never initialised it must hear about it)
        mskf_set_error("mskf_ekf_cfg.compression_mode must be 0 (auto), 1 (Gram only), 2 (Householder TSQR always) or 3 (the reference's rule)");
        return MSKF_ERR_INVALID;
    }
    E.ld = (int)align_up((size_t)(EKF_IMU_DIM + 6 * E.max_clones), 8);
    E.d = EKF_IMU_DIM;
    E.nmax = 4 * E.max_clones;
    int rc;
    const size_t pl = (size_t)E.ld * E.ld;
    hipStream_t st = s->ctx->stream;     // the fills are drained below: the stream may be re-attached to another context later
    // one allocation and one fill for the stream's fixed-size filter buffers (eight of each per stream showed up as
    // thousands of fill kernels in the profile of a 1536-stream run)
    EkfExtra *X = new EkfExtra();
    s->ekf_extra = X;
    {
        auto pad = [](size_t n) { return (n + 31) / 32 * 32; };      // 256-byte aligned sub-buffers
        const size_t n_gate = pad((size_t)EKF_SLOTS * E.nmax * E.nmax), n_act = pad((size_t)E.ld), n_chi = pad(128), n_pl = pad(pl);
        if ((rc = dev_alloc(&E.pool, 5 * n_pl + n_act + n_gate + n_chi, st)) != MSKF_OK) return rc;
        double *q = E.pool;
        E.P = q; q += n_pl; E.T = q; q += n_pl; E.S = q; q += n_pl; E.W = q; q += n_pl; X->P_alt = q; q += n_pl;
        E.act = (int *)q; q += n_act;           // ld ints fit
        E.gate_S = q; q += n_gate;
        E.chi2 = q;
    }
    {
        // stacked Jacobian + row masks: the APPLIED lost-feature stack is capped at max_stack_rows + one block and the pruning
        // stack is five rows per feature the map can hold (every live grid slot), which is what this first block is sized for.
        // The rows are laid out BEFORE the cap, though (every lost feature's block has its place): a frame that loses most of
        // its features at once (a blackout) needs more and grows the buffer, stream-ordered (mskf_ekf_update_batch_begin)
        const int live = s->fe.grid_row * s->fe.grid_col * std::max(s->fe.grid_max_feature_num, 1) + 64;
        const int cap = std::min(kMaxRows, std::max(std::max(2048, s->ekf.max_stack_rows + 4 * E.max_clones + 64), 5 * live));
        if ((rc = dev_alloc(&E.Hs, (size_t)cap * E.ld + (size_t)cap, st)) != MSKF_OK) return rc;
        E.rs = E.Hs + (size_t)cap * E.ld;
        E.max_rows = cap;
    }
    MSKF_HIPCHK(hipStreamSynchronize(st));
    double tab[100];
    tab[0] = 0.0;
    for (int i = 1; i < 100; ++i) tab[i] = s->ekf.chi2_mode == 1 ? mskf_chi2_ppf95[i - 1] : mskf_chi2_ppf05[i - 1];  // msckf_vio.cpp:181-185
    MSKF_HIPCHK(hipMemcpy(E.chi2, tab, sizeof(tab), hipMemcpyHostToDevice));
    MSKF_HIPCHK(hipEventCreateWithFlags(&X->small_done, hipEventDisableTiming));
    return MSKF_OK;
}

void mskf_ekf_stream_free(mskf_stream *s) {
    EkfStreamState &E = s->ekf_state;
    if (E.pool) (void)hipFree(E.pool);          // P, T, S, W, P_alt, act, gate_S, chi2
    if (E.Hs) { if (E.hs_async) (void)hipFreeAsync(E.Hs, s->ctx_ekf->stream); else (void)hipFree(E.Hs); }      // Hs + rowmask
    if (E.h_arena) (void)hipHostFree(E.h_arena);
    if (E.d_arena) (void)hipFree(E.d_arena);
    if (E.h_out) (void)hipHostFree(E.h_out);
    if (E.d_out) (void)hipFree(E.d_out);
    if (EkfExtra *X = extra_of(s)) {
        for (double *p : X->retired_hs) (void)hipFree(p);
        if (X->h_small) (void)hipHostFree(X->h_small);
        if (X->d_small) (void)hipFree(X->d_small);
        if (X->small_done) (void)hipEventDestroy(X->small_done);
        delete X;
        s->ekf_extra = nullptr;
    }
}

// The per-stream part of a descriptor that never changes (noise levels, extrinsics, buffer pointers that are only
// replaced together with `desc_valid = false`) is built once and copied; d and the ping-pong P are patched in.
static void base_desc(const mskf_stream *s, EkfStreamDev &D) {
    const EkfStreamState &E = s->ekf_state;
    EkfExtra *X = extra_of(const_cast<mskf_stream *>(s));
    if (!X->desc_valid) {
        EkfStreamDev &T = X->desc_static;
        std::memset(&T, 0, sizeof(T));
        T.ld = E.ld;
        T.sigma2 = s->ekf.noise_feature * s->ekf.noise_feature;   // msckf_vio.cpp:74,81
        T.max_stack_rows = s->ekf.max_stack_rows;
        T.qr_mode = s->ekf.compression_mode;
        T.chi2 = E.chi2;
        T.remove_index = T.remove_index2 = -1;
        // continuous_noise_cov diagonal blocks: gyro, gyro bias, acc, acc bias (msckf_vio.cpp:70-80, 174-178)
        T.qc[0] = s->ekf.noise_gyro * s->ekf.noise_gyro;
        T.qc[1] = s->ekf.noise_gyro_bias * s->ekf.noise_gyro_bias;
        T.qc[2] = s->ekf.noise_acc * s->ekf.noise_acc;
        T.qc[3] = s->ekf.noise_acc_bias * s->ekf.noise_acc_bias;
        T.T = E.T; T.S = E.S; T.W = E.W; T.act = E.act; T.gate_S = E.gate_S; T.nmax = E.nmax;
        hm::Rigid T01 = hm::Rigid::from_rowmajor16(s->calib.T_cam1_cam0);   // CAMState::T_cam0_cam1, msckf_vio.cpp:121-122
        std::memcpy(T.R_c0_c1, T01.R.m, sizeof(T.R_c0_c1));
        for (int i = 0; i < 3; ++i) T.t_c0_c1[i] = T01.t[i];
        X->desc_valid = true;
    }
    D = X->desc_static;
    D.P = E.P; D.d = E.d;
    D.Hs = E.Hs; D.rowmask = (unsigned long long *)E.rs;
}

static int small_begin(mskf_stream *s, EkfExtra *X, size_t bytes) {
    if (X->small_pending) { MSKF_HIPCHK(hipEventSynchronize(X->small_done)); X->small_pending = false; }
    if (bytes > X->small_cap) {
        MSKF_HIPCHK(hipStreamSynchronize(s->ctx_ekf->stream));
        int rc = arena_ensure(&X->h_small, &X->d_small, &X->small_cap, bytes);
        if (rc != MSKF_OK) return rc;
    }
    return MSKF_OK;
}

extern "C" int mskf_ekf_reset(mskf_stream *s, const double *P0) {
    if (!s || !P0) return MSKF_ERR_INVALID;
    EkfStreamState &E = s->ekf_state;
    MSKF_HIPCHK(hipSetDevice(s->ctx_ekf->device));
    hipStream_t st = s->ctx_ekf->stream;       // everything on the stream's own queue: the null stream is not ordered against it
    MSKF_HIPCHK(hipMemsetAsync(E.P, 0, sizeof(double) * (size_t)E.ld * E.ld, st));
    MSKF_HIPCHK(hipMemcpy2DAsync(E.P, sizeof(double) * E.ld, P0, sizeof(double) * EKF_IMU_DIM, sizeof(double) * EKF_IMU_DIM, EKF_IMU_DIM,
                                 hipMemcpyHostToDevice, st));
    MSKF_HIPCHK(hipStreamSynchronize(st));
    E.d = EKF_IMU_DIM;
    return MSKF_OK;
}

extern "C" int mskf_ekf_set_cov(mskf_stream *s, const double *P, int d) {
    if (!s || !P) return MSKF_ERR_INVALID;
    EkfStreamState &E = s->ekf_state;
    if (d < EKF_IMU_DIM || (d - EKF_IMU_DIM) % 6 || d > EKF_IMU_DIM + 6 * E.max_clones) return MSKF_ERR_CAPACITY;
    MSKF_HIPCHK(hipSetDevice(s->ctx_ekf->device));
    hipStream_t st = s->ctx_ekf->stream;
    MSKF_HIPCHK(hipMemsetAsync(E.P, 0, sizeof(double) * (size_t)E.ld * E.ld, st));
    MSKF_HIPCHK(hipMemcpy2DAsync(E.P, sizeof(double) * E.ld, P, sizeof(double) * d, sizeof(double) * d, d, hipMemcpyHostToDevice, st));
    MSKF_HIPCHK(hipStreamSynchronize(st));
    E.d = d;
    return MSKF_OK;
}

extern "C" int mskf_ekf_set_compression_mode(mskf_stream *s, int mode) {
    if (!s) return MSKF_ERR_INVALID;
    if (mode < 0 || mode > 3) { mskf_set_error("compression_mode must be 0 (auto), 1 (Gram only), 2 (Householder TSQR always) or 3 (the reference's rule)"); return MSKF_ERR_INVALID; }
    if (s->ctx_ekf->pend_upd.active) { mskf_set_error("an update batch of the stream's context is pending"); return MSKF_ERR_INVALID; }
    s->ekf.compression_mode = mode;
    extra_of(s)->desc_valid = false;          // the cached descriptor carries the mode
    return MSKF_OK;
}

extern "C" int mskf_ekf_get_dim(mskf_stream *s, int *d) {
    if (!s || !d) return MSKF_ERR_INVALID;
    *d = s->ekf_state.d;
    return MSKF_OK;
}

extern "C" int mskf_ekf_get_cov(mskf_stream *s, double *P, int capacity) {
    if (!s || !P) return MSKF_ERR_INVALID;
    EkfStreamState &E = s->ekf_state;
    if (capacity < E.d * E.d) return MSKF_ERR_CAPACITY;
    MSKF_HIPCHK(hipSetDevice(s->ctx_ekf->device));
    MSKF_HIPCHK(hipStreamSynchronize(s->ctx_ekf->stream));
    MSKF_HIPCHK(hipMemcpy2D(P, sizeof(double) * E.d, E.P, sizeof(double) * E.ld, sizeof(double) * E.d, E.d, hipMemcpyDeviceToHost));
    return MSKF_OK;
}

extern "C" int mskf_ekf_propagate(mskf_stream *s, int n_steps, const double *Phi, const double *Q) {
    if (!s || n_steps < 0 || (n_steps && (!Phi || !Q))) return MSKF_ERR_INVALID;
    if (n_steps == 0) return MSKF_OK;
    EkfExtra *X = extra_of(s);
    mskf_ctx *ctx = s->ctx_ekf;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    const size_t nn = EKF_IMU_DIM * EKF_IMU_DIM;
    const size_t bytes = sizeof(double) * 2 * nn * (size_t)n_steps + sizeof(EkfStreamDev);
    int rc = small_begin(s, X, bytes);
    if (rc != MSKF_OK) return rc;
    EkfStreamDev *D = (EkfStreamDev *)X->h_small;
    double *pq = (double *)(X->h_small + sizeof(EkfStreamDev));
    for (int k = 0; k < n_steps; ++k) {
        std::memcpy(pq + (size_t)k * 2 * nn, Phi + (size_t)k * nn, sizeof(double) * nn);
        std::memcpy(pq + (size_t)k * 2 * nn + nn, Q + (size_t)k * nn, sizeof(double) * nn);
    }
    base_desc(s, *D);
    D->PhiQ = (const double *)(X->d_small + sizeof(EkfStreamDev));
    D->n_steps = n_steps;
    MSKF_HIPCHK(hipMemcpyAsync(X->d_small, X->h_small, bytes, hipMemcpyHostToDevice, ctx->stream));
    MSKF_HIPCHK(hipEventRecord(X->small_done, ctx->stream));
    X->small_pending = true;
    ekf_launch_propagate((const EkfStreamDev *)X->d_small, 1, ctx->stream);
    MSKF_HIPCHK(hipGetLastError());
    return MSKF_OK;
}

extern "C" int mskf_ekf_propagate_imu(mskf_stream *s, int n_steps, const mskf_imu_step *steps) {
    if (!s || n_steps < 0 || (n_steps && !steps)) return MSKF_ERR_INVALID;
    if (n_steps == 0) return MSKF_OK;
    EkfExtra *X = extra_of(s);
    mskf_ctx *ctx = s->ctx_ekf;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(EkfStreamDev) + sizeof(mskf_imu_step) * (size_t)n_steps;
    int rc = small_begin(s, X, bytes);
    if (rc != MSKF_OK) return rc;
    EkfStreamDev *D = (EkfStreamDev *)X->h_small;
    std::memcpy(X->h_small + sizeof(EkfStreamDev), steps, sizeof(mskf_imu_step) * (size_t)n_steps);
    base_desc(s, *D);
    D->imu_steps = (const mskf_imu_step *)(X->d_small + sizeof(EkfStreamDev));
    D->n_steps = n_steps;
    MSKF_HIPCHK(hipMemcpyAsync(X->d_small, X->h_small, bytes, hipMemcpyHostToDevice, ctx->stream));
    MSKF_HIPCHK(hipEventRecord(X->small_done, ctx->stream));
    X->small_pending = true;
    {
        const int ts = mskf_t_begin(ctx, MSKF_K_EKF_PROPAGATE);
        ekf_launch_propagate((const EkfStreamDev *)X->d_small, 1, ctx->stream);
        mskf_t_end(ctx, ts, 1);
    }
    MSKF_HIPCHK(hipGetLastError());
    return MSKF_OK;
}

extern "C" int mskf_ekf_predict_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const int32_t *n_steps,
                                      const mskf_imu_step *const *steps, const double *const *J) {
    if (!ctx || n <= 0 || !streams || !n_steps || !steps || !J) return MSKF_ERR_INVALID;
    // (the position-variance read-out works out of the same pinned arena, its kernel reads its descriptors there: not before its _end)
    if (ctx->pend_pv.active) { mskf_set_error("a position-variance read-out of this context is still pending"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    size_t bytes = align_up(sizeof(EkfStreamDev) * (size_t)n, 64);
    for (int i = 0; i < n; ++i) {
        if (!streams[i] || streams[i]->ctx_ekf != ctx || n_steps[i] < 0 || (n_steps[i] && !steps[i])) return MSKF_ERR_INVALID;
        bytes += align_up(sizeof(mskf_imu_step) * (size_t)n_steps[i], 64) + align_up(sizeof(double) * 6 * EKF_IMU_DIM, 64);
    }
    if (!ctx->pred_done) MSKF_HIPCHK(hipEventCreateWithFlags(&ctx->pred_done, hipEventDisableTiming));
    if (ctx->pred_pending) { MSKF_HIPCHK(hipEventSynchronize(ctx->pred_done)); ctx->pred_pending = false; }
    if (bytes > ctx->pred_arena.cap) {
        MSKF_HIPCHK(hipStreamSynchronize(st));
        int rc = ctx->pred_arena.ensure(bytes);
        if (rc != MSKF_OK) return rc;
    }
    char *h = ctx->pred_arena.h, *dv = ctx->pred_arena.d;
    EkfStreamDev *D = (EkfStreamDev *)h;
    size_t off = align_up(sizeof(EkfStreamDev) * (size_t)n, 64);
    bool any = false;
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        EkfStreamState &E = s->ekf_state;
        base_desc(s, D[i]);
        D[i].n_steps = n_steps[i];
        if (n_steps[i] > 0) {
            D[i].PhiQ = nullptr;                           // Phi_k, Q_k are built inside k_ekf_propagate from the IMU steps
            std::memcpy(h + off, steps[i], sizeof(mskf_imu_step) * (size_t)n_steps[i]);
            D[i].imu_steps = (const mskf_imu_step *)(dv + off);
            off += align_up(sizeof(mskf_imu_step) * (size_t)n_steps[i], 64);
            any = true;
        }
        if (J[i]) {
            if (E.d + 6 > EKF_IMU_DIM + 6 * E.max_clones) { mskf_set_error("clone capacity exceeded"); return MSKF_ERR_CAPACITY; }
            std::memcpy(h + off, J[i], sizeof(double) * 6 * EKF_IMU_DIM);
            D[i].J = (const double *)(dv + off);
            off += align_up(sizeof(double) * 6 * EKF_IMU_DIM, 64);
            any = true;
        }
    }
    if (!any) return MSKF_OK;
    { const MskfCopy cp = {dv, h, off}; const int crc = mskf_copy_async(ctx, &cp, 1); if (crc != MSKF_OK) return crc; }
    MSKF_HIPCHK(hipEventRecord(ctx->pred_done, st));
    ctx->pred_pending = true;
    {
        const int ts = mskf_t_begin(ctx, MSKF_K_EKF_PROPAGATE);
        ekf_launch_propagate((const EkfStreamDev *)dv, n, st);
        mskf_t_end(ctx, ts, n);
    }
    MSKF_HIPCHK(hipGetLastError());
    for (int i = 0; i < n; ++i) if (J[i]) streams[i]->ekf_state.d += 6;
    return MSKF_OK;
}

extern "C" int mskf_ekf_get_pos_var_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, double *out) {
    const int rc = mskf_ekf_get_pos_var_batch_begin(ctx, n, streams, out);
    return rc != MSKF_OK ? rc : mskf_ekf_get_pos_var_batch_end(ctx);
}

extern "C" int mskf_ekf_get_pos_var_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, double *out) {
    if (!ctx || n <= 0 || !streams || !out) return MSKF_ERR_INVALID;
    if (ctx->pend_pv.active) return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t desc_bytes = align_up(sizeof(EkfStreamDev) * (size_t)n, 64);
    const size_t bytes = desc_bytes + sizeof(double) * 3 * (size_t)n;
    if (!ctx->pred_done) MSKF_HIPCHK(hipEventCreateWithFlags(&ctx->pred_done, hipEventDisableTiming));
    if (ctx->pred_pending) { MSKF_HIPCHK(hipEventSynchronize(ctx->pred_done)); ctx->pred_pending = false; }
    if (bytes > ctx->pred_arena.cap) {
        MSKF_HIPCHK(hipStreamSynchronize(st));
        int rc = ctx->pred_arena.ensure(bytes);
        if (rc != MSKF_OK) return rc;
    }
    EkfStreamDev *D = (EkfStreamDev *)ctx->pred_arena.h;
    for (int i = 0; i < n; ++i) {
        if (!streams[i] || streams[i]->ctx_ekf != ctx) return MSKF_ERR_INVALID;
        base_desc(streams[i], D[i]);
    }
    // (no staging copies: the kernel reads its descriptors from the pinned arena and writes the 3 n doubles straight into it)
    ekf_launch_posvar((const EkfStreamDev *)ctx->pred_arena.h, n, (double *)(ctx->pred_arena.h + desc_bytes), st);
    { const int wrc = mskf_wait_event(ctx, &ctx->pend_pv.done, true); if (wrc != MSKF_OK) return wrc; }
    ctx->pend_pv.active = true; ctx->pend_pv.n = n; ctx->pend_pv.out = out; ctx->pend_pv.desc_bytes = desc_bytes;
    return MSKF_OK;
}

extern "C" int mskf_ekf_get_pos_var_batch_end(mskf_ctx *ctx) {
    if (!ctx) return MSKF_ERR_INVALID;
    if (!ctx->pend_pv.active) return MSKF_OK;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    ctx->pend_pv.active = false;
    { const int wrc = mskf_wait_event(ctx, &ctx->pend_pv.done, false); if (wrc != MSKF_OK) return wrc; }
    mskf_t_collect(ctx);
    std::memcpy(ctx->pend_pv.out, ctx->pred_arena.h + ctx->pend_pv.desc_bytes, sizeof(double) * 3 * (size_t)ctx->pend_pv.n);
    return MSKF_OK;
}

extern "C" int mskf_ekf_get_pos_var(mskf_stream *s, double out[3]) {
    if (!s || !out) return MSKF_ERR_INVALID;
    mskf_stream *ss[1] = {s};
    return mskf_ekf_get_pos_var_batch(s->ctx_ekf, 1, ss, out);
}

extern "C" int mskf_ekf_augment(mskf_stream *s, const double *J) {
    if (!s || !J) return MSKF_ERR_INVALID;
    EkfStreamState &E = s->ekf_state;
    if (E.d + 6 > EKF_IMU_DIM + 6 * E.max_clones) { mskf_set_error("clone capacity exceeded"); return MSKF_ERR_CAPACITY; }
    EkfExtra *X = extra_of(s);
    mskf_ctx *ctx = s->ctx_ekf;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    // the augment descriptor lives after a possible propagate descriptor: use a second region of the small arena
    const size_t off = 0;
    const size_t bytes = sizeof(EkfStreamDev) + sizeof(double) * 6 * EKF_IMU_DIM;
    int rc = small_begin(s, X, bytes);
    if (rc != MSKF_OK) return rc;
    EkfStreamDev *D = (EkfStreamDev *)(X->h_small + off);
    base_desc(s, *D);
    std::memcpy(X->h_small + off + sizeof(EkfStreamDev), J, sizeof(double) * 6 * EKF_IMU_DIM);
    D->J = (const double *)(X->d_small + off + sizeof(EkfStreamDev));
    MSKF_HIPCHK(hipMemcpyAsync(X->d_small + off, X->h_small + off, bytes, hipMemcpyHostToDevice, ctx->stream));
    MSKF_HIPCHK(hipEventRecord(X->small_done, ctx->stream));
    X->small_pending = true;
    {
        const int ts = mskf_t_begin(ctx, MSKF_K_EKF_AUGMENT);
        ekf_launch_augment((const EkfStreamDev *)(X->d_small + off), 1, ctx->stream);
        mskf_t_end(ctx, ts, 1);
    }
    MSKF_HIPCHK(hipGetLastError());
    E.d += 6;
    return MSKF_OK;
}

extern "C" int mskf_ekf_remove_clones_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const int32_t *idx) {
    if (!ctx || n <= 0 || !streams || !idx) return MSKF_ERR_INVALID;
    if (ctx->pend_pv.active) { mskf_set_error("a position-variance read-out of this context is still pending"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t bytes = sizeof(EkfStreamDev) * (size_t)n;
    if (!ctx->pred_done) MSKF_HIPCHK(hipEventCreateWithFlags(&ctx->pred_done, hipEventDisableTiming));
    if (ctx->pred_pending) { MSKF_HIPCHK(hipEventSynchronize(ctx->pred_done)); ctx->pred_pending = false; }
    if (bytes > ctx->pred_arena.cap) {
        MSKF_HIPCHK(hipStreamSynchronize(st));
        int rc = ctx->pred_arena.ensure(bytes);
        if (rc != MSKF_OK) return rc;
    }
    EkfStreamDev *D = (EkfStreamDev *)ctx->pred_arena.h;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        if (!s || s->ctx_ekf != ctx) return MSKF_ERR_INVALID;
        EkfStreamState &E = s->ekf_state;
        EkfExtra *X = extra_of(s);
        const int nc = (E.d - EKF_IMU_DIM) / 6;
        int a = idx[2 * i], b = idx[2 * i + 1];
        if (a < 0 && b >= 0) std::swap(a, b);
        if (b >= 0 && b < a) std::swap(a, b);
        if (a >= nc || b >= nc || (a >= 0 && a == b)) return MSKF_ERR_INVALID;
        base_desc(s, D[i]);
        D[i].remove_index = a; D[i].remove_index2 = b; D[i].P_dst = X->P_alt;
        any |= a >= 0;
    }
    if (!any) return MSKF_OK;
    { const MskfCopy cp = {ctx->pred_arena.d, ctx->pred_arena.h, bytes}; const int crc = mskf_copy_async(ctx, &cp, 1); if (crc != MSKF_OK) return crc; }
    MSKF_HIPCHK(hipEventRecord(ctx->pred_done, st));
    ctx->pred_pending = true;
    {
        const int ts = mskf_t_begin(ctx, MSKF_K_EKF_REMOVE);
        ekf_launch_remove_clone((const EkfStreamDev *)ctx->pred_arena.d, n, st);
        mskf_t_end(ctx, ts, n);
    }
    MSKF_HIPCHK(hipGetLastError());
    for (int i = 0; i < n; ++i) {
        const EkfStreamDev &Di = D[i];
        if (Di.remove_index < 0) continue;
        EkfStreamState &E = streams[i]->ekf_state;
        std::swap(E.P, extra_of(streams[i])->P_alt);
        E.d -= Di.remove_index2 >= 0 ? 12 : 6;
    }
    return MSKF_OK;
}

extern "C" int mskf_ekf_remove_clone(mskf_stream *s, int clone_index) {
    if (!s) return MSKF_ERR_INVALID;
    const int nc = (s->ekf_state.d - EKF_IMU_DIM) / 6;
    if (clone_index < 0 || clone_index >= nc) return MSKF_ERR_INVALID;
    mskf_stream *ss[1] = {s};
    const int32_t idx[2] = {clone_index, -1};
    return mskf_ekf_remove_clones_batch(s->ctx_ekf, 1, ss, idx);
}

extern "C" int mskf_ekf_update_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, mskf_ekf_update_args *args) {
    const int rc = mskf_ekf_update_batch_begin(ctx, n, streams, args);
    return rc != MSKF_OK ? rc : mskf_ekf_update_batch_end(ctx);
}

extern "C" int mskf_ekf_update_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, mskf_ekf_update_args *args) {
    if (!ctx || n <= 0 || !streams || !args) return MSKF_ERR_INVALID;
    if (ctx->pend_upd.active) { mskf_set_error("an update batch of this context is still pending (call mskf_ekf_update_batch_end)"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    const auto t_h0 = std::chrono::steady_clock::now();
    hipStream_t st = ctx->stream;
    int rc = ctx->ekf_desc.ensure(n);
    if (rc != MSKF_OK) return rc;
    int max_feat = 0, max_m = 0, max_d = 0, max_frows = 0, max_tri = 0, max_clones_cfg = 0;
    // Which kernels handle a stream is decided per STREAM, from that stream's features alone (EkfStreamDev::route): the same
    // stream takes the same route, and therefore runs the same arithmetic, whatever else is in the batch.
    //   pairs : every feature has exactly two Jacobian observations, all of the same ordered clone pair (the pruning update)
    //   wave  : every feature has <= 4 Jacobian observations and a triangulation over <= 32 clones: class [0] of the
    //           feature kernel, one wavefront per feature; otherwise a feature is in class [1] (<= 16 observations) or [2]
    //   small : the stream touches at most four clones (<= 24 active columns): whole update in k_ekf_small_update
    int max_frows_cls[3] = {0, 0, 0};
    std::vector<int> &cnt_cls = ctx->pend_upd.cnt_cls;          // [3][n]
    cnt_cls.assign((size_t)3 * n, 0);
    std::vector<int> route(n, 0), na_max(n, 0);
    bool any_pairs = false, any_small = false, any_general = false;
    bool any_householder = false, any_gram = false;      // among the general-route streams: compressed by k_ekf_tsqr / by the Gram factorisation
    int max_feat_pairs = 0;
    double fl_feat = 0, fl_qr = 0, fl_upd = 0;   // algorithmic FP64 flops of this launch
    struct Lay { size_t clones, feats, obs_clone, obs_z, tri; int n_tri; size_t o_dx, o_gamma, o_rows, o_status, o_pos; int m_total; };
    bool any_pv_nofeat = false;
    std::vector<Lay> lay(n);
    size_t in_bytes = 0, out_bytes = 0;
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        mskf_ekf_update_args &a = args[i];
        if (!s || s->ctx_ekf != ctx) return MSKF_ERR_INVALID;
        EkfStreamState &E = s->ekf_state;
        if (a.n_clones * 6 + EKF_IMU_DIM != E.d) { mskf_set_error("n_clones does not match the covariance dimension"); return MSKF_ERR_INVALID; }
        if (a.n_feat < 0 || a.n_obs < 0) return MSKF_ERR_INVALID;
        if (a.n_feat && (!a.clones || !a.features || !a.obs_clone || !a.obs_z || !a.delta_x || !a.feat_status || !a.rows_out)) return MSKF_ERR_INVALID;
        Lay &L = lay[i];
        int m_total = 0;
        unsigned long long clone_mask = 0ULL;          // clones any Jacobian block of this stream touches: bounds the active columns
        int n_tri = 0;
        bool pairs = a.n_feat > 0, wave = a.n_feat > 0;
        int pair_a = -1, pair_b = -1, s_frows_cls[3] = {0, 0, 0};
        for (int j = 0; j < a.n_feat; ++j) {
            const mskf_ekf_feature &f = a.features[j];
            n_tri += f.needs_init ? 1 : 0;
            if (f.n_obs < 2 || f.n_obs > E.max_clones || f.obs_start < 0 || f.obs_start + f.n_obs > a.n_obs) return MSKF_ERR_INVALID;
            if (f.needs_init && (f.n_init < 1 || f.n_init > E.max_clones || f.init_start < 0 || f.init_start + f.n_init > a.n_obs)) return MSKF_ERR_INVALID;
            if (f.n_obs != 2) pairs = false;
            else if (pairs) {
                // the pair kernel keeps the two clones' blocks in the order (lower, higher) and shares their covariance
                // block among the features: both observations distinct, ascending, and the same pair for every feature
                const int c0 = a.obs_clone[f.obs_start], c1 = a.obs_clone[f.obs_start + 1];
                if (j == 0) { pair_a = c0; pair_b = c1; }
                if (!(c0 < c1) || c0 != pair_a || c1 != pair_b) pairs = false;
            }
            if (4 * f.n_obs > 16 || (f.needs_init && f.n_init > 32)) wave = false;      // (TRI_SMALL_CLONES)
            m_total += 4 * f.n_obs - 3;
            max_frows = std::max(max_frows, 4 * f.n_obs);
            {
                const int cls_n = std::max(f.n_obs, f.needs_init ? f.n_init : 0), cls = cls_n <= 16 ? 1 : 2;
                s_frows_cls[cls] = std::max(s_frows_cls[cls], 4 * f.n_obs);
                ++cnt_cls[(size_t)cls * n + i];
            }
            for (int o = 0; o < f.n_obs; ++o) {
                const int c = a.obs_clone[f.obs_start + o];
                if (c < 0 || c >= a.n_clones) return MSKF_ERR_INVALID;
                clone_mask |= 1ULL << c;
            }
            const double nj = 4.0 * f.n_obs - 3.0, M = f.n_obs, dd = E.d;
            fl_feat += 2.0 * nj * (4.0 * M) * (6.0 * M) + 2.0 * nj * dd * dd + 2.0 * nj * nj * dd;
        }
        if (a.n_feat > 0) {
            const double dd = E.d, mm = m_total;
            if (m_total > E.d) fl_qr += 2.0 * mm * dd * dd - (2.0 / 3.0) * dd * dd * dd;
            fl_upd += (4.0 + 1.0 / 3.0 + 2.0 + 2.0 + 2.0) * dd * dd * dd;
        }
        if (m_total > kMaxRows) { mskf_set_error("stacked Jacobian exceeds the row capacity"); return MSKF_ERR_CAPACITY; }
        L.m_total = m_total;
        L.clones = in_bytes;
        L.feats = align_up(L.clones + sizeof(mskf_clone_state) * (size_t)a.n_clones, 16);
        L.obs_clone = align_up(L.feats + sizeof(EkfFeatDev) * (size_t)a.n_feat, 16);
        L.obs_z = align_up(L.obs_clone + sizeof(int) * (size_t)a.n_obs, 16);
        L.tri = align_up(L.obs_z + sizeof(double) * 4 * (size_t)a.n_obs, 16);
        in_bytes = align_up(L.tri + sizeof(int) * (size_t)n_tri, 64);
        L.o_dx = out_bytes;
        L.o_gamma = align_up(L.o_dx + sizeof(double) * (size_t)E.ld, 16);
        L.o_pos = align_up(L.o_gamma + sizeof(double) * (size_t)a.n_feat, 16);
        L.o_rows = align_up(L.o_pos + sizeof(double) * 3 * (size_t)a.n_feat, 16);
        L.o_status = L.o_rows + 32 + 32;           // rows_out (5 ints, padded to 32 bytes) + 3 position variances
        any_pv_nofeat |= a.pos_var_out != nullptr && a.n_feat == 0;
        out_bytes = align_up(L.o_status + (size_t)a.n_feat, 64);
        if (m_total > E.max_rows) {
            // Growth of the stacked-Jacobian buffer, STREAM-ORDERED: hipFree / hipMalloc synchronise the whole device, and a
            // group that grows a buffer in the middle of a run then waits until every other group's queue is idle (measured in
            // the round-3 bench: the two groups that met their largest pruning update inside the timed window stood still for
            // 0.3 s each).  The first allocation is sized for what a stream of this configuration can stack (stream init).
            // The new block is allocated FIRST: if that fails the call fails with the stream's buffers as they were (a later,
            // smaller update still finds a valid Hs of max_rows rows).  The old block goes back to the pool it came from: a
            // stream-ordered free for a stream-ordered block, retirement until the stream is destroyed for the first one
            // (hipMalloc'ed at stream creation; hipFreeAsync does not take such a pointer without a device-wide wait).
            const int cap = std::min(kMaxRows, std::max(2048, m_total + m_total / 2));
            const size_t bytes = ((size_t)cap * E.ld + (size_t)cap) * sizeof(double);
            double *grown = nullptr;
            MSKF_HIPCHK(hipMallocAsync((void **)&grown, bytes, st));
            if (hipMemsetAsync(grown, 0, bytes, st) != hipSuccess) { (void)hipFreeAsync(grown, st); mskf_set_error("hipMemsetAsync of the grown stacked-Jacobian buffer failed"); return MSKF_ERR_HIP; }
            if (E.Hs) { if (E.hs_async) (void)hipFreeAsync(E.Hs, st); else extra_of(s)->retired_hs.push_back(E.Hs); }      // (rs lives behind Hs in the same allocation)
            E.Hs = grown;
            E.rs = E.Hs + (size_t)cap * E.ld;
            E.max_rows = cap;
            E.hs_async = true;
        }
        if (a.n_feat > 0) {
            if (pairs) wave = false;
            const bool small = 6 * __builtin_popcountll(clone_mask) <= ekf_small_update_max_na();
            route[i] = (pairs ? 1 : 0) | (wave ? 2 : 0) | (small ? 4 : 0);
            na_max[i] = 6 * __builtin_popcountll(clone_mask);
            any_pairs |= pairs; any_small |= small; any_general |= !small;
            if (!small) { const bool hh = s->ekf.compression_mode == 2 || s->ekf.compression_mode == 3; any_householder |= hh; any_gram |= !hh; }
            if (pairs) { max_feat_pairs = std::max(max_feat_pairs, a.n_feat); max_tri = std::max(max_tri, n_tri); }
            if (pairs || wave) {
                // the whole stream is class [0] (wave) or handled by the pair kernels: none of its features in [1] / [2]
                cnt_cls[i] = pairs ? 0 : cnt_cls[(size_t)n + i] + cnt_cls[(size_t)2 * n + i];
                cnt_cls[(size_t)n + i] = 0; cnt_cls[(size_t)2 * n + i] = 0;
            } else {
                for (int c = 1; c < 3; ++c) max_frows_cls[c] = std::max(max_frows_cls[c], s_frows_cls[c]);
            }
        }
        max_clones_cfg = std::max(max_clones_cfg, E.max_clones);
        L.n_tri = n_tri;
        max_feat = std::max(max_feat, a.n_feat);
        max_m = std::max(max_m, m_total);
        max_d = std::max(max_d, E.d);
    }
    // work lists: stream << 16 | slot << 8 | n_slots per feature group in flight
    if (n > 0xffff) { mskf_set_error("too many streams in one update batch"); return MSKF_ERR_CAPACITY; }
    int n_work[3] = {0, 0, 0};
    for (int c = 0; c < 3; ++c) for (int i = 0; i < n; ++i) n_work[c] += std::min(cnt_cls[(size_t)c * n + i], EKF_SLOTS);
    const size_t work_off = in_bytes;
    in_bytes = align_up(in_bytes + sizeof(int) * (size_t)(n_work[0] + n_work[1] + n_work[2]), 64);
    if (in_bytes > ctx->upd_in.cap || out_bytes > ctx->upd_out.cap) {
        MSKF_HIPCHK(hipStreamSynchronize(st));
        if ((rc = ctx->upd_in.ensure(in_bytes)) != MSKF_OK) return rc;
        if ((rc = ctx->upd_out.ensure(out_bytes)) != MSKF_OK) return rc;
    }
    char *hin = ctx->upd_in.h, *din = ctx->upd_in.d, *hout = ctx->upd_out.h, *dout = ctx->upd_out.d;
    {
        int *w = (int *)(hin + work_off);
        for (int c = 0; c < 3; ++c)
            for (int i = 0; i < n; ++i) {
                const int ns = std::min(cnt_cls[(size_t)c * n + i], EKF_SLOTS);
                for (int k = 0; k < ns; ++k) *w++ = (i << 16) | (k << 8) | ns;
            }
    }
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        mskf_ekf_update_args &a = args[i];
        const Lay &L = lay[i];
        if (a.n_clones) std::memcpy(hin + L.clones, a.clones, sizeof(mskf_clone_state) * (size_t)a.n_clones);
        EkfFeatDev *fd = (EkfFeatDev *)(hin + L.feats);
        int *tri = (int *)(hin + L.tri);
        int n_tri_w = 0;
        int row = 0;
        for (int j = 0; j < a.n_feat; ++j) {
            const mskf_ekf_feature &f = a.features[j];
            fd[j].obs_start = f.obs_start; fd[j].n_obs = f.n_obs;
            fd[j].needs_init = f.needs_init; fd[j].init_start = f.init_start; fd[j].n_init = f.n_init;
            fd[j].row_off = row;
            fd[j].position[0] = f.position[0]; fd[j].position[1] = f.position[1]; fd[j].position[2] = f.position[2];
            fd[j].colmask = 0ULL;
            row += 4 * f.n_obs - 3;
            if (f.needs_init) tri[n_tri_w++] = j;
        }
        if (a.n_obs) {
            std::memcpy(hin + L.obs_clone, a.obs_clone, sizeof(int) * (size_t)a.n_obs);
            std::memcpy(hin + L.obs_z, a.obs_z, sizeof(double) * 4 * (size_t)a.n_obs);
        }
        EkfStreamDev &D = ctx->ekf_desc.h[i];
        base_desc(s, D);
        D.n_clones = a.n_clones; D.n_feat = a.n_feat; D.n_obs = a.n_obs;
        D.route = route[i]; D.na_max = na_max[i];
        D.dof_offset = a.dof_offset; D.apply_row_cap = a.apply_row_cap;
        D.m_total = L.m_total;
        for (int k = 0; k < 3; ++k) D.gravity[k] = a.gravity[k];
        D.clones = (const mskf_clone_state *)(din + L.clones);
        D.feats = (EkfFeatDev *)(din + L.feats);
        D.obs_clone = (const int *)(din + L.obs_clone);
        D.obs_z = (const double *)(din + L.obs_z);
        D.tri_idx = (const int *)(din + L.tri);
        D.n_tri = L.n_tri;
        D.delta_x = (double *)(dout + L.o_dx);
        D.gamma = (double *)(dout + L.o_gamma);
        D.pos_out = (double *)(dout + L.o_pos);
        D.rows_out = (int *)(dout + L.o_rows);
        D.pos_var_out = a.pos_var_out ? (double *)(dout + L.o_rows + 32) : nullptr;
        D.feat_status = (uint8_t *)(dout + L.o_status);
    }
    if (max_feat > 0) {
        {
            const MskfCopy cp[2] = {{din, hin, in_bytes}, {ctx->ekf_desc.d, ctx->ekf_desc.h, sizeof(EkfStreamDev) * (size_t)n}};
            if ((rc = mskf_copy_async(ctx, cp, 2)) != MSKF_OK) return rc;
        }
        int ts = mskf_t_begin(ctx, MSKF_K_EKF_FEATURES);
        if (any_pairs) ekf_launch_pair_features(ctx->ekf_desc.d, n, max_feat_pairs, max_tri, st);      // the pruning update
        if (n_work[0] + n_work[1] + n_work[2] > 0) {
            const int *w0 = (const int *)(din + work_off);
            ekf_launch_features(ctx->ekf_desc.d, w0, n_work[0], w0 + n_work[0], n_work[1], w0 + n_work[0] + n_work[1], n_work[2], max_frows,
                                max_frows_cls[1], max_clones_cfg, st);
        }
        mskf_t_end(ctx, ts, (long long)fl_feat);
        // (which blocks are stacked - the 1500-row cap of :1002-1010 - is worked out by the first dense kernel of each route
        //  itself: ekf_cap.h; rounds 1-3 ran it as a launch of its own here)
        enum { GM_GRAM = 0, GM_T = 1, GM_S2 = 2, GM_PUPD = 3 };
        const double d3 = fl_upd / (4.0 + 1.0 / 3.0 + 2.0 + 2.0 + 2.0);     // sum of d^3 over the launch
        if (any_small) {
            // streams that stack blocks of at most four clones (the pruning update: the two clones being removed):
            // compression, gain and Y in one launch (k_ekf_small_update); the other streams leave it at once
            ts = mskf_t_begin(ctx, MSKF_K_EKF_SMALL);
            ekf_launch_small_update(ctx->ekf_desc.d, n, max_d, st);
            mskf_t_end(ctx, ts, any_general ? 0 : (long long)(fl_qr + fl_upd));
        }
        if (any_general) {
            // QR compression as Gram + semidefinite Cholesky (Householder TSQR inside the factorisation kernel for the
            // streams that need it), then the Kalman update (ekf_linalg.hip); small-route streams leave these at once
            if (any_gram) {
                // (also the first dense kernel of the general route: its tiles work out the stacking decision for every stream)
                ts = mskf_t_begin(ctx, MSKF_K_EKF_GEMM);
                ekf_launch_gemm(ctx->ekf_desc.d, n, GM_GRAM, max_d + 1, st);
                mskf_t_end(ctx, ts, (long long)fl_qr);
            }
            if (any_householder) {
                // Householder TSQR of the streams in compression_mode 2 / 3 (a kernel without LDS of its own: ekf_linalg.hip); with no
                // Gram launch before it, it is the first dense kernel and makes the stacking decision itself
                ts = mskf_t_begin(ctx, MSKF_K_EKF_TSQR);
                ekf_launch_tsqr(ctx->ekf_desc.d, n, max_d, any_gram ? 0 : 1, st);
                mskf_t_end(ctx, ts, (long long)fl_qr);
            }
            if (any_gram) {
                ts = mskf_t_begin(ctx, MSKF_K_EKF_CHOL);
                ekf_launch_chol(ctx->ekf_desc.d, n, 0, max_d, st);
                mskf_t_end(ctx, ts, (long long)(d3 / 3.0));
            }
            ts = mskf_t_begin(ctx, MSKF_K_EKF_GEMM);
            ekf_launch_gemm(ctx->ekf_desc.d, n, GM_T, max_d, st);
            mskf_t_end(ctx, ts, (long long)(2.0 * d3));
            ts = mskf_t_begin(ctx, MSKF_K_EKF_GEMM);
            ekf_launch_gemm(ctx->ekf_desc.d, n, GM_S2, max_d, st);
            mskf_t_end(ctx, ts, (long long)(2.0 * d3));
            ts = mskf_t_begin(ctx, MSKF_K_EKF_CHOL);
            ekf_launch_chol(ctx->ekf_desc.d, n, 1, max_d, st);
            mskf_t_end(ctx, ts, (long long)(d3 / 3.0));
            ts = mskf_t_begin(ctx, MSKF_K_EKF_TRSM);
            ekf_launch_trsm(ctx->ekf_desc.d, n, max_d, st);
            mskf_t_end(ctx, ts, (long long)(2.0 * d3));
        }
        // P <- P - Y^T Y and delta_x = Y^T w for every stream, whichever route produced Y
        ts = mskf_t_begin(ctx, MSKF_K_EKF_GEMM);
        ekf_launch_gemm(ctx->ekf_desc.d, n, GM_PUPD, max_d, st);
        mskf_t_end(ctx, ts, any_general ? (long long)(4.0 * d3) : 0);
        if (any_pv_nofeat) ekf_launch_posvar_upd(ctx->ekf_desc.d, n, st);       // (streams with features get theirs from the downdate's epilogue)
        (void)max_m;
        MSKF_HIPCHK(hipGetLastError());
        { const MskfCopy cp = {hout, dout, out_bytes}; if ((rc = mskf_copy_async(ctx, &cp, 1)) != MSKF_OK) return rc; }
        if ((rc = mskf_wait_event(ctx, &ctx->pend_upd.done, true)) != MSKF_OK) return rc;
    }
    {
        mskf_ctx::PendingUpdate &U = ctx->pend_upd;
        U.active = true; U.launched = max_feat > 0; U.n = n; U.streams = streams; U.args = args;
        U.lay.resize((size_t)5 * n);
        for (int i = 0; i < n; ++i) {
            U.lay[5 * i] = lay[i].o_dx; U.lay[5 * i + 1] = lay[i].o_gamma; U.lay[5 * i + 2] = lay[i].o_rows;
            U.lay[5 * i + 3] = lay[i].o_status; U.lay[5 * i + 4] = lay[i].o_pos;
        }
    }
    if (ctx->t_gate) ctx->host_s[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h0).count();
    return MSKF_OK;
}

// Wait for the batch started by mskf_ekf_update_batch_begin and copy delta_x, gate results, positions and the stacked
// row counts into the args given there.  No-op when nothing is pending.
extern "C" int mskf_ekf_update_batch_end(mskf_ctx *ctx) {
    if (!ctx) return MSKF_ERR_INVALID;
    mskf_ctx::PendingUpdate &U = ctx->pend_upd;
    if (!U.active) return MSKF_OK;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    U.active = false;
    if (U.launched) {
        int rc;
        if ((rc = mskf_wait_event(ctx, &U.done, false)) != MSKF_OK) return rc;
        mskf_t_collect(ctx);
    }
    const auto t_h1 = std::chrono::steady_clock::now();
    const int n = U.n;
    mskf_stream *const *streams = U.streams;
    mskf_ekf_update_args *args = U.args;
    const char *hout = ctx->upd_out.h;
    struct LayOut { size_t o_dx, o_gamma, o_rows, o_status, o_pos; };
    std::vector<LayOut> lay(n);
    for (int i = 0; i < n; ++i) lay[i] = LayOut{U.lay[5 * i], U.lay[5 * i + 1], U.lay[5 * i + 2], U.lay[5 * i + 3], U.lay[5 * i + 4]};
    for (int i = 0; i < n; ++i) {
        mskf_ekf_update_args &a = args[i];
        EkfStreamState &E = streams[i]->ekf_state;
        const LayOut &L = lay[i];
        const int d = E.d;
        if (a.pos_var_out) {
            if (U.launched) std::memcpy(a.pos_var_out, hout + L.o_rows + 32, sizeof(double) * 3);
            else a.pos_var_out[0] = a.pos_var_out[1] = a.pos_var_out[2] = -1.0;       // nothing ran: no value (variances are never negative)
        }
        if (!a.n_feat) {
            if (a.delta_x) std::memset(a.delta_x, 0, sizeof(double) * (size_t)d);
            if (a.rows_out) *a.rows_out = 0;
            if (a.diag_out) { a.diag_out[0] = 0; a.diag_out[1] = -1; }
            continue;
        }
        std::memcpy(a.delta_x, hout + L.o_dx, sizeof(double) * (size_t)d);
        if (a.gamma) std::memcpy(a.gamma, hout + L.o_gamma, sizeof(double) * (size_t)a.n_feat);
        *a.rows_out = ((const int *)(hout + L.o_rows))[0];
        if (a.diag_out) { const int dg = ((const int *)(hout + L.o_rows))[3]; a.diag_out[0] = (dg & 4) ? 2 : (dg & 1); a.diag_out[1] = dg >> 8; }
        std::memcpy(a.feat_status, hout + L.o_status, (size_t)a.n_feat);
        const double *po = (const double *)(hout + L.o_pos);
        for (int j = 0; j < a.n_feat; ++j)
            for (int k = 0; k < 3; ++k) a.features[j].position[k] = po[3 * j + k];
    }
    if (ctx->t_gate) ctx->host_s[1] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h1).count();
    return MSKF_OK;
}

// Test / diagnostic access to the work buffers of the last update of a stream (not used by the product path):
// which = 0 Hs (max_rows x ld), 1 rowmask (max_rows, 8-byte), 2 S (ld x ld), 3 T (ld x ld), 4 W (ld x ld), 5 act (ld ints).
// Copies min(capacity, size) bytes; *ld_out = row stride in doubles.
extern "C" int mskf_ekf_debug_read(mskf_stream *s, int which, void *out, size_t capacity, int *ld_out) {
    if (!s || !out) return MSKF_ERR_INVALID;
    EkfStreamState &E = s->ekf_state;
    MSKF_HIPCHK(hipSetDevice(s->ctx_ekf->device));
    MSKF_HIPCHK(hipStreamSynchronize(s->ctx_ekf->stream));
    const void *src = nullptr;
    size_t bytes = 0;
    const size_t pl = sizeof(double) * (size_t)E.ld * E.ld;
    switch (which) {
        case 0: src = E.Hs; bytes = sizeof(double) * (size_t)E.max_rows * E.ld; break;
        case 1: src = E.rs; bytes = sizeof(double) * (size_t)E.max_rows; break;
        case 2: src = E.S; bytes = pl; break;
        case 3: src = E.T; bytes = pl; break;
        case 4: src = E.W; bytes = pl; break;
        case 5: src = E.act; bytes = sizeof(int) * (size_t)E.ld; break;
        default: return MSKF_ERR_INVALID;
    }
    if (!src) return MSKF_ERR_INVALID;
    if (ld_out) *ld_out = E.ld;
    MSKF_HIPCHK(hipMemcpy(out, src, std::min(bytes, capacity), hipMemcpyDeviceToHost));
    return MSKF_OK;
}

extern "C" int mskf_ekf_update(mskf_stream *s, mskf_ekf_update_args *args) {
    if (!s || !args) return MSKF_ERR_INVALID;
    mskf_stream *ss[1] = {s};
    return mskf_ekf_update_batch(s->ctx_ekf, 1, ss, args);
}
