// mskf_capi_fe.cpp — C-ABI: contexts, streams and the front-end entry points (include/mskf_hip.h).
#include <sys/prctl.h>
#include <time.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <mutex>
#include <vector>
#include "mskf_internal.h"

static thread_local std::string g_last_error;
void mskf_set_error(const std::string &s) { g_last_error = s; }

extern "C" const char *mskf_last_error(void) { return g_last_error.c_str(); }
extern "C" int mskf_abi_version(void) { return 4; }   // 4: round 4 (2-point RANSAC inside the device frame: mskf_fe_frame_args.R_p_c / ransac_draws, mskf_fe_set_grid's draw counter)

extern "C" int mskf_ctx_create(int device, mskf_ctx **out) { return mskf_ctx_create_prio(device, 0, out); }

extern "C" int mskf_ctx_create_prio(int device, int high_priority, mskf_ctx **out) {
    if (!out) return MSKF_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mskf_set_error("no HIP device visible (this library has no CPU fallback)");
        return MSKF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { mskf_set_error("device index out of range"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(device));
    mskf_ctx *c = new mskf_ctx();
    c->device = device;
    hipError_t e;
    if (high_priority) {
        int lo = 0, hi = 0;   // numerically lower = more urgent
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) { delete c; mskf_set_error(hipGetErrorString(e)); return MSKF_ERR_HIP; }
    { const char *w = std::getenv("MSKF_WAIT"); c->wait_block = w && std::strcmp(w, "block") == 0; }
    *out = c;
    return MSKF_OK;
}

extern "C" int mskf_ctx_create_shared(mskf_ctx *parent, mskf_ctx **out) {
    if (!parent || !out) return MSKF_ERR_INVALID;
    mskf_ctx *c = new mskf_ctx();
    c->device = parent->device;
    c->stream = parent->stream;
    c->owns_stream = false;
    c->wait_block = parent->wait_block;
    *out = c;
    return MSKF_OK;
}

extern "C" void mskf_ctx_destroy(mskf_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    mskf_t_collect(c);
    for (auto &e : c->t_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (int i = 0; i < 3; ++i) c->desc[i].release();
    c->cell_arena.release(); c->trk_in.release(); c->trk_out.release(); c->upd_in.release(); c->upd_out.release();
    c->jobs.release();
    c->book_desc.release(); c->book_out.release();
    if (c->pend_frame.done) (void)hipEventDestroy(c->pend_frame.done);
    c->ekf_desc.release();
    c->pred_arena.release();
    if (c->pred_done) (void)hipEventDestroy(c->pred_done);
    if (c->wait_ev) (void)hipEventDestroy(c->wait_ev);
    if (c->flag_h) (void)hipHostFree((void *)c->flag_h);
    if (c->cell_ev) (void)hipEventDestroy(c->cell_ev);
    if (c->pend_trk.done) (void)hipEventDestroy(c->pend_trk.done);
    if (c->pend_upd.done) (void)hipEventDestroy(c->pend_upd.done);
    if (c->pend_pv.done) (void)hipEventDestroy(c->pend_pv.done);
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void fe_launch_mark(volatile unsigned int *flag, unsigned int seq, hipStream_t st);
extern "C" size_t fe_book_lds_budget(void);

extern "C" void fe_launch_copy(void *const *dst, const void *const *src, const size_t *bytes, int n_segs, hipStream_t st);
int mskf_copy_async(mskf_ctx *c, const MskfCopy *segs, int n) {
    static const bool sdma = [] { const char *e = std::getenv("MSKF_SDMA_COPIES"); return e && e[0] == '1'; }();
    if (sdma) {
        for (int i = 0; i < n; ++i) if (segs[i].bytes) MSKF_HIPCHK(hipMemcpyAsync(segs[i].dst, segs[i].src, segs[i].bytes, hipMemcpyDefault, c->stream));
        return MSKF_OK;
    }
    for (int i0 = 0; i0 < n; i0 += MSKF_COPY_SEGS) {
        void *dst[MSKF_COPY_SEGS]; const void *src[MSKF_COPY_SEGS]; size_t bytes[MSKF_COPY_SEGS];
        int m = 0;
        for (int i = i0; i < n && i < i0 + MSKF_COPY_SEGS; ++i) {
            if (!segs[i].bytes) continue;
            if (((uintptr_t)segs[i].dst | (uintptr_t)segs[i].src) & 15) { mskf_set_error("staging copy with an unaligned end"); return MSKF_ERR_INVALID; }
            dst[m] = segs[i].dst; src[m] = segs[i].src; bytes[m] = segs[i].bytes; ++m;
        }
        if (m) fe_launch_copy(dst, src, bytes, m, c->stream);
    }
    MSKF_HIPCHK(hipGetLastError());
    return MSKF_OK;
}

int mskf_wait_event(mskf_ctx *c, hipEvent_t *ev_slot, bool record) {
    if (c->wait_block) {
        if (!*ev_slot) MSKF_HIPCHK(hipEventCreateWithFlags(ev_slot, hipEventDisableTiming | hipEventBlockingSync));
        if (record) { MSKF_HIPCHK(hipEventRecord(*ev_slot, c->stream)); return MSKF_OK; }
        MSKF_HIPCHK(hipEventSynchronize(*ev_slot));
        return MSKF_OK;
    }
    if (!c->flag_h) {
        unsigned int *p = nullptr;
        MSKF_HIPCHK(hipHostMalloc((void **)&p, 8 * 64, hipHostMallocCoherent | hipHostMallocMapped));      // one cache line per slot
        std::memset(p, 0, 8 * 64);
        c->flag_h = p;
    }
    int k = 0;
    while (k < 8 && c->flag_slot[k] && c->flag_slot[k] != ev_slot) ++k;
    if (k == 8) { mskf_set_error("too many completion marks"); return MSKF_ERR_INVALID; }
    if (record) c->flag_slot[k] = ev_slot;       // a slot is claimed when its first mark is recorded; waiting only reads the context
    else if (c->flag_slot[k] != ev_slot) { mskf_set_error("waiting for a completion mark that was never recorded"); return MSKF_ERR_INVALID; }
    volatile unsigned int *w = c->flag_h + 16 * k;
    if (record) {
        // the mark is a stream write-value command (no dispatch: a one-thread kernel waits 37 us for a CU slot on a busy
        // device, profiles/r02_kernel_stats.csv of the kernel-mark build); a runtime that refuses the command on pinned host
        // memory falls back to the one-thread kernel k_mark
        static std::atomic<bool> use_write{true};   // (contexts are driven from several host threads)
        ++c->flag_seq[k];
        if (use_write.load(std::memory_order_relaxed)) {
            if (hipStreamWriteValue32(c->stream, (void *)w, c->flag_seq[k], 0) == hipSuccess) return MSKF_OK;
            (void)hipGetLastError();
            use_write.store(false, std::memory_order_relaxed);
        }
        fe_launch_mark(w, c->flag_seq[k], c->stream);
        MSKF_HIPCHK(hipGetLastError());
        return MSKF_OK;
    }
    const unsigned int want = c->flag_seq[k];
    // a mark that never arrives (device fault, lost queue) must not hang the caller for ever: after MSKF_WAIT_TIMEOUT_S
    // seconds (default 120) the stream is asked for its error state and the wait fails
    static const double limit_s = [] { const char *e = std::getenv("MSKF_WAIT_TIMEOUT_S"); const double v = e ? std::atof(e) : 120.0; return v > 0 ? v : 120.0; }();
    // MSKF_WAIT=nap[:us]: the same mark, polled between short sleeps (default 40 us) instead of spun on: a wait of some
    // milliseconds then costs the core a few per cent of its time and the waiter at most one sleep of latency, which leaves a
    // host that runs under a CPU quota its cores for the other groups' host phases
    static const long nap_ns = [] {
        const char *e = std::getenv("MSKF_WAIT");
        if (!e || std::strncmp(e, "nap", 3) != 0) return 0L;
        const long us = e[3] == ':' ? std::atol(e + 4) : 40L;
        return 1000L * (us > 0 ? us : 40L);
    }();
    if (nap_ns) {
        static thread_local bool slack_set = false;
        if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); slack_set = true; }     // (the default slack of 50 us would double every sleep)
    }
    unsigned long long spins = 0, naps = 0;
    bool timing = false;
    std::chrono::steady_clock::time_point t0;
    while ((int)(__atomic_load_n((const unsigned int *)w, __ATOMIC_ACQUIRE) - want) < 0) {
        bool check;
        if (nap_ns && spins >= 64) { const struct timespec ts = {0, nap_ns}; (void)nanosleep(&ts, nullptr); check = (++naps & 0xFFULL) == 0; }
        else { __builtin_ia32_pause(); check = (++spins & 0xFFFFFULL) == 0; }
        if (check) {                                             // about every 10 ms
            const auto now = std::chrono::steady_clock::now();
            if (!timing) { t0 = now; timing = true; }
            else if (std::chrono::duration<double>(now - t0).count() > limit_s) {
                const hipError_t e = hipStreamQuery(c->stream);
                mskf_set_error(e != hipSuccess && e != hipErrorNotReady ? hipGetErrorString(e) : "completion mark not written within the wait limit");
                std::fprintf(stderr, "mskf_wait_event: ctx %p slot %d mark %u wanted %u, stream query %d\n", (void *)c, k, *w, want, (int)e);
                return MSKF_ERR_HIP;
            }
        }
    }
    return MSKF_OK;
}
int mskf_wait(mskf_ctx *c) {
    int rc = mskf_wait_event(c, &c->wait_ev, true);
    if (rc != MSKF_OK) return rc;
    return mskf_wait_event(c, &c->wait_ev, false);
}

extern "C" int mskf_ctx_sync(mskf_ctx *c) {
    if (!c) return MSKF_ERR_INVALID;
    { const int rc = mskf_wait(c); if (rc != MSKF_OK) return rc; }
    mskf_t_collect(c);
    return MSKF_OK;
}

int mskf_t_begin(mskf_ctx *c, int kind) {
    if (!c->timing || !c->t_gate) return -1;
    if ((c->t_all[kind]++ % c->timing_period) != 0) return -1;      // sampled: the events themselves cost device and host time
    TimingSlot t;
    if (!c->t_pool.empty()) { t.a = c->t_pool.back().first; t.b = c->t_pool.back().second; c->t_pool.pop_back(); }
    else {
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return -1;
    }
    t.kind = kind; t.units = 0;
    (void)hipEventRecord(t.a, c->stream);
    c->t_pending.push_back(t);
    return (int)c->t_pending.size() - 1;
}
void mskf_t_end(mskf_ctx *c, int slot, long long units) {
    if (slot < 0) return;
    (void)hipEventRecord(c->t_pending[slot].b, c->stream);
    c->t_pending[slot].units = units;
}
// Units of a slot that was begun earlier: the slot index is only valid until the next mskf_t_collect (any synchronising
// call of the context collects), so it is checked against the pending list and the kind it was begun with.
void mskf_t_set_units(mskf_ctx *c, int slot, int kind, long long units) {
    if (slot < 0 || (size_t)slot >= c->t_pending.size() || c->t_pending[slot].kind != kind) return;
    c->t_pending[slot].units = units;
}
void mskf_t_collect(mskf_ctx *c) {
    for (auto &t : c->t_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            c->t_ms[t.kind] += ms; c->t_launches[t.kind] += 1; c->t_units[t.kind] += t.units;
        }
        c->t_pool.push_back({t.a, t.b});
    }
    c->t_pending.clear();
}

extern "C" int mskf_ctx_set_timing(mskf_ctx *c, int enable) {
    if (!c) return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(c->device));
    MSKF_HIPCHK(hipStreamSynchronize(c->stream));
    mskf_t_collect(c);
    c->timing = enable != 0;
    c->timing_period = enable > 1 ? enable : 1;
    return MSKF_OK;
}

extern "C" int mskf_ctx_set_wait_mode(mskf_ctx *c, int block) {
    if (!c) return MSKF_ERR_INVALID;
    if (c->pend_trk.active || c->pend_upd.active || c->pend_pv.active || c->pend_frame.active) { mskf_set_error("a batch of this context is pending"); return MSKF_ERR_INVALID; }
    c->wait_block = block != 0;
    return MSKF_OK;
}

extern "C" int mskf_ctx_timing_gate(mskf_ctx *c, int on) {
    if (!c) return MSKF_ERR_INVALID;
    c->t_gate = on != 0;          // no synchronisation: launches already begun keep the state they were begun with
    return MSKF_OK;
}

extern "C" int mskf_ctx_get_timing(mskf_ctx *c, double *ms, long long *launches, long long *units, int reset) {
    if (!c || !ms || !launches || !units) return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(c->device));
    MSKF_HIPCHK(hipStreamSynchronize(c->stream));
    mskf_t_collect(c);
    for (int k = 0; k < MSKF_K_COUNT; ++k) {
        // sampled timing: the sums are scaled from the timed launches to all launches of the kind
        const double sc = c->t_launches[k] > 0 ? (double)c->t_all[k] / (double)c->t_launches[k] : 0.0;
        ms[k] = c->t_ms[k] * sc; launches[k] = c->t_launches[k] > 0 ? c->t_all[k] : 0; units[k] = (long long)((double)c->t_units[k] * sc);
        if (reset) { c->t_ms[k] = 0; c->t_launches[k] = 0; c->t_units[k] = 0; c->t_all[k] = 0; }
    }
    return MSKF_OK;
}

extern "C" void *mskf_ctx_hip_stream(mskf_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int mskf_ctx_get_host_time(mskf_ctx *c, double out[4], int reset) {
    if (!c || !out) return MSKF_ERR_INVALID;
    for (int k = 0; k < 4; ++k) { out[k] = c->host_s[k]; if (reset) c->host_s[k] = 0; }
    return MSKF_OK;
}

void fill_pyr(const mskf_stream *s, int idx, PyrDev &p) {
    for (int l = 0; l < MSKF_LEVELS; ++l) {
        p.lvl[l] = s->pyr[idx] + s->lvl_off[l];
        p.w[l] = s->lw[l];
        p.h[l] = s->lh[l];
    }
    if (s->lvl0[idx]) p.lvl[0] = s->lvl0[idx];
}

// Device-side books of a stream (fe_book.h): the three feature lists, detection / candidate lists and the results of the two
// track calls, in ONE allocation.  Streams whose grid_min / grid_max exceed the short-list bound of the kernels keep their
// books on the host (cap stays 0).
static int book_alloc(mskf_stream *s) {
    mskf_stream::Book &K = s->book;
    const mskf_fe_cfg &fe = s->fe;
    if (fe.grid_min_feature_num > FB_MAXK || fe.grid_max_feature_num > FB_MAXK || fe.grid_min_feature_num < 0 ||
        fe.grid_max_feature_num < fe.grid_min_feature_num) return MSKF_OK;
    K.grid_h = s->h / fe.grid_row; K.grid_w = s->w / fe.grid_col;                 // image_processor.cpp:250-251
    if (K.grid_h <= 0 || K.grid_w <= 0) return MSKF_OK;
    K.n_cells = fe.grid_row * fe.grid_col;
    K.n_codes = std::max(((s->h - 1) / K.grid_h) * fe.grid_col + (s->w - 1) / K.grid_w + 1, K.n_cells);   // Q7: partial rows / columns
    const int cap = K.n_codes * std::max(fe.grid_max_feature_num, 1) + 8;
    const int cand_cap = K.n_cells * std::max(fe.grid_max_feature_num, 1) + 8;
    const int det_cap = fe.det_rows * fe.det_cols;
    // the bookkeeping kernel keeps its lists in LDS: a configuration whose lists do not fit the budget the kernel can get
    // (a very fine detector grid) keeps its books on the host
    if (4 * fe_book_scratch_ints(cap, cand_cap, det_cap, K.n_codes, det_cap) > fe_book_lds_budget()) return MSKF_OK;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_st = take(sizeof(FeBookState));
    size_t o_grid[3][8];
    for (int g = 0; g < 3; ++g) {
        o_grid[g][0] = take(8 * (size_t)cap); o_grid[g][1] = take(4 * (size_t)cap); o_grid[g][2] = take(4 * (size_t)cap); o_grid[g][3] = take(4 * (size_t)cap);
        for (int q = 4; q < 8; ++q) o_grid[g][q] = take(8 * (size_t)cap);
    }
    const size_t o_det_pt = take(8 * (size_t)det_cap), o_det_sc = take(4 * (size_t)det_cap);
    const size_t o_cpt = take(8 * (size_t)cand_cap), o_cidx = take(4 * (size_t)cand_cap), o_csc = take(4 * (size_t)cand_cap);
    const size_t o_coff = take(4 * (size_t)(K.n_cells + 1)), o_ccnt = take(4 * (size_t)(K.n_cells + 1)), o_cell = take(4 * (size_t)(K.n_codes + 1));
    size_t o_t[5], o_c[5];
    for (int q = 0; q < 4; ++q) o_t[q] = take(8 * (size_t)cap);
    o_t[4] = take((size_t)cap);
    for (int q = 0; q < 4; ++q) o_c[q] = take(8 * (size_t)cand_cap);
    o_c[4] = take((size_t)cand_cap);
    const size_t o_rs_pair = take(8 * 4 * (size_t)cap), o_rs_pt = take(4 * 4 * (size_t)cap), o_rs_sc = take(8 * 48);
    MSKF_HIPCHK(hipMalloc((void **)&K.mem, off));
    MSKF_HIPCHK(hipMemsetAsync(K.mem, 0, off, s->ctx->stream));
    MSKF_HIPCHK(hipStreamSynchronize(s->ctx->stream));
    char *m = K.mem;
    K.st = (FeBookState *)(m + o_st);
    for (int g = 0; g < 3; ++g)
        K.grid[g] = FeGridArr{(unsigned long long *)(m + o_grid[g][0]), (int *)(m + o_grid[g][1]), (int *)(m + o_grid[g][2]), (float *)(m + o_grid[g][3]),
                              (mskf_point2f *)(m + o_grid[g][4]), (mskf_point2f *)(m + o_grid[g][5]), (mskf_point2f *)(m + o_grid[g][6]), (mskf_point2f *)(m + o_grid[g][7])};
    K.det_pt = (mskf_point2f *)(m + o_det_pt); K.det_score = (int *)(m + o_det_sc);
    K.cand_pt = (mskf_point2f *)(m + o_cpt); K.cand_index = (int *)(m + o_cidx); K.cand_score = (int *)(m + o_csc);
    K.cand_off = (int *)(m + o_coff); K.cand_cnt = (int *)(m + o_ccnt); K.cell_count = (int *)(m + o_cell);
    K.t_out0 = (mskf_point2f *)(m + o_t[0]); K.t_out1 = (mskf_point2f *)(m + o_t[1]); K.t_und0 = (mskf_point2f *)(m + o_t[2]); K.t_und1 = (mskf_point2f *)(m + o_t[3]);
    K.t_status = (uint8_t *)(m + o_t[4]);
    K.c_out0 = (mskf_point2f *)(m + o_c[0]); K.c_out1 = (mskf_point2f *)(m + o_c[1]); K.c_und0 = (mskf_point2f *)(m + o_c[2]); K.c_und1 = (mskf_point2f *)(m + o_c[3]);
    K.c_status = (uint8_t *)(m + o_c[4]);
    K.rs_pair = (double *)(m + o_rs_pair); K.rs_pt = (float *)(m + o_rs_pt); K.rs_scalar = (double *)(m + o_rs_sc);
    K.cap = cap; K.cand_cap = cand_cap; K.det_cap = det_cap;
    return MSKF_OK;
}

extern "C" int mskf_stream_create(mskf_ctx *ctx, const mskf_calib *calib, const mskf_fe_cfg *fe, const mskf_ekf_cfg *ekf,
                                  mskf_stream **out) {
    if (!ctx || !calib || !fe || !ekf || !out) return MSKF_ERR_INVALID;
    for (int m : {calib->cam0_model, calib->cam1_model})
        if (m != MSKF_MODEL_RADTAN && m != MSKF_MODEL_EQUIDISTANT) {
            mskf_set_error("unknown distortion model (radtan and equidistant are implemented)");
            return MSKF_ERR_UNSUPPORTED;
        }
    if (calib->width < 64 || calib->height < 64 || fe->det_rows <= 0 || fe->det_cols <= 0 || fe->grid_row <= 0 || fe->grid_col <= 0)
        return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    mskf_stream *s = new mskf_stream();
    s->ctx = ctx;
    s->ctx_ekf = ctx;
    s->home_ctx = ctx;
    s->calib = *calib; s->fe = *fe; s->ekf = *ekf;
    s->w = calib->width; s->h = calib->height;
    size_t off = 0;
    int w = s->w, h = s->h;
    for (int l = 0; l < MSKF_LEVELS; ++l) {
        s->lw[l] = w; s->lh[l] = h; s->lvl_off[l] = off;
        off += ((size_t)w * h + 255) & ~(size_t)255;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    s->pyr_bytes = off;
    int rc = MSKF_OK;
    for (int i = 0; i < 3 && rc == MSKF_OK; ++i) {
        hipError_t e = hipMalloc((void **)&s->pyr[i], s->pyr_bytes);
        if (e != hipSuccess) { mskf_set_error(hipGetErrorString(e)); rc = MSKF_ERR_HIP; }
    }
    // point capacity: every live grid slot plus every detector cell
    const int det_cells = fe->det_rows * fe->det_cols;
    s->pt_cap = (fe->grid_row + 1) * (fe->grid_col + 1) * (fe->grid_max_feature_num + 1) + det_cells + 64;
    if (rc == MSKF_OK) rc = mskf_ekf_stream_init(s);
    if (rc != MSKF_OK) { mskf_stream_destroy(s); return rc; }

    for (int i = 0; i < 4; ++i) {
        s->cam0.K[i] = calib->cam0_intrinsics[i]; s->cam0.D[i] = calib->cam0_distortion[i];
        s->cam0.model = calib->cam0_model; s->cam1.model = calib->cam1_model;
        s->cam1.K[i] = calib->cam1_intrinsics[i]; s->cam1.D[i] = calib->cam1_distortion[i];
    }
    // image_processor.cpp:63-72 (loadParameters) and :544,:587-591 (stereoMatch)
    using namespace hm;
    Rigid m4_cam0_imu = Rigid::from_rowmajor16(calib->T_cam0_imu);
    Mat3 R_cam0_imu = m4_cam0_imu.R.transpose();
    Vec3 t_cam0_imu = -(R_cam0_imu * m4_cam0_imu.t);
    Rigid m4_cam1_cam0 = Rigid::from_rowmajor16(calib->T_cam1_cam0);
    Rigid T_cam1_imu = m4_cam1_cam0 * m4_cam0_imu;
    Mat3 R_cam1_imu = T_cam1_imu.R.transpose();
    Vec3 t_cam1_imu = -(R_cam1_imu * T_cam1_imu.t);
    Mat3 R_cam0_cam1 = R_cam1_imu.transpose() * R_cam0_imu;
    Vec3 t_cam0_cam1 = R_cam1_imu.transpose() * (t_cam0_imu - t_cam1_imu);
    Mat3 E = skew(t_cam0_cam1) * R_cam0_cam1;
    std::memcpy(s->R01, R_cam0_cam1.m, sizeof(s->R01));
    std::memcpy(s->E, E.m, sizeof(s->E));
    const double norm_pixel_unit = 4.0 / (s->cam0.K[0] + s->cam0.K[1] + s->cam1.K[0] + s->cam1.K[1]);
    s->epi_thresh = fe->stereo_threshold * norm_pixel_unit;
    s->det_ch = (s->h + fe->det_rows - 1) / fe->det_rows;
    s->det_cw = (s->w + fe->det_cols - 1) / fe->det_cols;
    rc = book_alloc(s);
    if (rc != MSKF_OK) { mskf_stream_destroy(s); return rc; }
    ctx->streams.push_back(s);
    *out = s;
    return MSKF_OK;
}

extern "C" mskf_ctx *mskf_stream_ctx(mskf_stream *s) { return s ? s->ctx : nullptr; }
extern "C" mskf_ctx *mskf_stream_ekf_ctx(mskf_stream *s) { return s ? s->ctx_ekf : nullptr; }
extern "C" int mskf_stream_set_ekf_ctx(mskf_stream *s, mskf_ctx *c) {
    if (!s || !c || c->device != s->ctx->device) return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(c->device));
    MSKF_HIPCHK(hipStreamSynchronize(s->ctx_ekf->stream));
    MSKF_HIPCHK(hipStreamSynchronize(c->stream));
    s->ctx_ekf = c;
    return MSKF_OK;
}

extern "C" int mskf_stream_rebind(mskf_stream *s, mskf_ctx *fe_ctx, mskf_ctx *ekf_ctx) {
    if (!s) return MSKF_ERR_INVALID;
    if ((fe_ctx && fe_ctx->device != s->home_ctx->device) || (ekf_ctx && ekf_ctx->device != s->home_ctx->device)) return MSKF_ERR_INVALID;
    if (fe_ctx) s->ctx = fe_ctx;
    if (ekf_ctx) s->ctx_ekf = ekf_ctx;
    return MSKF_OK;
}

struct mskf_point { hipEvent_t ev = nullptr; };
extern "C" int mskf_ctx_record_point(mskf_ctx *ctx, mskf_point **point) {
    if (!ctx || !point) return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    if (!*point) { *point = new mskf_point(); MSKF_HIPCHK(hipEventCreateWithFlags(&(*point)->ev, hipEventDisableTiming)); }
    MSKF_HIPCHK(hipEventRecord((*point)->ev, ctx->stream));
    return MSKF_OK;
}
extern "C" int mskf_ctx_wait_point(mskf_ctx *ctx, mskf_point *point) {
    if (!ctx || !point || !point->ev) return MSKF_ERR_INVALID;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    MSKF_HIPCHK(hipStreamWaitEvent(ctx->stream, point->ev, 0));
    return MSKF_OK;
}
extern "C" void mskf_point_destroy(mskf_point *point) {
    if (!point) return;
    if (point->ev) (void)hipEventDestroy(point->ev);
    delete point;
}

extern "C" void mskf_stream_destroy(mskf_stream *s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->ctx_ekf && s->ctx_ekf != s->ctx) (void)hipStreamSynchronize(s->ctx_ekf->stream);
    for (int i = 0; i < 3; ++i) if (s->pyr[i]) (void)hipFree(s->pyr[i]);
    if (s->book.mem) (void)hipFree(s->book.mem);
    mskf_ekf_stream_free(s);
    auto &v = s->home_ctx->streams;
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == s) { v.erase(v.begin() + i); break; }
    delete s;
}

static void fill_fe_desc(const mskf_stream *s, FeStreamDev &d) {
    std::memset(&d, 0, sizeof(d));
    fill_pyr(s, s->i_prev0, d.prev0);
    fill_pyr(s, s->i_curr0, d.curr0);
    fill_pyr(s, s->i_curr1, d.curr1);
    d.cam0 = s->cam0; d.cam1 = s->cam1;
    std::memcpy(d.R01, s->R01, sizeof(d.R01));
    std::memcpy(d.E, s->E, sizeof(d.E));
    d.epi_thresh = s->epi_thresh;
    d.det_rows = s->fe.det_rows; d.det_cols = s->fe.det_cols; d.cell_w = s->det_cw; d.cell_h = s->det_ch;
    d.det_floor = s->det_floor;
    d.cell_keys = (unsigned long long *)(s->ctx->cell_arena.d + s->cell_off);
}

static int push_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, bool copy_cells);

extern "C" int mskf_fe_push_stereo_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const uint8_t *const *cam0,
                                         const uint8_t *const *cam1, int on_device) {
    return push_batch(ctx, n, streams, cam0, cam1, on_device, true);
}

// copy_cells = false: the per-cell maxima stay on the device (the bookkeeping kernel of a device frame reads them there)
static int push_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, bool copy_cells) {
    if (!ctx || n <= 0 || !streams || !cam0 || !cam1) return MSKF_ERR_INVALID;
    // the pinned staging of a pending batch (descriptors, pyramid jobs, the cell arena) may still be in flight
    if (ctx->pend_frame.active || ctx->pend_trk.active) { mskf_set_error("a batch of this context is still pending"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = ctx->jobs.ensure((size_t)n * 2);
    if (rc != MSKF_OK) return rc;
    rc = ctx->desc[0].ensure(n);
    if (rc != MSKF_OK) return rc;
    int max_w = 0, max_h = 0;
    size_t cell_bytes = 0;
    {
        for (int i = 0; i < n; ++i) { if (!streams[i] || streams[i]->ctx != ctx) return MSKF_ERR_INVALID; cell_bytes += sizeof(unsigned long long) * (size_t)streams[i]->fe.det_rows * streams[i]->fe.det_cols; }
        if (cell_bytes > ctx->cell_arena.cap) { MSKF_HIPCHK(hipStreamSynchronize(st)); rc = ctx->cell_arena.ensure(cell_bytes); if (rc != MSKF_OK) return rc; ctx->cell_keys_dirty = true; }
        ++ctx->push_gen;
        size_t off = 0;
        for (int i = 0; i < n; ++i) { streams[i]->cell_off = off; streams[i]->push_gen = ctx->push_gen; off += sizeof(unsigned long long) * (size_t)streams[i]->fe.det_rows * streams[i]->fe.det_cols; }
    }
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        if (!s || s->ctx != ctx || !cam0[i] || !cam1[i]) return MSKF_ERR_INVALID;
        if (on_device == 3) {
            // level 0 already sits in the stream's own planes (mskf_fe_push_stereo with padded rows)
            if (cam0[i] != s->pyr[s->i_curr0] || cam1[i] != s->pyr[s->i_curr1]) return MSKF_ERR_INVALID;
            s->lvl0[s->i_curr0] = nullptr; s->lvl0[s->i_curr1] = nullptr;
        } else if (on_device == 2) {
            // borrowed device images: level 0 is read in place (caller keeps them valid and unchanged until the
            // second-next push of this stream: the previous frame's cam0 is the LK template of the next frame)
            s->lvl0[s->i_curr0] = cam0[i];
            s->lvl0[s->i_curr1] = cam1[i];
        } else {
            const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
            s->lvl0[s->i_curr0] = nullptr; s->lvl0[s->i_curr1] = nullptr;
            MSKF_HIPCHK(hipMemcpyAsync(s->pyr[s->i_curr0], cam0[i], (size_t)s->w * s->h, kind, st));
            MSKF_HIPCHK(hipMemcpyAsync(s->pyr[s->i_curr1], cam1[i], (size_t)s->w * s->h, kind, st));
        }
        s->has_curr = true;
        max_w = std::max(max_w, s->w); max_h = std::max(max_h, s->h);
    }
    // pyramid levels 1..3 of both cameras of every stream: ONE launch (k_pyr_down3)
    static_assert(MSKF_LEVELS == 4, "k_pyr_down3 builds exactly the levels 1, 2, 3");
    {
        long long px = 0;
        for (int i = 0; i < n; ++i) {
            mskf_stream *s = streams[i];
            for (int c = 0; c < 2; ++c) {
                Pyr3Job &j = ctx->jobs.h[2 * (size_t)i + c];
                const int pi = c == 0 ? s->i_curr0 : s->i_curr1;
                uint8_t *base = s->pyr[pi];
                j.src = s->lvl0[pi] ? s->lvl0[pi] : base + s->lvl_off[0];
                j.d1 = base + s->lvl_off[1]; j.d2 = base + s->lvl_off[2]; j.d3 = base + s->lvl_off[3];
                j.w0 = s->lw[0]; j.h0 = s->lh[0];
            }
            for (int l = 1; l < MSKF_LEVELS; ++l) px += 2LL * s->lw[l] * s->lh[l];
        }
        // (one staging launch for the pyramid jobs and the detector's descriptors)
        for (int i = 0; i < n; ++i) fill_fe_desc(streams[i], ctx->desc[0].h[i]);
        const MskfCopy cp[2] = {{ctx->jobs.d, ctx->jobs.h, sizeof(Pyr3Job) * 2 * (size_t)n}, {ctx->desc[0].d, ctx->desc[0].h, sizeof(FeStreamDev) * (size_t)n}};
        if ((rc = mskf_copy_async(ctx, cp, 2)) != MSKF_OK) return rc;
        const int ts = mskf_t_begin(ctx, MSKF_K_PYR);
        fe_launch_pyr_down3(ctx->jobs.d, 2 * n, max_w, max_h, st);
        mskf_t_end(ctx, ts, px);          // units: output pixels of the three levels
    }
    // detector per-cell maxima on cam0 level 0
    {
        // the keys carry the push generation in their top byte: a newer push wins every atomicMax, so the key array is
        // cleared only when it is fresh or the 8-bit generation wraps (not once per frame)
        const unsigned int gen = (unsigned int)((ctx->push_gen - 1) % 255ULL) + 1U;
        if (ctx->cell_keys_dirty || gen == 1U) {
            MSKF_HIPCHK(hipMemsetAsync(ctx->cell_arena.d, 0, ctx->cell_arena.cap, st));
            ctx->cell_keys_dirty = false;
        }
        long long px = 0;
        for (int i = 0; i < n; ++i) px += (long long)streams[i]->w * streams[i]->h;
        const int ts = mskf_t_begin(ctx, MSKF_K_DETECT);
        fe_launch_detect(ctx->desc[0].d, n, max_w, max_h, gen, st);
        mskf_t_end(ctx, ts, px);
    }
    if (copy_cells) {
        { const MskfCopy cp = {ctx->cell_arena.h, ctx->cell_arena.d, cell_bytes}; const int crc = mskf_copy_async(ctx, &cp, 1); if (crc != MSKF_OK) return crc; }
        { const int erc = mskf_wait_event(ctx, &ctx->cell_ev, true); if (erc != MSKF_OK) return erc; }
    }
    MSKF_HIPCHK(hipGetLastError());
    return MSKF_OK;
}

extern "C" int mskf_fe_push_stereo(mskf_stream *s, const uint8_t *cam0, const uint8_t *cam1, int width, int height, int pitch,
                                   double time_stamp) {
    if (!s || !cam0 || !cam1) return MSKF_ERR_INVALID;
    if (width != s->w || height != s->h || pitch < width) { mskf_set_error("image size differs from the calibration"); return MSKF_ERR_INVALID; }
    s->time_stamp = time_stamp;
    if (pitch != width) {
        // padded rows: 2D copies straight into the stream's own level-0 planes (dense, pitch = width), then the batch
        // path with "level 0 already resident" (on_device = 3): no temporary allocation, nothing to free or leak
        MSKF_HIPCHK(hipSetDevice(s->ctx->device));
        MSKF_HIPCHK(hipMemcpy2DAsync(s->pyr[s->i_curr0], width, cam0, pitch, width, height, hipMemcpyHostToDevice, s->ctx->stream));
        MSKF_HIPCHK(hipMemcpy2DAsync(s->pyr[s->i_curr1], width, cam1, pitch, width, height, hipMemcpyHostToDevice, s->ctx->stream));
        const uint8_t *a[1] = {s->pyr[s->i_curr0]}, *b[1] = {s->pyr[s->i_curr1]};
        mskf_stream *ss[1] = {s};
        return mskf_fe_push_stereo_batch(s->ctx, 1, ss, a, b, 3);
    }
    const uint8_t *a[1] = {cam0}, *b[1] = {cam1};
    mskf_stream *ss[1] = {s};
    return mskf_fe_push_stereo_batch(s->ctx, 1, ss, a, b, 0);
}

extern "C" int mskf_fe_push_stereo_device(mskf_stream *s, const uint8_t *d_cam0, const uint8_t *d_cam1, int width, int height,
                                          double time_stamp) {
    if (!s || !d_cam0 || !d_cam1) return MSKF_ERR_INVALID;
    if (width != s->w || height != s->h) return MSKF_ERR_INVALID;
    s->time_stamp = time_stamp;
    const uint8_t *a[1] = {d_cam0}, *b[1] = {d_cam1};
    mskf_stream *ss[1] = {s};
    return mskf_fe_push_stereo_batch(s->ctx, 1, ss, a, b, 1);
}

// keys -> corners: gen (8) | score (24) | ~order (32), order = row-major position inside the cell; a key of another
// generation (an older push, or 0) means no corner in this one
static inline unsigned int stream_gen(const mskf_stream *s) { return (unsigned int)((s->push_gen - 1) % 255ULL) + 1U; }
static inline long long score_of_key(unsigned long long k, unsigned int gen) { return (k >> 56) == gen ? (long long)((k >> 32) & 0xFFFFFFULL) : 0LL; }
static inline void corner_of_key(const mskf_stream *s, int cell, unsigned long long k, mskf_corner &o) {
    o.cell = cell;
    const long long sc = score_of_key(k, stream_gen(s));
    if (sc == 0) { o.x = 0.f; o.y = 0.f; o.score = 0; return; }
    const int cols = s->fe.det_cols, cw = s->det_cw, ch = s->det_ch;
    const unsigned int order = 0xFFFFFFFFu - (unsigned int)(k & 0xFFFFFFFFULL);
    const int cy = cell / cols, cx = cell - cy * cols;
    o.score = (int)sc;
    o.y = (float)(cy * ch + (int)(order / (unsigned)cw));
    o.x = (float)(cx * cw + (int)(order % (unsigned)cw));
}

static int cell_keys_ready(mskf_stream *s) {
    if (!s->has_curr) { mskf_set_error("no stereo pair pushed yet"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(s->ctx->device));
    if (s->push_gen != s->ctx->push_gen) { mskf_set_error("cell maxima are stale: another push happened on this context"); return MSKF_ERR_INVALID; }
    // wait for the copy of this push's maxima only (not the whole stream: on a shared stream another context may have
    // queued later work); no ctx mutation here: callable concurrently for different streams
    return mskf_wait_event(s->ctx, &s->ctx->cell_ev, false);
}

extern "C" int mskf_fe_set_detect_floor(mskf_stream *s, int min_score) {
    if (!s || min_score < 0 || min_score >= (1 << 24)) return MSKF_ERR_INVALID;
    s->det_floor = min_score;           // takes effect with the next push
    return MSKF_OK;
}

extern "C" int mskf_fe_get_cell_maxima(mskf_stream *s, mskf_corner *out, int capacity, int *n_out) {
    if (!s || !out || !n_out) return MSKF_ERR_INVALID;
    const int n = s->fe.det_rows * s->fe.det_cols;
    if (capacity < n) return MSKF_ERR_CAPACITY;
    const int rc = cell_keys_ready(s);
    if (rc != MSKF_OK) return rc;
    const unsigned long long *keys = (const unsigned long long *)(s->ctx->cell_arena.h + s->cell_off);
    for (int cell = 0; cell < n; ++cell) corner_of_key(s, cell, keys[cell], out[cell]);
    *n_out = n;
    return MSKF_OK;
}

extern "C" int mskf_fe_get_cell_candidates(mskf_stream *s, int min_score, mskf_corner *out, int capacity, int *n_out) {
    if (!s || !out || !n_out) return MSKF_ERR_INVALID;
    const int n = s->fe.det_rows * s->fe.det_cols;
    if (min_score < s->det_floor) { mskf_set_error("min_score is below the stream's detector floor (mskf_fe_set_detect_floor)"); return MSKF_ERR_INVALID; }
    const int rc = cell_keys_ready(s);
    if (rc != MSKF_OK) return rc;
    const unsigned long long *keys = (const unsigned long long *)(s->ctx->cell_arena.h + s->cell_off);
    const unsigned int gen = stream_gen(s);
    int m = 0;
    for (int cell = 0; cell < n; ++cell) {
        const unsigned long long k = keys[cell];
        if (score_of_key(k, gen) <= (long long)min_score) continue;       // also skips empty cells (no key of this generation)
        if (m >= capacity) return MSKF_ERR_CAPACITY;
        corner_of_key(s, cell, k, out[m++]);
    }
    *n_out = m;
    return MSKF_OK;
}

extern "C" int mskf_fe_track_batch(mskf_ctx *ctx, int n, mskf_stream *const *streams, const mskf_fe_track_args *args) {
    const int rc = mskf_fe_track_batch_begin(ctx, n, streams, args);
    return rc != MSKF_OK ? rc : mskf_fe_track_batch_end(ctx);
}

extern "C" int mskf_fe_track_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, const mskf_fe_track_args *args) {
    if (!ctx || n <= 0 || !streams || !args) return MSKF_ERR_INVALID;
    if (ctx->pend_trk.active) { mskf_set_error("a track batch of this context is still pending (call mskf_fe_track_batch_end)"); return MSKF_ERR_INVALID; }
    if (ctx->pend_frame.active) { mskf_set_error("a device frame of this context is still pending (call mskf_fe_frame_batch_end)"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    const auto t_h0 = std::chrono::steady_clock::now();
    hipStream_t st = ctx->stream;
    int rc = ctx->desc[1].ensure(n);
    if (rc != MSKF_OK) return rc;
    int max_pts = 0;
    size_t in_bytes = 0, out_bytes = 0;
    std::vector<size_t> in_off(n);
    std::vector<size_t> &out_off = ctx->pend_trk.out_off;
    out_off.resize(n);
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        const mskf_fe_track_args &a = args[i];
        if (!s || s->ctx != ctx || a.n < 0) return MSKF_ERR_INVALID;
        if (a.n > s->pt_cap) { mskf_set_error("too many points for this stream"); return MSKF_ERR_CAPACITY; }
        if (a.n > 0 && (!a.in_pts || !a.out0 || !a.out1 || !a.und0 || !a.und1 || !a.status)) return MSKF_ERR_INVALID;
        if (!s->has_curr) { mskf_set_error("no stereo pair pushed yet"); return MSKF_ERR_INVALID; }
        in_off[i] = in_bytes; out_off[i] = out_bytes;
        const size_t np = (size_t)a.n;
        in_bytes += (sizeof(mskf_point2f) * np + 63) & ~(size_t)63;
        out_bytes += (4 * sizeof(mskf_point2f) * np + np + 63) & ~(size_t)63;   // out0 out1 und0 und1 status
        max_pts = std::max(max_pts, a.n);
    }
    if (max_pts <= 0) return MSKF_OK;      // nothing to track: no batch pending, _end is a no-op
    if (in_bytes > ctx->trk_in.cap || out_bytes > ctx->trk_out.cap) {
        MSKF_HIPCHK(hipStreamSynchronize(st));
        if ((rc = ctx->trk_in.ensure(in_bytes)) != MSKF_OK) return rc;
        if ((rc = ctx->trk_out.ensure(out_bytes)) != MSKF_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        const mskf_fe_track_args &a = args[i];
        const size_t np = (size_t)a.n;
        FeStreamDev &d = ctx->desc[1].h[i];
        fill_fe_desc(s, d);
        d.n_pts = a.n;
        d.do_temporal = a.do_temporal;
        std::memcpy(d.Hpred, a.Hpred, sizeof(d.Hpred));
        if (np) std::memcpy(ctx->trk_in.h + in_off[i], a.in_pts, sizeof(mskf_point2f) * np);
        d.in_pts = (const mskf_point2f *)(ctx->trk_in.d + in_off[i]);
        char *o = ctx->trk_out.d + out_off[i];
        d.out0 = (mskf_point2f *)o; d.out1 = d.out0 + np; d.und0 = d.out1 + np; d.und1 = d.und0 + np;
        d.status = (uint8_t *)(d.und1 + np);
    }
    {
        const MskfCopy cp[2] = {{ctx->trk_in.d, ctx->trk_in.h, in_bytes}, {ctx->desc[1].d, ctx->desc[1].h, sizeof(FeStreamDev) * (size_t)n}};
        if ((rc = mskf_copy_async(ctx, cp, 2)) != MSKF_OK) return rc;
    }
    // temporal track -> bounds gate + stereo guess -> stereo track -> gates + undistortion: one launch (k_track4)
    const int ts = mskf_t_begin(ctx, MSKF_K_LK);
    fe_launch_track(ctx->desc[1].d, n, max_pts, st);
    mskf_t_end(ctx, ts, 0);
    { const MskfCopy cp = {ctx->trk_out.h, ctx->trk_out.d, out_bytes}; if ((rc = mskf_copy_async(ctx, &cp, 1)) != MSKF_OK) return rc; }
    MSKF_HIPCHK(hipGetLastError());
    if ((rc = mskf_wait_event(ctx, &ctx->pend_trk.done, true)) != MSKF_OK) return rc;
    ctx->pend_trk.active = true; ctx->pend_trk.n = n; ctx->pend_trk.args = args;
    ctx->pend_trk.ts = ts;
    if (ctx->t_gate) ctx->host_s[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h0).count();
    return MSKF_OK;
}

// Wait for the batch started by mskf_fe_track_batch_begin and copy its results into the args given there (which, like
// their output arrays, must still be valid).  No-op when nothing is pending.
extern "C" int mskf_fe_track_batch_end(mskf_ctx *ctx) {
    if (!ctx) return MSKF_ERR_INVALID;
    if (!ctx->pend_trk.active) return MSKF_OK;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    int rc;
    ctx->pend_trk.active = false;      // also when the wait fails (timeout, stream error): the context must stay usable for a retry or a teardown
    if ((rc = mskf_wait_event(ctx, &ctx->pend_trk.done, false)) != MSKF_OK) return rc;
    const int n = ctx->pend_trk.n;
    const mskf_fe_track_args *args = ctx->pend_trk.args;
    const std::vector<size_t> &out_off = ctx->pend_trk.out_off;
    const int ts = ctx->pend_trk.ts;
    const auto t_h1 = std::chrono::steady_clock::now();
    long long tracks_t = 0, tracks_s = 0, pts = 0;
    for (int i = 0; i < n; ++i) {
        const mskf_fe_track_args &a = args[i];
        const size_t np = (size_t)a.n;
        if (!np) continue;
        const char *o = ctx->trk_out.h + out_off[i];
        std::memcpy(a.out0, o, sizeof(mskf_point2f) * np);
        std::memcpy(a.out1, o + sizeof(mskf_point2f) * np, sizeof(mskf_point2f) * np);
        std::memcpy(a.und0, o + 2 * sizeof(mskf_point2f) * np, sizeof(mskf_point2f) * np);
        std::memcpy(a.und1, o + 3 * sizeof(mskf_point2f) * np, sizeof(mskf_point2f) * np);
        std::memcpy(a.status, o + 4 * sizeof(mskf_point2f) * np, np);
        // units of an LK launch = point tracks it executed: temporal (n of the temporal streams), stereo (the points that
        // passed the temporal gate, or all n of a stereo-only stream)
        pts += (long long)np;
        if (a.do_temporal) { tracks_t += (long long)np; for (size_t k = 0; k < np; ++k) tracks_s += (a.status[k] & 1); }
        else tracks_s += (long long)np;
    }
    mskf_t_set_units(ctx, ts, MSKF_K_LK, tracks_t + tracks_s);      // point tracks of the launch: temporal + stereo
    (void)pts;
    mskf_t_collect(ctx);
    if (ctx->t_gate) ctx->host_s[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h1).count();
    return MSKF_OK;
}

extern "C" int mskf_fe_track(mskf_stream *s, const mskf_fe_track_args *args) {
    if (!s || !args) return MSKF_ERR_INVALID;
    mskf_stream *ss[1] = {s};
    return mskf_fe_track_batch(s->ctx, 1, ss, args);
}

extern "C" int mskf_fe_swap(mskf_stream *s) {
    if (!s) return MSKF_ERR_INVALID;
    std::swap(s->i_prev0, s->i_curr0);
    s->has_curr = false;
    return MSKF_OK;
}

// ------------------------------------------------------------------------------------------ a whole frame on the device
extern "C" int mskf_fe_grid_capacity(mskf_stream *s) { return s ? s->book.cap : 0; }

extern "C" int mskf_fe_set_grid(mskf_stream *s, int n, const uint64_t *id, const int32_t *lifetime, const mskf_point2f *cam0, const mskf_point2f *cam1,
                                const mskf_point2f *und0, const mskf_point2f *und1, uint64_t next_feature_id, const int32_t tracking_counters[3],
                                uint64_t ransac_draws) {
    if (!s || n < 0 || (n && (!id || !lifetime || !cam0 || !cam1 || !und0 || !und1))) return MSKF_ERR_INVALID;
    mskf_stream::Book &K = s->book;
    if (!K.cap) { mskf_set_error("this stream keeps its books on the host (grid_min / grid_max above the device limit)"); return MSKF_ERR_UNSUPPORTED; }
    if (n > K.cap) return MSKF_ERR_CAPACITY;
    MSKF_HIPCHK(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    MSKF_HIPCHK(hipStreamSynchronize(st));
    const FeGridArr &G = K.grid[K.parity];
    if (n) {
        MSKF_HIPCHK(hipMemcpyAsync(G.id, id, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        MSKF_HIPCHK(hipMemcpyAsync(G.lifetime, lifetime, 4 * (size_t)n, hipMemcpyHostToDevice, st));
        MSKF_HIPCHK(hipMemcpyAsync(G.cam0, cam0, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        MSKF_HIPCHK(hipMemcpyAsync(G.cam1, cam1, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        MSKF_HIPCHK(hipMemcpyAsync(G.und0, und0, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        MSKF_HIPCHK(hipMemcpyAsync(G.und1, und1, 8 * (size_t)n, hipMemcpyHostToDevice, st));
    }
    FeBookState h;
    std::memset(&h, 0, sizeof(h));
    h.next_id = next_feature_id; h.n_prev = n; h.n_curr = n;
    h.ransac_draws = ransac_draws;
    if (tracking_counters) { h.after_tracking = tracking_counters[0]; h.after_matching = tracking_counters[1]; h.after_ransac = tracking_counters[2]; }
    MSKF_HIPCHK(hipMemcpyAsync(K.st, &h, sizeof(h), hipMemcpyHostToDevice, st));
    MSKF_HIPCHK(hipStreamSynchronize(st));
    K.n_prev = n; K.n_cand_last = -1; K.grid_set = true;
    return MSKF_OK;
}

extern "C" int mskf_fe_frame_batch_begin(mskf_ctx *ctx, int n, mskf_stream *const *streams, const uint8_t *const *cam0, const uint8_t *const *cam1,
                                         int on_device, mskf_fe_frame_args *args) {
    if (!ctx || n <= 0 || !streams || !cam0 || !cam1 || !args) return MSKF_ERR_INVALID;
    if (ctx->pend_frame.active || ctx->pend_trk.active) { mskf_set_error("a batch of this context is still pending"); return MSKF_ERR_INVALID; }
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        if (!s || s->ctx != ctx) return MSKF_ERR_INVALID;
        if (!s->book.cap) { mskf_set_error("this stream keeps its books on the host (grid_min / grid_max above the device limit)"); return MSKF_ERR_UNSUPPORTED; }
        if (!s->book.grid_set) { mskf_set_error("no grid on the device yet: the first frame goes through mskf_fe_track + mskf_fe_set_grid"); return MSKF_ERR_INVALID; }
        const mskf_fe_frame_args &a = args[i];
        if (a.capacity < s->book.cap || !a.id || !a.lifetime || !a.cam0 || !a.cam1 || !a.und0 || !a.und1) return MSKF_ERR_INVALID;
    }
    const auto t_h0 = std::chrono::steady_clock::now();
    int rc = push_batch(ctx, n, streams, cam0, cam1, on_device, false);
    if (rc != MSKF_OK) return rc;
    if ((rc = ctx->desc[1].ensure(n)) != MSKF_OK || (rc = ctx->desc[2].ensure(n)) != MSKF_OK || (rc = ctx->book_desc.ensure(n)) != MSKF_OK) return rc;
    std::vector<size_t> &out_off = ctx->pend_frame.out_off;
    out_off.resize(n);
    size_t out_bytes = 0, scratch_bytes = 0;
    int max_prev = 0, max_cand_est = 0;
    for (int i = 0; i < n; ++i) {
        const mskf_stream::Book &K = streams[i]->book;
        out_off[i] = out_bytes;
        out_bytes += (64 + 44 * (size_t)K.cap + 255) & ~(size_t)255;
        scratch_bytes = std::max(scratch_bytes, 4 * fe_book_scratch_ints(K.cap, K.cand_cap, K.det_cap, K.n_codes, K.det_cap));
        max_prev = std::max(max_prev, K.n_prev);
        // candidates are counted on the device: the launch is sized from the last frame's count, a block takes several point
        // groups if there are more this time
        const int est = K.n_cand_last < 0 ? K.cand_cap / 2 : std::min(K.cand_cap, K.n_cand_last + K.n_cand_last / 2 + 32);
        max_cand_est = std::max(max_cand_est, est);
    }
    if (out_bytes > ctx->book_out.cap) {
        MSKF_HIPCHK(hipStreamSynchronize(st));
        if ((rc = ctx->book_out.ensure(out_bytes)) != MSKF_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        mskf_stream::Book &K = s->book;
        const mskf_fe_cfg &fe = s->fe;
        // first track call: the previous grid's cam0 points, temporal + stereo
        FeStreamDev &d1 = ctx->desc[1].h[i];
        fill_fe_desc(s, d1);
        d1.n_pts = K.n_prev; d1.n_pts_dev = nullptr; d1.do_temporal = 1;
        std::memcpy(d1.Hpred, args[i].Hpred, sizeof(d1.Hpred));
        d1.in_pts = K.grid[K.parity].cam0;
        d1.out0 = K.t_out0; d1.out1 = K.t_out1; d1.und0 = K.t_und0; d1.und1 = K.t_und1; d1.status = K.t_status;
        // second track call: the candidates fe_book1 leaves in the stream's list, stereo only
        FeStreamDev &d2 = ctx->desc[2].h[i];
        fill_fe_desc(s, d2);
        d2.n_pts = 0; d2.n_pts_dev = &K.st->n_cand; d2.do_temporal = 0;
        d2.Hpred[0] = d2.Hpred[4] = d2.Hpred[8] = 1.0;
        d2.in_pts = K.cand_pt;
        d2.out0 = K.c_out0; d2.out1 = K.c_out1; d2.und0 = K.c_und0; d2.und1 = K.c_und1; d2.status = K.c_status;
        FeBookDev &B = ctx->book_desc.h[i];
        std::memset(&B, 0, sizeof(B));
        B.grid_row = fe.grid_row; B.grid_col = fe.grid_col; B.grid_min = fe.grid_min_feature_num; B.grid_max = fe.grid_max_feature_num;
        B.n_codes = K.n_codes; B.n_cells = K.n_cells; B.grid_w = K.grid_w; B.grid_h = K.grid_h;
        B.det_rows = fe.det_rows; B.det_cols = fe.det_cols; B.det_cw = s->det_cw; B.det_ch = s->det_ch;
        B.thr_score = fe.fast_threshold * 256;
        B.q4 = (fe.compat_flags & MSKF_COMPAT_Q4_RESPONSE_INDEX) ? 1 : 0;
        // twoPointRansac between the tracks (:482-500; commented out in the reference, Q5): iterations as :920-921
        B.ransac = (fe.compat_flags & MSKF_COMPAT_Q5_NO_RANSAC) ? 0 : 1;
        B.ransac_iters = static_cast<int>(std::ceil(std::log(1 - 0.99) / std::log(1 - 0.7 * 0.7)));
        B.ransac_thr = fe.ransac_threshold;
        B.ransac_npu[0] = 2.0 / (s->cam0.K[0] + s->cam0.K[1]); B.ransac_npu[1] = 2.0 / (s->cam1.K[0] + s->cam1.K[1]);
        std::memcpy(B.R_p_c, args[i].R_p_c, sizeof(B.R_p_c));
        B.rs_pair = K.rs_pair; B.rs_pt = K.rs_pt; B.rs_scalar = K.rs_scalar;
        B.cap = K.cap; B.cand_cap = K.cand_cap; B.det_cap = K.det_cap;
        B.gen = (unsigned int)((s->push_gen - 1) % 255ULL) + 1U;
        B.st = K.st;
        B.prev = K.grid[K.parity]; B.curr = K.grid[K.parity ^ 1]; B.tracked = K.grid[2];
        B.t_out0 = K.t_out0; B.t_out1 = K.t_out1; B.t_und0 = K.t_und0; B.t_und1 = K.t_und1; B.t_status = K.t_status;
        B.cell_keys = (const unsigned long long *)(ctx->cell_arena.d + s->cell_off);
        B.det_pt = K.det_pt; B.det_score = K.det_score;
        B.cand_pt = K.cand_pt; B.cand_index = K.cand_index; B.cand_score = K.cand_score; B.cand_off = K.cand_off; B.cand_cnt = K.cand_cnt;
        B.c_out0 = K.c_out0; B.c_out1 = K.c_out1; B.c_und0 = K.c_und0; B.c_und1 = K.c_und1; B.c_status = K.c_status;
        B.cell_count = K.cell_count;
        char *o = ctx->book_out.d + out_off[i];
        B.x_info = (int *)o;
        B.x_id = (unsigned long long *)(o + 64);
        B.x_lifetime = (int *)(o + 64 + 8 * (size_t)K.cap);
        B.x_cam0 = (mskf_point2f *)(o + 64 + 12 * (size_t)K.cap);
        B.x_cam1 = B.x_cam0 + K.cap; B.x_und0 = B.x_cam1 + K.cap; B.x_und1 = B.x_und0 + K.cap;
    }
    {
        const MskfCopy cp[3] = {{ctx->desc[1].d, ctx->desc[1].h, sizeof(FeStreamDev) * (size_t)n}, {ctx->desc[2].d, ctx->desc[2].h, sizeof(FeStreamDev) * (size_t)n},
                                {ctx->book_desc.d, ctx->book_desc.h, sizeof(FeBookDev) * (size_t)n}};
        if ((rc = mskf_copy_async(ctx, cp, 3)) != MSKF_OK) return rc;
    }
    const int ts1 = mskf_t_begin(ctx, MSKF_K_LK);
    fe_launch_track(ctx->desc[1].d, n, max_prev, st);
    mskf_t_end(ctx, ts1, 0);
    int tb = mskf_t_begin(ctx, MSKF_K_FE_BOOK);
    fe_launch_book(ctx->book_desc.d, n, 0, scratch_bytes, st);
    mskf_t_end(ctx, tb, n);
    const int ts2 = mskf_t_begin(ctx, MSKF_K_LK);
    fe_launch_track(ctx->desc[2].d, n, std::max(max_cand_est, 4), st);
    mskf_t_end(ctx, ts2, 0);
    tb = mskf_t_begin(ctx, MSKF_K_FE_BOOK);
    fe_launch_book(ctx->book_desc.d, n, 1, scratch_bytes, st);
    mskf_t_end(ctx, tb, n);
    { const MskfCopy cp = {ctx->book_out.h, ctx->book_out.d, out_bytes}; if ((rc = mskf_copy_async(ctx, &cp, 1)) != MSKF_OK) return rc; }
    MSKF_HIPCHK(hipGetLastError());
    if ((rc = mskf_wait_event(ctx, &ctx->pend_frame.done, true)) != MSKF_OK) return rc;
    // state rotation (:192-200): the grid just built is the next frame's previous grid, curr cam0 becomes prev cam0
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams[i];
        s->book.parity ^= 1;
        std::swap(s->i_prev0, s->i_curr0);
        s->has_curr = false;
    }
    ctx->pend_frame.active = true; ctx->pend_frame.n = n; ctx->pend_frame.streams = streams; ctx->pend_frame.args = args;
    ctx->pend_frame.ts1 = ts1; ctx->pend_frame.ts2 = ts2;
    if (ctx->t_gate) ctx->host_s[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h0).count();
    return MSKF_OK;
}

extern "C" int mskf_fe_frame_batch_end(mskf_ctx *ctx) {
    if (!ctx) return MSKF_ERR_INVALID;
    mskf_ctx::PendingFrame &F = ctx->pend_frame;
    if (!F.active) return MSKF_OK;
    MSKF_HIPCHK(hipSetDevice(ctx->device));
    F.active = false;
    int rc;
    if ((rc = mskf_wait_event(ctx, &F.done, false)) != MSKF_OK) return rc;
    const auto t_h1 = std::chrono::steady_clock::now();
    long long tracks1 = 0, tracks2 = 0;
    bool overflow = false;
    for (int i = 0; i < F.n; ++i) {
        mskf_stream *s = F.streams[i];
        mskf_stream::Book &K = s->book;
        mskf_fe_frame_args &a = F.args[i];
        const char *o = ctx->book_out.h + F.out_off[i];
        const int *info = (const int *)o;
        const int m = info[0];
        if (m < 0 || m > K.cap || info[8]) { overflow = true; continue; }
        a.n = m; a.n_candidates = info[1];
        a.before_tracking = info[2]; a.after_tracking = info[3]; a.after_matching = info[4]; a.after_ransac = info[5];
        a.next_feature_id = (uint64_t)(unsigned int)info[6] | ((uint64_t)(unsigned int)info[7] << 32);
        a.n_new = info[9];
        a.ransac_draws = (uint64_t)(unsigned int)info[12] | ((uint64_t)(unsigned int)info[13] << 32);
        std::memcpy(a.id, o + 64, 8 * (size_t)m);
        std::memcpy(a.lifetime, o + 64 + 8 * (size_t)K.cap, 4 * (size_t)m);
        const char *p = o + 64 + 12 * (size_t)K.cap;
        std::memcpy(a.cam0, p, 8 * (size_t)m);
        std::memcpy(a.cam1, p + 8 * (size_t)K.cap, 8 * (size_t)m);
        std::memcpy(a.und0, p + 16 * (size_t)K.cap, 8 * (size_t)m);
        std::memcpy(a.und1, p + 24 * (size_t)K.cap, 8 * (size_t)m);
        tracks1 += (long long)a.before_tracking + (a.before_tracking > 0 ? a.after_tracking : 0);
        tracks2 += a.n_candidates;
        K.n_prev = m; K.n_cand_last = a.n_candidates;
    }
    mskf_t_set_units(ctx, F.ts1, MSKF_K_LK, tracks1);
    mskf_t_set_units(ctx, F.ts2, MSKF_K_LK, tracks2);
    mskf_t_collect(ctx);
    if (ctx->t_gate) ctx->host_s[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h1).count();
    if (overflow) { mskf_set_error("device bookkeeping reported a capacity overflow"); return MSKF_ERR_CAPACITY; }
    return MSKF_OK;
}

extern "C" int mskf_fe_get_level(mskf_stream *s, int role, int level, uint8_t *out, int capacity, int *w, int *h) {
    if (!s || !out || role < 0 || role > 2 || level < 0 || level >= MSKF_LEVELS) return MSKF_ERR_INVALID;
    const int idx = role == 0 ? s->i_prev0 : (role == 1 ? s->i_curr0 : s->i_curr1);
    const size_t bytes = (size_t)s->lw[level] * s->lh[level];
    if ((size_t)capacity < bytes) return MSKF_ERR_CAPACITY;
    MSKF_HIPCHK(hipSetDevice(s->ctx->device));
    MSKF_HIPCHK(hipStreamSynchronize(s->ctx->stream));
    const uint8_t *src = (level == 0 && s->lvl0[idx]) ? s->lvl0[idx] : s->pyr[idx] + s->lvl_off[level];
    MSKF_HIPCHK(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    if (w) *w = s->lw[level];
    if (h) *h = s->lh[level];
    return MSKF_OK;
}
