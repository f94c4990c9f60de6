// mskf_internal.h — private definitions of the C-ABI handles (mskf_ctx / mskf_stream).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../../include/mskf_hip.h"
#include "../hip/fe_device.h"
#include "../hip/fe_book.h"
#include "../hip/ekf_device.h"
#include "host_math.h"

struct Pyr3Job { const uint8_t *src; uint8_t *d1, *d2, *d3; int w0, h0; };      // one image: level 0 in, levels 1..3 out (fe_kernels.hip)

extern "C" {
void fe_launch_pyr_down3(const Pyr3Job *jobs_dev, int n_jobs, int max_w0, int max_h0, hipStream_t st);
// Host wait for everything queued on the context's stream.  MSKF_WAIT=block (default when the process runs
// more waiting host threads than it has cores to spin on) parks the thread on an interrupt-driven event instead of
// spinning in hipStreamSynchronize, leaving the core to the other groups' host phases.
int mskf_wait(mskf_ctx *c);
void fe_launch_detect(const FeStreamDev *streams_dev, int n_streams, int max_w, int max_h, unsigned int gen, hipStream_t st);
void fe_launch_track(const FeStreamDev *streams_dev, int n_streams, int max_pts, hipStream_t st);
void fe_launch_book(const FeBookDev *books_dev, int n_streams, int which, size_t scratch_bytes, hipStream_t st);
}

void mskf_set_error(const std::string &s);

#define MSKF_HIPCHK(expr)                                                                            \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            mskf_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
            return MSKF_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

template <typename T>
struct PinnedDev {  // a pinned host array with a device twin
    T *h = nullptr, *d = nullptr;
    size_t cap = 0;
    std::vector<std::pair<T *, T *>> retired;      // outgrown buffers, freed with the owner
    // Growth NEVER frees: hipFree / hipHostFree wait for every stream of the device, and a context that outgrew a staging
    // buffer in the middle of a run stood still until all the other groups' queues were idle (0.3 s in the round-3 bench).
    // The outgrown pair is retired and freed by release(); with doubling that is at most as much again as the final size.
    int ensure(size_t n) {
        if (n <= cap) return MSKF_OK;
        if (h || d) retired.push_back({h, d});
        h = d = nullptr; cap = 0;
        size_t c = n < 16 ? 16 : 2 * n;
        MSKF_HIPCHK(hipHostMalloc((void **)&h, c * sizeof(T), hipHostMallocDefault));
        MSKF_HIPCHK(hipMalloc((void **)&d, c * sizeof(T)));
        cap = c;
        return MSKF_OK;
    }
    void release() {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        for (auto &r : retired) { if (r.first) (void)hipHostFree(r.first); if (r.second) (void)hipFree(r.second); }
        retired.clear();
        h = d = nullptr; cap = 0;
    }
};

struct TimingSlot { hipEvent_t a, b; int kind; long long units; };

struct mskf_ctx {
    bool timing = false;
    bool t_gate = true;                          // mskf_ctx_timing_gate: launches begun (and host seconds spent) while it is off are not accounted
    int timing_period = 1;                       // every n-th launch of a kind is timed (mskf_ctx_set_timing)
    long long t_all[MSKF_K_COUNT] = {0};         // launches of a kind since the last reset, timed or not
    double host_s[4] = {0, 0, 0, 0};   // host seconds inside the batched entry points: [0] update pack, [1] update unpack, [2] track pack, [3] track unpack
    std::vector<TimingSlot> t_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> t_pool;
    double t_ms[MSKF_K_COUNT] = {0};
    long long t_launches[MSKF_K_COUNT] = {0}, t_units[MSKF_K_COUNT] = {0};
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;          // false: created by mskf_ctx_create_shared on another context's stream
    PinnedDev<FeStreamDev> desc[3];   // 0: push/detect, 1: track (first track call of a device frame), 2: second track call of a device frame
    PinnedDev<FeBookDev> book_desc;   // bookkeeping descriptors of a device frame batch
    PinnedDev<char> book_out;         // what a device frame batch returns: per stream 16 ints + the published grid
    struct PendingFrame {
        bool active = false; int n = 0; mskf_stream *const *streams = nullptr; struct mskf_fe_frame_args *args = nullptr;
        std::vector<size_t> out_off; int ts1 = -1, ts2 = -1;
        hipEvent_t done = nullptr;
    } pend_frame;
    hipEvent_t wait_ev = nullptr;     // mark of mskf_wait
    bool wait_block = false;
    volatile unsigned int *flag_h = nullptr;   // pinned host words the mark kernels write (spinning mode), one per mark slot
    unsigned int flag_seq[8] = {0};
    hipEvent_t *flag_slot[8] = {nullptr};
    PinnedDev<char> cell_arena;       // per-cell maximum keys of every stream of the last push batch (one D2H copy)
    PinnedDev<char> trk_in, trk_out;  // input points / results of every stream of a track batch (one copy each way)
    unsigned long long push_gen = 0;
    hipEvent_t cell_ev = nullptr;     // recorded behind the D2H copy of the per-cell maxima of the last push
    bool cell_keys_dirty = true;      // the key array holds bytes no generation tag explains (fresh allocation): clear before use
    PinnedDev<Pyr3Job> jobs;
    PinnedDev<EkfStreamDev> ekf_desc;
    PinnedDev<char> upd_in, upd_out;     // inputs / results of every stream of an update batch (one copy each way)
    PinnedDev<char> pred_arena;          // descriptors + IMU steps + J of mskf_ekf_predict_batch
    hipEvent_t pred_done = nullptr;
    bool pred_pending = false;
    std::vector<mskf_stream *> streams;
    // a batch between its *_begin and *_end call (one of each kind per context)
    struct PendingTrack {
        bool active = false; int n = 0; const mskf_fe_track_args *args = nullptr;
        std::vector<size_t> out_off; int ts = -1;
        hipEvent_t done = nullptr;
    } pend_trk;
    struct PendingUpdate {
        bool active = false, launched = false; int n = 0; mskf_stream *const *streams = nullptr; mskf_ekf_update_args *args = nullptr;
        std::vector<size_t> lay;   // per stream: o_dx, o_gamma, o_rows, o_status, o_pos
        std::vector<int> cnt_cls;              // scratch: features per size class of the feature kernel and stream, [3][n]
        hipEvent_t done = nullptr;
    } pend_upd;
    struct PendingPosVar { bool active = false; int n = 0; double *out = nullptr; size_t desc_bytes = 0; hipEvent_t done = nullptr; } pend_pv;
};
// Completion marks.  mskf_wait_event(c, slot, true) marks the point the context's stream has reached, (.., false) waits
// for that mark.  Spinning mode (default): the mark is a sequence number written into pinned host memory by a stream
// write-value command (or a one-thread kernel), the wait spins on that word in user space (no HIP call inside the wait: hipEventSynchronize / hipStreamSynchronize
// spinning in several threads at once slows every other thread's launches down, measured -25 %).  MSKF_WAIT=block: a
// blocking-sync HIP event, the thread is parked.  `slot` identifies the mark (one per kind of pending batch).
// Staging copies between the library's own pinned host arenas and device memory, enqueued on the context's stream as ONE kernel
// launch for up to MSKF_COPY_SEGS segments (fe_kernels.hip: why not hipMemcpyAsync); MSKF_SDMA_COPIES=1 goes back to
// hipMemcpyAsync (one call per segment).  Both ends must be addressable from a kernel: device memory or hipHostMalloc'ed memory.
struct MskfCopy { void *dst; const void *src; size_t bytes; };
int mskf_copy_async(mskf_ctx *c, const MskfCopy *segs, int n);
int mskf_wait_event(mskf_ctx *c, hipEvent_t *ev_slot, bool record);

struct mskf_stream {
    mskf_ctx *ctx = nullptr;
    mskf_ctx *ctx_ekf = nullptr;      // context the mskf_ekf_* calls run on (== ctx unless re-attached)
    mskf_ctx *home_ctx = nullptr;     // the context the stream was created on (its bookkeeping list); ctx / ctx_ekf may move (mskf_stream_rebind)
    mskf_calib calib;
    mskf_fe_cfg fe;
    mskf_ekf_cfg ekf;
    // ---- front-end
    int w = 0, h = 0;
    int lw[MSKF_LEVELS], lh[MSKF_LEVELS];
    size_t lvl_off[MSKF_LEVELS];
    size_t pyr_bytes = 0;
    uint8_t *pyr[3] = {nullptr, nullptr, nullptr};
    const uint8_t *lvl0[3] = {nullptr, nullptr, nullptr};   // level 0 of each pyramid: own buffer or a borrowed device image
    int i_prev0 = 0, i_curr0 = 1, i_curr1 = 2;
    bool has_curr = false;
    int pt_cap = 0;
    size_t cell_off = 0;              // slice of ctx->cell_arena
    unsigned long long push_gen = 0;
    CamDev cam0, cam1;
    double R01[9], E[9], epi_thresh = 0;
    int det_cw = 0, det_ch = 0;
    int det_floor = 0;                // mskf_fe_set_detect_floor
    double time_stamp = 0;
    // ---- device-side bookkeeping (fe_book.h): grids, candidate lists and track results of the stream, one allocation
    struct Book {
        char *mem = nullptr;
        int cap = 0, cand_cap = 0, det_cap = 0, n_codes = 0, n_cells = 0, grid_w = 0, grid_h = 0;
        FeBookState *st = nullptr;
        FeGridArr grid[3];                         // [parity], [parity ^ 1]: previous / current grid; [2]: this frame's survivors
        mskf_point2f *det_pt = nullptr; int *det_score = nullptr;
        mskf_point2f *cand_pt = nullptr; int *cand_index = nullptr, *cand_score = nullptr, *cand_off = nullptr, *cand_cnt = nullptr, *cell_count = nullptr;
        mskf_point2f *t_out0 = nullptr, *t_out1 = nullptr, *t_und0 = nullptr, *t_und1 = nullptr; uint8_t *t_status = nullptr;
        mskf_point2f *c_out0 = nullptr, *c_out1 = nullptr, *c_und0 = nullptr, *c_und1 = nullptr; uint8_t *c_status = nullptr;
        double *rs_pair = nullptr, *rs_scalar = nullptr; float *rs_pt = nullptr;     // scratch of the 2-point RANSAC (fe_book.h)
        int parity = 0;                            // grid[parity] holds the published grid of the last frame
        int n_prev = 0, n_cand_last = -1;          // host copies of the counts (launch sizing)
        bool grid_set = false;
    } book;
    // ---- EKF
    EkfStreamState ekf_state;
    void *ekf_extra = nullptr;
};

// timing helpers: t_begin records the start event and returns a slot index (or -1), t_end records the stop event
int mskf_t_begin(mskf_ctx *c, int kind);
void mskf_t_end(mskf_ctx *c, int slot, long long units);
void mskf_t_collect(mskf_ctx *c);   // call after the stream has been synchronised
void mskf_t_set_units(mskf_ctx *c, int slot, int kind, long long units);   // units of a slot begun earlier (bounds- and kind-checked)
void fill_pyr(const mskf_stream *s, int idx, PyrDev &p);
int mskf_ekf_stream_init(mskf_stream *s);
void mskf_ekf_stream_free(mskf_stream *s);
