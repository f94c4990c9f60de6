// batch_runner.h — steps many independent VIO streams (cg::System objects) in lockstep so that every
// device phase of a frame is ONE batched C-ABI call (mskf_fe_push_stereo_batch, mskf_fe_track_batch,
// mskf_ekf_update_batch, ...).  Streams are independent units: no data crosses streams.
// A BatchGroup owns one mskf_ctx (one HIP stream); a MultiRunner runs several groups on their own host
// threads so the host bookkeeping of one group overlaps the kernels of another.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "system.h"

namespace cg {

struct StreamSequence {   // a looping pre-rendered stereo sequence + IMU samples for one stream
    const uint8_t *cam0_base = nullptr, *cam1_base = nullptr;   // frame key k at base + k * frame_bytes
    int on_device = 1;
    size_t frame_bytes = 0;
    int n_static = 0, n_loop = 1;
    long long t0_ns = 0, frame_dt_ns = 50000000LL;
    const mskf_imu_sample *imu = nullptr;   // sample j, j in [0, n_imu)
    int n_imu = 0;
    int imu_cursor = 0;       // next sample for the front-end (and for the filter in lockstep mode)
    int imu_cursor_ekf = 0;   // next sample for the filter stage of the pipeline
};

// A measurement window inside one continuous pipelined run of all groups (bench.py).  The groups' hardware queues are not
// served evenly (some groups are a quarter of the run ahead of others), so the window is defined on the WORK, not on any one
// group's frames: it opens when the groups together have completed n_groups x W frames (front-end and filter) and closes
// when they have completed n_groups x (W + K) — exactly K steps' worth of stream-frames are finished inside it, with every
// group busy from before it opens until after it closes (a group runs at least W + K frames and then keeps stepping until
// the window is closed; frames started before it closes are finished but not counted).  Each stage opens its accounting
// gates (kernel timing of its context, host bookkeeping slots, phase times) at its first frame boundary inside the window
// and closes them at the first one after it.
struct TimedShared {
    std::atomic<long> completed{0};
    long target_open = 0, target_close = 0;
    std::atomic<int> phase{0};               // 0 warm-up, 1 window open, 2 closed
    double t_open = 0, t_close = 0;          // steady_clock seconds
};
struct TimedWindow {
    TimedShared *shared = nullptr;
    int mark_end = 0;                        // absolute frame index: the sentinel snapshot is taken after frame mark_end - 1 of this group
    int max_extra = 0;
    // results
    double t_fe_begin = 0, t_fe_end = 0, t_ekf_begin = 0, t_ekf_end = 0;   // when the stages opened / closed their gates
    int fe_frames = 0, ekf_frames = 0;       // frames each stage started between its open and close
    int frames_done = 0;                     // frames the group processed in this run
    int frames_at_close = 0;                 // ... of which completed (both stages) when the shared window closed (balanced runner)
};

struct FrameBatch {           // what the front-end stage hands to the filter stage: one frame of every stream
    int frame = 0;
    std::vector<std::shared_ptr<CameraMeasurement>> msg;   // snapshot: live + stale entries + first tail record
    std::vector<size_t> tail_start, total;
};

// Fork-join helper: the per-stream host phases of a group are independent, so they are split over a few
// persistent worker threads (the calling thread takes a share too).
class ForkJoin {
  public:
    explicit ForkJoin(int n_threads);
    ~ForkJoin();
    void run(int n, const std::function<void(int)> &fn);   // fn(i) for i in [0, n), returns when all are done
    bool prof_on_ = true;      // the caller's host-profile gate, adopted by the workers for the batch
  private:
    void worker(int id);
    int nt_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0;
    unsigned long long gen_ = 0;
    std::atomic<int> next_{0}, done_{0};
    bool stop_ = false;
};

class BatchGroup {
  public:
    // host_threads / ekf_host_threads: threads that share the per-stream host phases of the front-end / filter stage (0 = as
    // host_threads); halves = 2: two staggered half-batches per stage on contexts sharing the stage's HIP stream
    BatchGroup(int device, int n, const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, int host_threads = 1, int ekf_host_threads = 0, int halves = 1);
    ~BatchGroup();
    bool ok() const { return ok_; }
    int size() const { return (int)systems_.size(); }
    System &system(int i) { return *systems_[i]; }
    void imu(int i, const mskf_imu_sample &s);
    // one frame of every stream: front-end + back-end
    int step(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t, bool is_draw);
    // frames [first, first+n) of the attached sequences, IMU fed in the reference harness order (Q10)
    int run(int first, int n);
    // same frames as a two-stage pipeline: a front-end thread (context ctx()) and a filter thread (context
    // ekf_ctx()); the front-end never reads filter state, so the results are identical to run()
    int run_pipelined(int first, int n, TimedWindow *win = nullptr);
    // ---- the two stages of a frame as separate calls on BORROWED contexts, for the balanced runner (MultiRunner::run_balanced):
    // a batch (this object's streams and host state) is handed to whichever worker (a context + a host thread) is free.
    // fe_stage: IMU feed + front-end of frame k + the hand-off snapshot; ekf_stage: IMU feed + filter of a snapshot.  `acc`
    // receives the phase times (the worker's accounting, PH_*).  A stage of a batch is run by one worker at a time.
    int fe_stage(mskf_ctx *ctx, int k, double *acc, std::unique_ptr<FrameBatch> &out);
    int ekf_stage(mskf_ctx *ctx, FrameBatch *fb, double *acc);
    void rebind_home();                       // streams back on the group's own contexts (after a balanced run)
    void fill_handoff(int k, std::unique_ptr<FrameBatch> &fb);
    void snapshot_fe_mark();                  // mark_dump: front-end half / filter half of local stream 0
    void snapshot_ekf_mark();
    std::deque<std::unique_ptr<FrameBatch>> handoff;          // balanced runner: frames through the front-end, waiting for the filter
    std::vector<std::unique_ptr<FrameBatch>> handoff_pool;
    // device contexts: the streams of a group are driven as up to two half-batches, each with its own staging
    // context; the halves of a stage share one HIP stream (mskf_ctx_create_shared), so a group still uses two queues
    int n_halves() const { return (int)half_.size(); }
    mskf_ctx *ctx(int h = 0) const { return half_[h].ctx; }
    mskf_ctx *ekf_ctx(int h = 0) const { return half_[h].ctx_ekf; }
    std::vector<StreamSequence> seq;
    const std::string &error() const { return error_; }
    // phases of the front-end thread: PH_IMU .. PH_FE_QWAIT (without PH_EKF_*); of the filter thread: PH_EKF_QWAIT, PH_IMU_EKF, PH_EKF_A .. PH_POSVAR
    enum { PH_PUSH = 0, PH_PREP1, PH_TRACK1, PH_AFTER1, PH_TRACK2, PH_AFTER2, PH_EKF_A, PH_UPD1, PH_EKF_B, PH_UPD2, PH_EKF_C, PH_POSVAR, PH_IMU,
           PH_HANDOFF, PH_FE_QWAIT, PH_EKF_QWAIT, PH_IMU_EKF, PH_COUNT };
    double phase_s[PH_COUNT] = {0};   // wall seconds per phase of step() (host bookkeeping vs device calls)
    double window_phase_s[PH_COUNT] = {0};   // the same inside the last TimedWindow (each stage between its own marks)
    // what local stream 0 had computed when the stages closed the window (the sentinel of bench.py): front-end state after
    // frame mark_end - 1, filter state after the same frame
    struct MarkDump { std::vector<unsigned long long> ids; std::vector<int> life; std::vector<Point2f> c0, c1; double imu[28] = {0}; bool fe_valid = false, ekf_valid = false; } mark_dump;
    void set_gates(bool on);          // accounting gates of both contexts (call while no stage is running)

  private:
    int step_fe(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t, bool is_draw);
    int step_ekf(const FrameBatch *fb);
    int feed_imu(int k, bool to_fe, bool to_ekf);
    // A half-batch: streams [i0, i0 + n).  While the device works on one half the host thread prepares or digests the
    // other (the *_begin / *_end entry points of the C-ABI), so host bookkeeping and device time overlap inside a thread.
    struct Half {
        mskf_ctx *ctx = nullptr, *ctx_ekf = nullptr;
        int i0 = 0, n = 0;
        std::vector<mskf_stream *> sub_s;              // streams with a non-empty update, and their args (valid until *_end)
        std::vector<mskf_ekf_update_args> sub_a;
        std::vector<int> sub_i;
        std::vector<int32_t> ns, rm;
        std::vector<const mskf_imu_step *> sp;
        std::vector<const double *> jp;
        std::vector<double> pv;
        bool any = false, upd_pending = false, pv_pending = false;
    };
    std::vector<Half> half_;
    mskf_ctx *home_fe_ = nullptr, *home_ekf_ = nullptr;       // the contexts this group created (half_[0] may point at borrowed ones)
    double *acc_fe_ = nullptr, *acc_ekf_ = nullptr;           // where step_fe / step_ekf account their phases (default: phase_s)
    mskf_point *ekf_tail_ = nullptr;                          // the filter stage's last enqueued work (clone removal is not waited for)
    mskf_ctx *ekf_tail_ctx_ = nullptr;
    bool ok_ = false;
    std::string error_;
    std::vector<std::unique_ptr<System>> systems_;
    std::vector<mskf_stream *> streams_;
    std::vector<mskf_fe_track_args> a1_, a2_;
    std::vector<mskf_fe_frame_args> fa_;
    std::vector<mskf_ekf_update_args> u_;
    std::vector<const uint8_t *> p0_, p1_;
    std::vector<double> t_;
    std::unique_ptr<ForkJoin> pool_, pool_ekf_;
    void par(int n, const std::function<void(int)> &fn) { if (pool_) pool_->run(n, fn); else for (int i = 0; i < n; ++i) fn(i); }
};

class MultiRunner {
  public:
    MultiRunner(int device, int n_groups, int per_group, const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf,
                int host_threads = 1, int ekf_host_threads = 0, int halves = 1);
    ~MultiRunner();
    bool ok() const;
    int n_streams() const { return n_groups_ * per_group_; }
    int n_groups() const { return n_groups_; }
    BatchGroup &group(int g) { return *groups_[g]; }
    BatchGroup &group_of(int stream, int &local) { local = stream % per_group_; return *groups_[stream / per_group_]; }
    System &system(int stream) { int l; BatchGroup &g = group_of(stream, l); return g.system(l); }
    StreamSequence &sequence(int stream) { int l; BatchGroup &g = group_of(stream, l); return g.seq[l]; }
    void imu(int stream, const mskf_imu_sample &s) { int l; BatchGroup &g = group_of(stream, l); g.imu(l, s); }
    int step(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t);
    int run(int first, int n, bool threaded, bool pipelined = false);
    // ONE pipelined run of every group (no fill / drain at the warm-up / timed boundary): *elapsed_s = the time in which the
    // groups together completed frames n_groups x warmup + 1 ... n_groups x (warmup + steps) of the run (TimedShared).
    int run_timed(int first, int warmup, int steps, int max_extra, double *elapsed_s);
    // The same measurement with the batches NOT tied to queues: every group contributes a front-end worker and a filter worker
    // (its two contexts and two threads), and a worker takes, frame by frame, the batch that is furthest behind and ready for its
    // stage.  The hardware queues of a device are not served evenly, and which ones fall behind changes from run to run (measured:
    // 78 k to 99 k stereo frames/s for the same code with fixed batch-to-queue binding, some groups at 10 ms per frame and others
    // at 26); a slow queue then simply takes fewer batches, all streams advance at the same pace, and the device stays loaded.
    // Results are identical (a stream's arithmetic does not depend on the queue it runs on).
    int run_balanced(int first, int warmup, int steps, int max_extra, double *elapsed_s, bool plain = false);
    // workers of the balanced runner: front-end / filter workers for the n_groups batches (0 = one per batch)
    void set_workers(int fe_workers, int ekf_workers) { fe_workers_ = fe_workers; ekf_workers_ = ekf_workers; }
    int frames_done(int g) const { return next_[g]; }          // next frame index of group g (absolute)
    const TimedWindow &window(int g) const { return win_[g]; }
    // group g works `g * delta` frames ahead of the frame index passed to run(): replicas of one looping sequence in
    // different groups then never read the same stereo pair at the same time (no cache sharing across groups).  The
    // first run() after the call lets every group catch up through its own offset.
    void set_stagger(int delta) { for (int g = 0; g < n_groups_; ++g) off_[g] = g * delta; }
    int group_offset(int g) const { return off_[g]; }
    std::string error() const;

  private:
    int n_groups_, per_group_;
    std::vector<std::unique_ptr<BatchGroup>> groups_;
    int fe_workers_ = 0, ekf_workers_ = 0;
    std::vector<int> off_, next_;   // per group: frame offset, next frame not yet processed
    std::vector<TimedWindow> win_;
};

}  // namespace cg
