// batch_runner.cpp — see batch_runner.h.
#include "batch_runner.h"
#include "host_prof.h"
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <deque>

namespace cg {

ForkJoin::ForkJoin(int n_threads) : nt_(n_threads) {
    for (int i = 1; i < nt_; ++i) th_.emplace_back([this, i]() { worker(i); });
}
ForkJoin::~ForkJoin() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
}
void ForkJoin::worker(int) {
    unsigned long long seen = 0;
    for (;;) {
        const std::function<void(int)> *fn;
        int n;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&]() { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            fn = fn_; n = n_;
            hostprof::enabled() = prof_on_;
        }
        for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*fn)(i);
        done_.fetch_add(1);
    }
}
void ForkJoin::run(int n, const std::function<void(int)> &fn) {
    if (nt_ <= 1 || n <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    next_.store(0); done_.store(0);
    { std::lock_guard<std::mutex> lk(mu_); fn_ = &fn; n_ = n; prof_on_ = hostprof::enabled(); ++gen_; }
    cv_.notify_all();
    for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) fn(i);
    while (done_.load() < nt_ - 1) std::this_thread::yield();
}

BatchGroup::BatchGroup(int device, int n, const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, int host_threads, int ekf_host_threads, int halves) {
    // per-stream host phases of a group are independent: optional helper threads for the front-end / filter stages
    const int ht_fe = std::max(1, host_threads), ht_ekf = std::max(1, ekf_host_threads > 0 ? ekf_host_threads : host_threads);
    if (ht_fe > 1) pool_.reset(new ForkJoin(ht_fe));
    if (ht_ekf > 1) pool_ekf_.reset(new ForkJoin(ht_ekf));
    // one batch per group and stage; halves = 2 splits it into two staggered half-batches (one half's host phase under the
    // other's kernels) for a single-group deployment.  With several groups the other groups already fill those gaps and
    // twice the launches of half the size cost more than the overlap returns (54.5 k vs 70.6 k stereo frames/s, round 2).
    int nh = std::max(1, std::min(2, halves));
    if (n < 2) nh = 1;
    half_.resize(nh);
    int rc = mskf_ctx_create(device, &half_[0].ctx);
    if (rc == MSKF_OK) rc = mskf_ctx_create_prio(device, 1, &half_[0].ctx_ekf);     // the filter is the serial chain of a frame: its queue is dispatched first
    for (int h = 1; h < nh && rc == MSKF_OK; ++h) {
        rc = mskf_ctx_create_shared(half_[0].ctx, &half_[h].ctx);
        if (rc == MSKF_OK) rc = mskf_ctx_create_shared(half_[0].ctx_ekf, &half_[h].ctx_ekf);
    }
    if (rc != MSKF_OK) { error_ = mskf_last_error(); return; }
    home_fe_ = half_[0].ctx; home_ekf_ = half_[0].ctx_ekf;
    for (int h = 0; h < nh; ++h) { half_[h].i0 = (int)((long long)n * h / nh); half_[h].n = (int)((long long)n * (h + 1) / nh) - half_[h].i0; }
    for (int i = 0; i < n; ++i) {
        const Half &H = half_[nh == 2 && i >= half_[1].i0 ? 1 : 0];
        systems_.emplace_back(new System(calib, fe, ekf, H.ctx, device));
        if (!systems_.back()->ok()) { error_ = std::string("stream setup failed: ") + mskf_last_error(); return; }
        systems_.back()->copy_draw_buffers = false;
        systems_.back()->imgproc_ptr_->setCompactTail(true);      // the Q1 tail of the message as a count (image_processor.h)
        streams_.push_back(systems_.back()->stream());
        // the filter half of every stream runs on its own context (own HIP stream): no device data is shared
        if (mskf_stream_set_ekf_ctx(streams_.back(), H.ctx_ekf) != MSKF_OK) { error_ = mskf_last_error(); return; }
    }
    a1_.resize(n); a2_.resize(n); u_.resize(n); p0_.resize(n); p1_.resize(n); t_.resize(n);
    seq.resize(n);
    ok_ = true;
}

BatchGroup::~BatchGroup() {
    rebind_home();
    if (ekf_tail_) mskf_point_destroy(ekf_tail_);
    systems_.clear();
    for (size_t h = half_.size(); h-- > 0;) {      // shared contexts first: they borrow half 0's streams
        if (half_[h].ctx_ekf) mskf_ctx_destroy(half_[h].ctx_ekf);
        if (half_[h].ctx) mskf_ctx_destroy(half_[h].ctx);
    }
}

void BatchGroup::imu(int i, const mskf_imu_sample &s) {
    std::shared_ptr<Imu> m(new Imu);
    m->time_stamp = s.time_stamp;
    m->angular_velocity = Vector3(s.angular_velocity[0], s.angular_velocity[1], s.angular_velocity[2]);
    m->linear_acceleration = Vector3(s.linear_acceleration[0], s.linear_acceleration[1], s.linear_acceleration[2]);
    systems_[i]->imu_callback(m);
}

#define BR_CHK(expr) do { int _rc = (expr); if (_rc != MSKF_OK) { error_ = std::string(#expr) + ": " + mskf_last_error(); return _rc; } } while (0)

// Front-end of one frame of every stream (System::stereo_callback): push (pyramids + detector) -> track (temporal LK,
// stereo LK, gates) -> host bucketing / candidates -> track (candidates) -> host (ids, prune, publish).  The halves are
// staggered: while the device tracks one half the thread does the other half's host part.
int BatchGroup::step_fe(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t, bool is_draw) {
    const int n = size();
    if (!ok_ || n == 0) return MSKF_ERR_INVALID;
    auto tp = std::chrono::steady_clock::now();
    double *acc = acc_fe_ ? acc_fe_ : phase_s;
    auto lap = [&](int ph) { auto t2 = std::chrono::steady_clock::now(); acc[ph] += std::chrono::duration<double>(t2 - tp).count(); tp = t2; };
    auto parh = [&](const Half &H, const std::function<void(int)> &fn) {
        if (pool_) pool_->run(H.n, [&](int k) { fn(H.i0 + k); }); else for (int i = H.i0; i < H.i0 + H.n; ++i) fn(i);
    };
    // Every frame after a stream's first runs as ONE device call per half-batch (mskf_fe_frame_batch_*): pyramids, detector,
    // both track calls and the bookkeeping between and after them; the host prepares the prediction and takes the grid.
    {
        bool all_dev = true;
        for (int i = 0; i < n && all_dev; ++i) all_dev = systems_[i]->imgproc_ptr_->canDeviceFrame();
        if (all_dev) {
            fa_.resize(n);
            for (Half &H : half_) {
                bool ok = true;
                for (int i = H.i0; i < H.i0 + H.n; ++i) ok = systems_[i]->imgproc_ptr_->frameBegin(t[i], fa_[i]) && ok;
                if (!ok) { error_ = "frameBegin failed: " + systems_[H.i0]->imgproc_ptr_->error(); return MSKF_ERR_INVALID; }
                lap(PH_PREP1);
                BR_CHK(mskf_fe_frame_batch_begin(H.ctx, H.n, streams_.data() + H.i0, cam0 + H.i0, cam1 + H.i0, on_device, fa_.data() + H.i0));
                lap(PH_PUSH);
            }
            for (Half &H : half_) {
                BR_CHK(mskf_fe_frame_batch_end(H.ctx));
                lap(PH_TRACK1);
                parh(H, [&](int i) {
                    systems_[i]->imgproc_ptr_->frameEnd(fa_[i], is_draw);
                    systems_[i]->set_feature_msg(systems_[i]->imgproc_ptr_->feature_msg_ptr_);
                });
                lap(PH_AFTER2);
            }
            return MSKF_OK;
        }
    }
    // image size comes from the calibration the stream was created with
    for (int i = 0; i < n; ++i) systems_[i]->imgproc_ptr_->phaseBegin(t[i], 0, 0);
    for (Half &H : half_) BR_CHK(mskf_fe_push_stereo_batch(H.ctx, H.n, streams_.data() + H.i0, cam0 + H.i0, cam1 + H.i0, on_device));
    lap(PH_PUSH);
    const bool first = systems_[0]->imgproc_ptr_->isFirstImage();
    for (Half &H : half_) {
        if (first) BR_CHK(mskf_ctx_sync(H.ctx));   // first frame: detections are read right away
        parh(H, [&](int i) { systems_[i]->imgproc_ptr_->phasePrepare1(a1_[i]); });
        lap(PH_PREP1);
        BR_CHK(mskf_fe_track_batch_begin(H.ctx, H.n, streams_.data() + H.i0, a1_.data() + H.i0));
        lap(PH_TRACK1);
    }
    for (Half &H : half_) {
        BR_CHK(mskf_fe_track_batch_end(H.ctx));     // (the detector's per-cell maxima of this push have arrived with it)
        lap(PH_TRACK1);
        parh(H, [&](int i) { systems_[i]->imgproc_ptr_->phaseAfter1(a2_[i]); });
        lap(PH_AFTER1);
        BR_CHK(mskf_fe_track_batch_begin(H.ctx, H.n, streams_.data() + H.i0, a2_.data() + H.i0));
        lap(PH_TRACK2);
    }
    for (Half &H : half_) {
        BR_CHK(mskf_fe_track_batch_end(H.ctx));
        lap(PH_TRACK2);
        parh(H, [&](int i) {
            systems_[i]->imgproc_ptr_->phaseAfter2(is_draw);
            systems_[i]->set_feature_msg(systems_[i]->imgproc_ptr_->feature_msg_ptr_);
        });
        lap(PH_AFTER2);
    }
    return MSKF_OK;
}

// Filter of one frame of every stream (System::backend_callback -> MsckfVio::featureCallback): predict (IMU propagation +
// augmentation) -> lost-feature update -> host -> pruning update -> clone removal -> position variances, staggered over
// the halves like the front-end.
int BatchGroup::step_ekf(const FrameBatch *fb) {
    const int n = size();
    auto tp = std::chrono::steady_clock::now();
    double *acc = acc_ekf_ ? acc_ekf_ : phase_s;
    auto lap = [&](int ph) { auto t2 = std::chrono::steady_clock::now(); acc[ph] += std::chrono::duration<double>(t2 - tp).count(); tp = t2; };
    auto parh = [&](const Half &H, const std::function<void(int)> &fn) {
        if (pool_ekf_) pool_ekf_->run(H.n, [&](int k) { fn(H.i0 + k); }); else for (int i = H.i0; i < H.i0 + H.n; ++i) fn(i);
    };
    std::vector<std::shared_ptr<CameraMeasurement>> msgs(n);
    for (int i = 0; i < n; ++i) {
        MsckfVio &v = *systems_[i]->msckfvio_ptr();
        if (fb) { msgs[i] = fb->msg[i]; v.setZeroTailHint(msgs[i].get(), fb->tail_start[i], fb->total[i]); }
        else {
            msgs[i] = systems_[i]->feature_msg();
            v.setZeroTailHint(msgs[i].get(), systems_[i]->imgproc_ptr_->zeroTailStart(), systems_[i]->imgproc_ptr_->messageSize());
        }
    }
    // streams of the half with a non-empty update -> one batched launch (args stay in H until the *_end call)
    auto begin_updates = [&](Half &H) -> int {
        H.sub_s.clear(); H.sub_a.clear(); H.sub_i.clear(); H.upd_pending = false;
        for (int i = H.i0; i < H.i0 + H.n; ++i) if (u_[i].n_feat > 0) { H.sub_s.push_back(streams_[i]); H.sub_a.push_back(u_[i]); H.sub_i.push_back(i); }
        if (H.sub_s.empty()) return MSKF_OK;
        H.upd_pending = true;
        return mskf_ekf_update_batch_begin(H.ctx_ekf, (int)H.sub_s.size(), H.sub_s.data(), H.sub_a.data());
    };
    auto end_updates = [&](Half &H) -> int {
        if (!H.upd_pending) return MSKF_OK;
        H.upd_pending = false;
        return mskf_ekf_update_batch_end(H.ctx_ekf);
    };
    bool any_all = false;
    for (Half &H : half_) {
        parh(H, [&](int i) { systems_[i]->msckfvio_ptr()->phaseA(msgs[i], u_[i], true); });
        H.any = false;
        for (int i = H.i0; i < H.i0 + H.n; ++i) H.any |= systems_[i]->msckfvio_ptr()->frameActive();
        any_all |= H.any;
        if (H.any) {
            H.ns.assign(H.n, 0); H.sp.assign(H.n, nullptr); H.jp.assign(H.n, nullptr);
            for (int k = 0; k < H.n; ++k) {
                MsckfVio &v = *systems_[H.i0 + k]->msckfvio_ptr();
                const bool act = v.frameActive();
                H.ns[k] = act ? (int)v.predictSteps().size() : 0;
                H.sp[k] = H.ns[k] ? v.predictSteps().data() : nullptr;
                H.jp[k] = act ? v.predictJ() : nullptr;
            }
            BR_CHK(mskf_ekf_predict_batch(H.ctx_ekf, H.n, streams_.data() + H.i0, H.ns.data(), H.sp.data(), H.jp.data()));
        }
        lap(PH_EKF_A);
        if (H.any) BR_CHK(begin_updates(H));
        lap(PH_UPD1);
    }
    if (!any_all) return MSKF_OK;
    for (Half &H : half_) {
        if (!H.any) continue;
        BR_CHK(end_updates(H));
        lap(PH_UPD1);
        parh(H, [&](int i) { if (systems_[i]->msckfvio_ptr()->frameActive()) systems_[i]->msckfvio_ptr()->phaseB(u_[i]); else std::memset(&u_[i], 0, sizeof(u_[i])); });
        lap(PH_EKF_B);
        BR_CHK(begin_updates(H));
        lap(PH_UPD2);
    }
    for (Half &H : half_) {
        if (!H.any) continue;
        BR_CHK(end_updates(H));
        lap(PH_UPD2);
        H.rm.assign(2 * (size_t)H.n, -1);
        parh(H, [&](int i) {
            MsckfVio &v = *systems_[i]->msckfvio_ptr();
            v.phaseC(true);
            H.rm[2 * (i - H.i0)] = v.pendingRemovals()[0]; H.rm[2 * (i - H.i0) + 1] = v.pendingRemovals()[1];
        });
        bool any_rm = false;
        for (int k = 0; k < H.n; ++k) any_rm |= H.rm[2 * k] >= 0;
        if (any_rm) BR_CHK(mskf_ekf_remove_clones_batch(H.ctx_ekf, H.n, streams_.data() + H.i0, H.rm.data()));
        lap(PH_EKF_C);
        // onlineReset (msckf_vio.cpp:1186-1236) needs P(12..14) of every stream: they came back with the frame's last update;
        // only when some stream had no update at all this frame they are fetched with a launch and a wait of their own
        bool all_pv = true;
        for (int k = 0; k < H.n && all_pv; ++k) { const MsckfVio &v = *systems_[H.i0 + k]->msckfvio_ptr(); all_pv = !v.frameActive() || v.havePosVar(); }
        if (all_pv) {
            for (int k = 0; k < H.n; ++k) { MsckfVio &v = *systems_[H.i0 + k]->msckfvio_ptr(); if (v.frameActive()) v.phaseD(v.posVar()); }
            lap(PH_POSVAR);
            continue;
        }
        H.pv.assign(3 * (size_t)H.n, 0.0);
        BR_CHK(mskf_ekf_get_pos_var_batch_begin(H.ctx_ekf, H.n, streams_.data() + H.i0, H.pv.data()));
        H.pv_pending = true;
        lap(PH_POSVAR);
    }
    for (Half &H : half_) {
        if (!H.pv_pending) continue;
        H.pv_pending = false;
        BR_CHK(mskf_ekf_get_pos_var_batch_end(H.ctx_ekf));
        for (int k = 0; k < H.n; ++k) systems_[H.i0 + k]->msckfvio_ptr()->phaseD(&H.pv[3 * k]);
        lap(PH_POSVAR);
    }
    return MSKF_OK;
}

int BatchGroup::step(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t, bool is_draw) {
    int rc = step_fe(cam0, cam1, on_device, t, is_draw);
    if (rc != MSKF_OK) return rc;
    return step_ekf(nullptr);
}

static double ns_to_sec(long long ns) {   // apps/run_euroc_single_thread.cpp:164-166,192 (Q9)
    const long long sec = ns / 1000000000LL, nsec = ns % 1000000000LL;
    const double stamp_ns = (double)(int)sec * 1e9 + (double)(int)nsec;
    return stamp_ns * 1e-9;
}

// do { feed IMU } while (t_imu <= t_img)  (apps/run_euroc_single_thread.cpp:209-238, Q10) for frame k of every stream
int BatchGroup::feed_imu(int k, bool to_fe, bool to_ekf) {
    const int n = size();
    for (int i = 0; i < n; ++i) {
        StreamSequence &q = seq[i];
        if (!q.cam0_base || !q.imu) { error_ = "no sequence attached"; return MSKF_ERR_INVALID; }
        const double t_img = ns_to_sec(q.t0_ns + (long long)k * q.frame_dt_ns);
        int &cur = to_fe ? q.imu_cursor : q.imu_cursor_ekf;
        double t_imu = 0.0;
        do {
            if (cur >= q.n_imu) { error_ = "IMU sequence exhausted"; return MSKF_ERR_CAPACITY; }
            const mskf_imu_sample &s = q.imu[cur++];
            std::shared_ptr<Imu> m(new Imu);
            m->time_stamp = s.time_stamp;
            m->angular_velocity = Vector3(s.angular_velocity[0], s.angular_velocity[1], s.angular_velocity[2]);
            m->linear_acceleration = Vector3(s.linear_acceleration[0], s.linear_acceleration[1], s.linear_acceleration[2]);
            if (to_fe) systems_[i]->imgproc_ptr_->imuCallback(m);      // System::imu_callback, system.cpp:45-48
            if (to_ekf) systems_[i]->msckfvio_ptr()->imuCallback(m);
            t_imu = s.time_stamp;
        } while (t_imu <= t_img);
        if (to_fe && to_ekf) q.imu_cursor_ekf = q.imu_cursor;
        if (to_fe) {
            const int key = k < q.n_static ? k : q.n_static + (k - q.n_static) % q.n_loop;
            p0_[i] = q.cam0_base + (size_t)key * q.frame_bytes;
            p1_[i] = q.cam1_base + (size_t)key * q.frame_bytes;
            t_[i] = t_img;
        }
    }
    return MSKF_OK;
}

int BatchGroup::run(int first, int n_frames) {
    for (int k = first; k < first + n_frames; ++k) {
        auto t_imu0 = std::chrono::steady_clock::now();
        int rc = feed_imu(k, true, true);
        if (rc != MSKF_OK) return rc;
        phase_s[PH_IMU] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_imu0).count();
        rc = step(p0_.data(), p1_.data(), seq[0].on_device, t_.data(), false);
        if (rc != MSKF_OK) return rc;
    }
    return MSKF_OK;
}

// hand-off = a snapshot of every stream's message (live + stale entries + the first tail record); `fb` is recycled when given
void BatchGroup::fill_handoff(int k, std::unique_ptr<FrameBatch> &fb) {
    const int n = size();
    if (!fb) fb.reset(new FrameBatch);
    fb->frame = k;
    fb->msg.resize(n); fb->tail_start.resize(n); fb->total.resize(n);
    for (int i = 0; i < n; ++i) {
        const ImageProcessor &ip = *systems_[i]->imgproc_ptr_;
        const CameraMeasurement &live = *ip.feature_msg_ptr_;
        const size_t total = ip.messageSize(), start = ip.zeroTailStart();
        const size_t keep = std::min(live.features.size(), start + 1);
        if (!fb->msg[i] || fb->msg[i].use_count() > 1) fb->msg[i].reset(new CameraMeasurement);
        fb->msg[i]->time_stamp = live.time_stamp;
        fb->msg[i]->features.assign(live.features.begin(), live.features.begin() + keep);
        fb->tail_start[i] = start; fb->total[i] = total;
    }
}

void BatchGroup::snapshot_fe_mark() {
    std::vector<ImageProcessor::FeatureIDType> ids;
    systems_[0]->imgproc_ptr_->dumpCurrent(ids, mark_dump.life, mark_dump.c0, mark_dump.c1);
    mark_dump.ids.assign(ids.begin(), ids.end());
    mark_dump.fe_valid = true;
}

void BatchGroup::snapshot_ekf_mark() {
    const IMUState &st = systems_[0]->msckfvio_ptr()->imuState();
    int k = 0;
    for (int i = 0; i < 4; ++i) mark_dump.imu[k++] = st.orientation.q[i];
    for (int i = 0; i < 3; ++i) mark_dump.imu[k++] = st.position[i];
    for (int i = 0; i < 3; ++i) mark_dump.imu[k++] = st.velocity[i];
    for (int i = 0; i < 3; ++i) mark_dump.imu[k++] = st.gyro_bias[i];
    for (int i = 0; i < 3; ++i) mark_dump.imu[k++] = st.acc_bias[i];
    for (int i = 0; i < 9; ++i) mark_dump.imu[k++] = st.R_imu_cam0.m[i];
    for (int i = 0; i < 3; ++i) mark_dump.imu[k++] = st.t_cam0_imu[i];
    mark_dump.ekf_valid = true;
}

void BatchGroup::rebind_home() {
    if (!home_fe_ || !home_ekf_) return;
    half_[0].ctx = home_fe_; half_[0].ctx_ekf = home_ekf_;
    for (mskf_stream *s : streams_) mskf_stream_rebind(s, home_fe_, home_ekf_);
    acc_fe_ = acc_ekf_ = nullptr;
}

// front-end stage of frame k on a borrowed context (the caller has made sure nothing of this batch's front-end is in flight)
int BatchGroup::fe_stage(mskf_ctx *ctx, int k, double *acc, std::unique_ptr<FrameBatch> &out) {
    if (half_.size() != 1) { error_ = "the balanced runner needs one batch per stage (halves = 1)"; return MSKF_ERR_UNSUPPORTED; }
    if (half_[0].ctx != ctx) { half_[0].ctx = ctx; for (mskf_stream *s : streams_) mskf_stream_rebind(s, ctx, nullptr); }
    acc_fe_ = acc;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = feed_imu(k, true, false);
    const auto t1 = std::chrono::steady_clock::now();
    acc[PH_IMU] += std::chrono::duration<double>(t1 - t0).count();
    if (rc == MSKF_OK) rc = step_fe(p0_.data(), p1_.data(), seq[0].on_device, t_.data(), false);
    if (rc != MSKF_OK) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    fill_handoff(k, out);
    acc[PH_HANDOFF] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count();
    return MSKF_OK;
}

// filter stage of a handed-off frame on a borrowed context.  The stage ends with work it does not wait for (clone removal):
// when the next frame of this batch runs on another context, that context's queue is ordered behind it.
int BatchGroup::ekf_stage(mskf_ctx *ctx, FrameBatch *fb, double *acc) {
    if (half_.size() != 1) { error_ = "the balanced runner needs one batch per stage (halves = 1)"; return MSKF_ERR_UNSUPPORTED; }
    if (half_[0].ctx_ekf != ctx) { half_[0].ctx_ekf = ctx; for (mskf_stream *s : streams_) mskf_stream_rebind(s, nullptr, ctx); }
    if (ekf_tail_ && ekf_tail_ctx_ && ekf_tail_ctx_ != ctx) { const int wrc = mskf_ctx_wait_point(ctx, ekf_tail_); if (wrc != MSKF_OK) { error_ = mskf_last_error(); return wrc; } }
    acc_ekf_ = acc;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = feed_imu(fb->frame, false, true);
    acc[PH_IMU_EKF] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc == MSKF_OK) rc = step_ekf(fb);
    if (rc != MSKF_OK) return rc;
    rc = mskf_ctx_record_point(ctx, &ekf_tail_);
    ekf_tail_ctx_ = ctx;
    if (rc != MSKF_OK) error_ = mskf_last_error();
    return rc;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void BatchGroup::set_gates(bool on) {
    for (Half &H : half_) { mskf_ctx_timing_gate(H.ctx, on ? 1 : 0); mskf_ctx_timing_gate(H.ctx_ekf, on ? 1 : 0); }
}

int BatchGroup::run_pipelined(int first, int n_frames, TimedWindow *win) {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<FrameBatch>> queue;
    std::vector<std::unique_ptr<FrameBatch>> pool;      // consumed batches, reused by the producer
    bool producer_done = false;
    std::atomic<int> ekf_rc{MSKF_OK};
    std::string ekf_err;
    static const int fe_phases[] = {PH_PUSH, PH_PREP1, PH_TRACK1, PH_AFTER1, PH_TRACK2, PH_AFTER2, PH_IMU, PH_HANDOFF, PH_FE_QWAIT};
    static const int ekf_phases[] = {PH_EKF_A, PH_UPD1, PH_EKF_B, PH_UPD2, PH_EKF_C, PH_POSVAR, PH_EKF_QWAIT, PH_IMU_EKF};
    // a stage opens / closes its own accounting: kernel timing of its contexts, the host-profile slots of its thread, and the
    // phase times it owns (difference between the two marks)
    auto gate = [&](bool fe, bool on) {
        for (Half &H : half_) mskf_ctx_timing_gate(fe ? H.ctx : H.ctx_ekf, on ? 1 : 0);
        hostprof::enabled() = on;
        const int *ph = fe ? fe_phases : ekf_phases;
        const int cnt = fe ? (int)(sizeof(fe_phases) / sizeof(int)) : (int)(sizeof(ekf_phases) / sizeof(int));
        for (int k = 0; k < cnt; ++k) window_phase_s[ph[k]] = on ? -phase_s[ph[k]] : window_phase_s[ph[k]] + phase_s[ph[k]];
    };
    if (win) { mark_dump.fe_valid = mark_dump.ekf_valid = false; }
    TimedShared *sh = win ? win->shared : nullptr;
    // a stage follows the shared window at its frame boundaries: 0 -> not opened yet, 1 -> open, 2 -> closed
    auto follow = [&](bool fe, int &mine, double &t_begin, double &t_end) {
        if (!sh) return;
        const int ph = sh->phase.load(std::memory_order_acquire);
        if (mine == 0 && ph >= 1) { t_begin = now_s(); gate(fe, true); mine = 1; }
        if (mine == 1 && ph == 2) { t_end = now_s(); gate(fe, false); mine = 2; }
    };
    int ekf_mine = 0, fe_mine = 0;
    std::thread consumer([&]() {
        if (win) hostprof::enabled() = false;
        for (;;) {
            std::unique_ptr<FrameBatch> fb;
            {
                const double tq = now_s();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return !queue.empty() || producer_done; });
                if (queue.empty()) break;
                fb = std::move(queue.front());
                queue.pop_front();
                phase_s[PH_EKF_QWAIT] += now_s() - tq;
            }
            cv.notify_all();
            if (ekf_rc.load() != MSKF_OK) continue;   // drain
            if (win) { follow(false, ekf_mine, win->t_ekf_begin, win->t_ekf_end); if (ekf_mine == 1) ++win->ekf_frames; }
            const double ti = now_s();
            int rc = feed_imu(fb->frame, false, true);
            phase_s[PH_IMU_EKF] += now_s() - ti;
            if (rc == MSKF_OK) rc = step_ekf(fb.get());
            if (rc != MSKF_OK) ekf_rc.store(rc);
            if (sh && rc == MSKF_OK) {
                // the frame is through both stages: it counts; the thread that completes the opening / closing frame stamps the window
                const long c = sh->completed.fetch_add(1) + 1;
                if (c == sh->target_open) { sh->t_open = now_s(); sh->phase.store(1, std::memory_order_release); }
                if (c == sh->target_close) { sh->t_close = now_s(); sh->phase.store(2, std::memory_order_release); win->frames_at_close = fb->frame + 1 - first; }
            }
            if (win && fb->frame == win->mark_end - 1) {
                if (rc == MSKF_OK) snapshot_ekf_mark();
            }
            { std::lock_guard<std::mutex> lk(mu); pool.push_back(std::move(fb)); }
        }
        // the window closed while this stage was inside its last frames (or never did): its accounting ends with its work
        if (win && ekf_mine == 1) { win->t_ekf_end = now_s(); gate(false, false); ekf_mine = 2; }
    });
    if (win) hostprof::enabled() = false;
    int rc = MSKF_OK;
    int k = first;
    for (;; ++k) {
        if (rc != MSKF_OK || ekf_rc.load() != MSKF_OK) break;
        if (k >= first + n_frames) {
            // past its own frames a group keeps the device loaded until the shared window is closed
            if (!sh || sh->phase.load() == 2 || k >= first + n_frames + win->max_extra) break;
        }
        if (win) { follow(true, fe_mine, win->t_fe_begin, win->t_fe_end); if (fe_mine == 1) ++win->fe_frames; }
        const double ti = now_s();
        rc = feed_imu(k, true, false);
        phase_s[PH_IMU] += now_s() - ti;
        if (rc == MSKF_OK) rc = step_fe(p0_.data(), p1_.data(), seq[0].on_device, t_.data(), false);
        if (rc != MSKF_OK) break;
        // hand-off = a snapshot of every stream's message (live + stale entries + the first tail record).  The batches and
        // their per-stream messages are recycled through `pool` (at most 2 queued + 1 in the filter stage + 1 being
        // filled), so the steady state allocates nothing
        const double th = now_s();
        std::unique_ptr<FrameBatch> fb;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!pool.empty()) { fb = std::move(pool.back()); pool.pop_back(); }
        }
        fill_handoff(k, fb);
        const double tw = now_s();
        phase_s[PH_HANDOFF] += tw - th;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return queue.size() < 2; });
            queue.push_back(std::move(fb));
        }
        cv.notify_all();
        phase_s[PH_FE_QWAIT] += now_s() - tw;
        if (win && k == win->mark_end - 1) {
            snapshot_fe_mark();
        }
    }
    if (win && fe_mine == 1) { win->t_fe_end = now_s(); gate(true, false); fe_mine = 2; }
    { std::lock_guard<std::mutex> lk(mu); producer_done = true; }
    cv.notify_all();
    consumer.join();
    if (win) win->frames_done = k - first;
    hostprof::enabled() = true;
    if (rc == MSKF_OK) rc = ekf_rc.load();
    return rc;
}

MultiRunner::MultiRunner(int device, int n_groups, int per_group, const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf,
                         int host_threads, int ekf_host_threads, int halves)
    : n_groups_(n_groups), per_group_(per_group), off_(n_groups, 0), next_(n_groups, 0), win_(n_groups) {
    // A device offers 16 hardware queues before streams get multiplexed (GPU_MAX_HW_QUEUES): two per group, front-end and filter
    for (int g = 0; g < n_groups; ++g) groups_.emplace_back(new BatchGroup(device, per_group, calib, fe, ekf, host_threads, ekf_host_threads, halves));
}

MultiRunner::~MultiRunner() { groups_.clear(); }

bool MultiRunner::ok() const {
    for (const auto &g : groups_) if (!g->ok()) return false;
    return !groups_.empty();
}

std::string MultiRunner::error() const {
    for (const auto &g : groups_) if (!g->error().empty()) return g->error();
    return std::string();
}

int MultiRunner::step(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t) {
    for (int g = 0; g < n_groups_; ++g) {
        int rc = groups_[g]->step(cam0 + (size_t)g * per_group_, cam1 + (size_t)g * per_group_, on_device, t + (size_t)g * per_group_, false);
        if (rc != MSKF_OK) return rc;
    }
    return MSKF_OK;
}

int MultiRunner::run(int first, int n, bool threaded, bool pipelined) {
    std::vector<int> rcs(n_groups_, MSKF_OK);
    auto one = [&](int g) {
        // frames [first + off, first + off + n) of the group, after catching up from where it stands
        const int from = (off_[g] > 0 || next_[g] > 0) ? std::min(next_[g], first + off_[g]) : first;
        const int cnt = first + off_[g] + n - from;
        next_[g] = from + cnt;
        return pipelined ? groups_[g]->run_pipelined(from, cnt) : groups_[g]->run(from, cnt);
    };
    if (!threaded || n_groups_ == 1) {
        for (int g = 0; g < n_groups_; ++g) { rcs[g] = one(g); if (rcs[g] != MSKF_OK) return rcs[g]; }
        return MSKF_OK;
    }
    if (pipelined && groups_[0]->n_halves() == 1) return run_balanced(first, 0, n, 0, nullptr, true);
    std::vector<std::thread> th;
    for (int g = 0; g < n_groups_; ++g) th.emplace_back([&, g]() { rcs[g] = one(g); });
    for (auto &t : th) t.join();
    for (int g = 0; g < n_groups_; ++g) if (rcs[g] != MSKF_OK) return rcs[g];
    return MSKF_OK;
}

int MultiRunner::run_timed(int first, int warmup, int steps, int max_extra, double *elapsed_s) {
    if (n_groups_ > 1 && groups_[0]->n_halves() == 1) return run_balanced(first, warmup, steps, max_extra, elapsed_s);
    std::vector<int> rcs(n_groups_, MSKF_OK);
    TimedShared shared;
    shared.target_open = (long)n_groups_ * warmup;
    shared.target_close = (long)n_groups_ * (warmup + steps);
    if (warmup <= 0) { shared.t_open = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); shared.phase.store(1); }
    std::vector<int> from(n_groups_), cnt(n_groups_);
    for (int g = 0; g < n_groups_; ++g) {
        from[g] = (off_[g] > 0 || next_[g] > 0) ? std::min(next_[g], first + off_[g]) : first;
        cnt[g] = first + off_[g] + warmup + steps - from[g];
        TimedWindow &w = win_[g];
        w = TimedWindow();
        w.shared = &shared;
        w.mark_end = first + off_[g] + warmup + steps;
        w.max_extra = max_extra;
        groups_[g]->set_gates(false);
    }
    std::vector<std::thread> th;
    for (int g = 0; g < n_groups_; ++g) th.emplace_back([&, g]() { rcs[g] = groups_[g]->run_pipelined(from[g], cnt[g], &win_[g]); });
    for (auto &t : th) t.join();
    for (int g = 0; g < n_groups_; ++g) {
        next_[g] = from[g] + win_[g].frames_done;
        groups_[g]->set_gates(true);
    }
    for (int g = 0; g < n_groups_; ++g) if (rcs[g] != MSKF_OK) return rcs[g];
    if (shared.phase.load() != 2) return MSKF_ERR_INVALID;        // the window never closed
    if (elapsed_s) *elapsed_s = shared.t_close - shared.t_open;
    return MSKF_OK;
}

int MultiRunner::run_balanced(int first, int warmup, int steps, int max_extra, double *elapsed_s, bool plain) {
    const int nb = n_groups_;
    TimedShared shared;
    shared.target_open = (long)nb * warmup;
    shared.target_close = (long)nb * (warmup + steps);
    if (plain) {     // an ordinary run: every batch does exactly its own frames (catch-up of staggered groups included), accounting on throughout
        shared.target_open = 0; shared.target_close = 0; max_extra = 0;
        for (int g = 0; g < nb; ++g) { const int from = (off_[g] > 0 || next_[g] > 0) ? std::min(next_[g], first + off_[g]) : first; shared.target_close += first + off_[g] + warmup + steps - from; }
        if (shared.target_close <= 0) return MSKF_OK;
    }
    if (shared.target_open <= 0) { shared.t_open = now_s(); shared.phase.store(1); }
    // per batch: frames [from, from + cnt) are its own, then it keeps going while the window is open (at most max_extra)
    struct Batch { int from = 0, cnt = 0, fe_next = 0, mark_end = 0; bool fe_busy = false, ekf_busy = false, fe_done = false; };
    std::vector<Batch> B(nb);
    for (int g = 0; g < nb; ++g) {
        B[g].from = (off_[g] > 0 || next_[g] > 0) ? std::min(next_[g], first + off_[g]) : first;
        B[g].cnt = first + off_[g] + warmup + steps - B[g].from;
        B[g].fe_next = B[g].from;
        B[g].mark_end = first + off_[g] + warmup + steps;
        win_[g] = TimedWindow();
        win_[g].shared = &shared;
        groups_[g]->set_gates(false);
        groups_[g]->mark_dump.fe_valid = groups_[g]->mark_dump.ekf_valid = false;
        groups_[g]->handoff.clear();
        for (int k = 0; k < BatchGroup::PH_COUNT; ++k) groups_[g]->window_phase_s[k] = 0;
    }
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> err{MSKF_OK};
    // a worker's accounting follows the shared window at its frame boundaries (as a stage does in run_pipelined); the phase
    // times of worker w are kept in group w's arrays whichever batches it ran
    static const int fe_phases[] = {BatchGroup::PH_PUSH, BatchGroup::PH_PREP1, BatchGroup::PH_TRACK1, BatchGroup::PH_AFTER1, BatchGroup::PH_TRACK2,
                                    BatchGroup::PH_AFTER2, BatchGroup::PH_IMU, BatchGroup::PH_HANDOFF, BatchGroup::PH_FE_QWAIT};
    static const int ekf_phases[] = {BatchGroup::PH_EKF_A, BatchGroup::PH_UPD1, BatchGroup::PH_EKF_B, BatchGroup::PH_UPD2, BatchGroup::PH_EKF_C,
                                     BatchGroup::PH_POSVAR, BatchGroup::PH_EKF_QWAIT, BatchGroup::PH_IMU_EKF};
    std::vector<mskf_ctx *> fe_ctx(nb), ekf_ctx(nb);      // worker w = the two contexts group w created (captured before any batch moves)
    for (int w = 0; w < nb; ++w) { fe_ctx[w] = groups_[w]->ctx(); ekf_ctx[w] = groups_[w]->ekf_ctx(); }
    auto gate = [&](int w, bool fe, bool on) {
        BatchGroup &G = *groups_[w];
        mskf_ctx_timing_gate(fe ? fe_ctx[w] : ekf_ctx[w], on ? 1 : 0);
        hostprof::enabled() = on;
        const int *ph = fe ? fe_phases : ekf_phases;
        const int cnt = fe ? (int)(sizeof(fe_phases) / sizeof(int)) : (int)(sizeof(ekf_phases) / sizeof(int));
        for (int k = 0; k < cnt; ++k) G.window_phase_s[ph[k]] = on ? -G.phase_s[ph[k]] : G.window_phase_s[ph[k]] + G.phase_s[ph[k]];
    };
    auto follow = [&](int w, bool fe, int &mine, double &t_begin, double &t_end) {
        const int ph = shared.phase.load(std::memory_order_acquire);
        if (mine == 0 && ph >= 1) { t_begin = now_s(); gate(w, fe, true); mine = 1; }
        if (mine == 1 && ph == 2) { t_end = now_s(); gate(w, fe, false); mine = 2; }
    };
    // Workers and batches are different things: there may be fewer workers of a kind than batches (set_workers; default one of
    // each per batch; worker w uses the contexts group w created).
    const int n_few = fe_workers_ > 0 ? std::min(nb, fe_workers_) : nb, n_ekw = ekf_workers_ > 0 ? std::min(nb, ekf_workers_) : nb;
    std::vector<std::thread> th;
    // ---- front-end workers: the batch that is furthest behind, not being worked on, with room in its hand-off queue
    for (int w = 0; w < n_few; ++w) th.emplace_back([&, w]() {
        hostprof::enabled() = false;
        int mine = 0;
        double *acc = groups_[w]->phase_s;
        TimedWindow &W = win_[w];
        for (;;) {
            int b = -1;
            {
                const double tq = now_s();
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (err.load() != MSKF_OK) return;
                    b = -1;
                    bool all_done = true;
                    for (int g = 0; g < nb; ++g) {
                        Batch &X = B[g];
                        if (X.fe_done) continue;
                        const bool own = X.fe_next < X.from + X.cnt;
                        if (!own && (shared.phase.load() == 2 || X.fe_next >= X.from + X.cnt + max_extra)) { X.fe_done = true; cv.notify_all(); continue; }
                        all_done = false;
                        if (X.fe_busy || groups_[g]->handoff.size() >= 2) continue;
                        if (b < 0 || X.fe_next - X.from < B[b].fe_next - B[b].from) b = g;
                    }
                    if (b >= 0 || all_done) break;
                    cv.wait(lk);
                }
                acc[BatchGroup::PH_FE_QWAIT] += now_s() - tq;
                if (b < 0) break;
                B[b].fe_busy = true;
            }
            follow(w, true, mine, W.t_fe_begin, W.t_fe_end);
            if (mine == 1) ++W.fe_frames;
            const int k = B[b].fe_next;
            std::unique_ptr<FrameBatch> fb;
            { std::lock_guard<std::mutex> lk(mu); auto &pool = groups_[b]->handoff_pool; if (!pool.empty()) { fb = std::move(pool.back()); pool.pop_back(); } }
            const int rc = groups_[b]->fe_stage(fe_ctx[w], k, acc, fb);
            if (rc == MSKF_OK && k == B[b].mark_end - 1) groups_[b]->snapshot_fe_mark();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc != MSKF_OK) err.store(rc);
                else { groups_[b]->handoff.push_back(std::move(fb)); ++B[b].fe_next; }
                B[b].fe_busy = false;
            }
            cv.notify_all();
            if (rc != MSKF_OK) return;
        }
        if (mine == 1) { W.t_fe_end = now_s(); gate(w, true, false); }
    });
    // ---- filter workers: the oldest handed-off frame of a batch whose filter is not being worked on
    for (int w = 0; w < n_ekw; ++w) th.emplace_back([&, w]() {
        hostprof::enabled() = false;
        int mine = 0;
        double *acc = groups_[w]->phase_s;
        TimedWindow &W = win_[w];
        for (;;) {
            int b = -1;
            std::unique_ptr<FrameBatch> fb;
            {
                const double tq = now_s();
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (err.load() != MSKF_OK) return;
                    b = -1;
                    bool pending = false;
                    for (int g = 0; g < nb; ++g) {
                        if (!B[g].fe_done || !groups_[g]->handoff.empty() || B[g].ekf_busy) pending = true;
                        if (B[g].ekf_busy || groups_[g]->handoff.empty()) continue;
                        if (b < 0 || groups_[g]->handoff.front()->frame - B[g].from < groups_[b]->handoff.front()->frame - B[b].from) b = g;
                    }
                    if (b >= 0 || !pending) break;
                    cv.wait(lk);
                }
                acc[BatchGroup::PH_EKF_QWAIT] += now_s() - tq;
                if (b < 0) break;
                B[b].ekf_busy = true;
                fb = std::move(groups_[b]->handoff.front());
                groups_[b]->handoff.pop_front();
            }
            cv.notify_all();
            follow(w, false, mine, W.t_ekf_begin, W.t_ekf_end);
            if (mine == 1) ++W.ekf_frames;
            const int frame = fb->frame;
            const int rc = groups_[b]->ekf_stage(ekf_ctx[w], fb.get(), acc);
            if (rc == MSKF_OK && frame == B[b].mark_end - 1) groups_[b]->snapshot_ekf_mark();     // (before the batch is released to the next worker)
            {
                // the frame is through both stages: it counts.  Progress and the shared count move together under the lock, so the
                // thread that completes the closing frame sees how far every batch had got at that moment (the frames a batch
                // finishes after that are drain, done but not counted)
                std::lock_guard<std::mutex> lk(mu);
                if (rc != MSKF_OK) err.store(rc);
                else {
                    win_[b].frames_done = frame + 1 - B[b].from;
                    const long c = shared.completed.fetch_add(1) + 1;
                    if (c == shared.target_open) { shared.t_open = now_s(); shared.phase.store(1, std::memory_order_release); }
                    if (c == shared.target_close) {
                        shared.t_close = now_s(); shared.phase.store(2, std::memory_order_release);
                        for (int g = 0; g < nb; ++g) win_[g].frames_at_close = win_[g].frames_done;
                    }
                }
                groups_[b]->handoff_pool.push_back(std::move(fb));
                B[b].ekf_busy = false;
            }
            cv.notify_all();
            if (rc != MSKF_OK) return;
        }
        if (mine == 1) { W.t_ekf_end = now_s(); gate(w, false, false); }
    });
    for (auto &t : th) t.join();
    hostprof::enabled() = true;
    // every worker's queue is drained (the last clone removals are not waited for by their stage), then the batches go home
    for (int w = 0; w < nb; ++w) { mskf_ctx_sync(fe_ctx[w]); mskf_ctx_sync(ekf_ctx[w]); }
    for (int g = 0; g < nb; ++g) {
        groups_[g]->rebind_home();
        groups_[g]->set_gates(true);
        next_[g] = B[g].from + win_[g].frames_done;
    }
    if (err.load() != MSKF_OK) return err.load();
    if (shared.phase.load() != 2) return MSKF_ERR_INVALID;
    if (elapsed_s) *elapsed_s = shared.t_close - shared.t_open;
    return MSKF_OK;
}

}  // namespace cg
