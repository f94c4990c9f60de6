// batch_runner.cpp — see batch_runner.h.
#include "batch_runner.h"
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
        }
        for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*fn)(i);
        done_.fetch_add(1);
    }
}
void ForkJoin::run(int n, const std::function<void(int)> &fn) {
    if (nt_ <= 1 || n <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    next_.store(0); done_.store(0);
    { std::lock_guard<std::mutex> lk(mu_); fn_ = &fn; n_ = n; ++gen_; }
    cv_.notify_all();
    for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) fn(i);
    while (done_.load() < nt_ - 1) std::this_thread::yield();
}

BatchGroup::BatchGroup(int device, int n, const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, int host_threads) {
    // per-stream host phases of a group are independent: optional helper threads for the front-end / filter halves
    // (MSKF_FE_HOST_THREADS / MSKF_EKF_HOST_THREADS override the common host_threads argument)
    int ht_fe = host_threads, ht_ekf = host_threads;
    if (const char *e = std::getenv("MSKF_FE_HOST_THREADS")) ht_fe = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("MSKF_EKF_HOST_THREADS")) ht_ekf = std::max(1, std::atoi(e));
    if (ht_fe > 1) pool_.reset(new ForkJoin(ht_fe));
    if (ht_ekf > 1) pool_ekf_.reset(new ForkJoin(ht_ekf));
    int rc = mskf_ctx_create(device, &ctx_);
    if (rc == MSKF_OK) {
        const char *pe = std::getenv("MSKF_EKF_PRIORITY");   // default on; MSKF_EKF_PRIORITY=0 disables
        rc = mskf_ctx_create_prio(device, !(pe && pe[0] == '0'), &ctx_ekf_);
    }
    if (rc != MSKF_OK) { error_ = mskf_last_error(); return; }
    for (int i = 0; i < n; ++i) {
        systems_.emplace_back(new System(calib, fe, ekf, ctx_, device));
        if (!systems_.back()->ok()) { error_ = std::string("stream setup failed: ") + mskf_last_error(); return; }
        systems_.back()->copy_draw_buffers = false;
        streams_.push_back(systems_.back()->stream());
        // the filter half of every stream runs on its own context (own HIP stream): no device data is shared
        if (mskf_stream_set_ekf_ctx(streams_.back(), ctx_ekf_) != MSKF_OK) { error_ = mskf_last_error(); return; }
    }
    a1_.resize(n); a2_.resize(n); u_.resize(n); p0_.resize(n); p1_.resize(n); t_.resize(n);
    seq.resize(n);
    ok_ = true;
}

BatchGroup::~BatchGroup() {
    systems_.clear();
    if (ctx_ekf_) mskf_ctx_destroy(ctx_ekf_);
    if (ctx_) mskf_ctx_destroy(ctx_);
}

void BatchGroup::imu(int i, const mskf_imu_sample &s) {
    std::shared_ptr<Imu> m(new Imu);
    m->time_stamp = s.time_stamp;
    m->angular_velocity = Vector3(s.angular_velocity[0], s.angular_velocity[1], s.angular_velocity[2]);
    m->linear_acceleration = Vector3(s.linear_acceleration[0], s.linear_acceleration[1], s.linear_acceleration[2]);
    systems_[i]->imu_callback(m);
}

#define BR_CHK(expr) do { int _rc = (expr); if (_rc != MSKF_OK) { error_ = std::string(#expr) + ": " + mskf_last_error(); return _rc; } } while (0)

int BatchGroup::step_fe(const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t, bool is_draw) {
    const int n = size();
    if (!ok_ || n == 0) return MSKF_ERR_INVALID;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](int ph) { auto t = std::chrono::steady_clock::now(); phase_s[ph] += std::chrono::duration<double>(t - tp).count(); tp = t; };
    // ---- front-end (System::stereo_callback for every stream)
    for (int i = 0; i < n; ++i) {
        mskf_stream *s = streams_[i];
        (void)s;
        ImageProcessor &ip = *systems_[i]->imgproc_ptr_;
        // image size comes from the calibration the stream was created with
        ip.phaseBegin(t[i], 0, 0);
    }
    BR_CHK(mskf_fe_push_stereo_batch(ctx_, n, streams_.data(), cam0, cam1, on_device));
    lap(PH_PUSH);
    if (systems_[0]->imgproc_ptr_->isFirstImage()) BR_CHK(mskf_ctx_sync(ctx_));   // first frame: detections are read right away
    par(n, [&](int i) { systems_[i]->imgproc_ptr_->phasePrepare1(a1_[i]); });
    lap(PH_PREP1);
    BR_CHK(mskf_fe_track_batch(ctx_, n, streams_.data(), a1_.data()));
    lap(PH_TRACK1);
    BR_CHK(mskf_ctx_sync(ctx_));
    par(n, [&](int i) { systems_[i]->imgproc_ptr_->phaseAfter1(a2_[i]); });
    lap(PH_AFTER1);
    BR_CHK(mskf_fe_track_batch(ctx_, n, streams_.data(), a2_.data()));
    lap(PH_TRACK2);
    par(n, [&](int i) {
        systems_[i]->imgproc_ptr_->phaseAfter2(is_draw);
        systems_[i]->set_feature_msg(systems_[i]->imgproc_ptr_->feature_msg_ptr_);
    });
    lap(PH_AFTER2);
    return MSKF_OK;
}

int BatchGroup::step_ekf(const FrameBatch *fb) {
    const int n = size();
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](int ph) { auto t = std::chrono::steady_clock::now(); phase_s[ph] += std::chrono::duration<double>(t - tp).count(); tp = t; };
    mskf_ctx *ctx_ = ctx_ekf_;   // every device call below belongs to the filter context
    std::vector<std::shared_ptr<CameraMeasurement>> msgs(n);
    for (int i = 0; i < n; ++i) {
        MsckfVio &v = *systems_[i]->msckfvio_ptr();
        if (fb) { msgs[i] = fb->msg[i]; v.setZeroTailHint(msgs[i].get(), fb->tail_start[i], fb->total[i]); }
        else {
            msgs[i] = systems_[i]->feature_msg();
            v.setZeroTailHint(msgs[i].get(), systems_[i]->imgproc_ptr_->zeroTailStart());
        }
    }
    std::vector<mskf_stream *> sub_s;
    std::vector<mskf_ekf_update_args> sub_a;
    std::vector<int> sub_i;
    auto run_updates = [&]() -> int {
        sub_s.clear(); sub_a.clear(); sub_i.clear();
        for (int i = 0; i < n; ++i) if (u_[i].n_feat > 0) { sub_s.push_back(streams_[i]); sub_a.push_back(u_[i]); sub_i.push_back(i); }
        if (sub_s.empty()) return MSKF_OK;
        return mskf_ekf_update_batch(ctx_, (int)sub_s.size(), sub_s.data(), sub_a.data());
    };
    bool any = false;
    auto par = [&](int cnt, const std::function<void(int)> &fn) { if (pool_ekf_) pool_ekf_->run(cnt, fn); else for (int i = 0; i < cnt; ++i) fn(i); };
    par(n, [&](int i) { systems_[i]->msckfvio_ptr()->phaseA(msgs[i], u_[i], true); });
    for (int i = 0; i < n; ++i) any |= systems_[i]->msckfvio_ptr()->frameActive();
    if (any) {
        std::vector<int32_t> ns(n);
        std::vector<const mskf_imu_step *> sp(n);
        std::vector<const double *> jp(n);
        for (int i = 0; i < n; ++i) {
            MsckfVio &v = *systems_[i]->msckfvio_ptr();
            const bool act = v.frameActive();
            ns[i] = act ? (int)v.predictSteps().size() : 0;
            sp[i] = ns[i] ? v.predictSteps().data() : nullptr;
            jp[i] = act ? v.predictJ() : nullptr;
        }
        BR_CHK(mskf_ekf_predict_batch(ctx_, n, streams_.data(), ns.data(), sp.data(), jp.data()));
    }
    lap(PH_EKF_A);
    if (!any) return MSKF_OK;
    BR_CHK(run_updates());
    lap(PH_UPD1);
    par(n, [&](int i) { if (systems_[i]->msckfvio_ptr()->frameActive()) systems_[i]->msckfvio_ptr()->phaseB(u_[i]); else std::memset(&u_[i], 0, sizeof(u_[i])); });
    lap(PH_EKF_B);
    BR_CHK(run_updates());
    lap(PH_UPD2);
    {
        std::vector<int32_t> rm(2 * (size_t)n, -1);
        bool any_rm = false;
        par(n, [&](int i) {
            MsckfVio &v = *systems_[i]->msckfvio_ptr();
            v.phaseC(true);
            rm[2 * i] = v.pendingRemovals()[0]; rm[2 * i + 1] = v.pendingRemovals()[1];
        });
        for (int i = 0; i < n; ++i) any_rm |= rm[2 * i] >= 0;
        if (any_rm) BR_CHK(mskf_ekf_remove_clones_batch(ctx_, n, streams_.data(), rm.data()));
    }
    lap(PH_EKF_C);
    std::vector<double> pv(3 * (size_t)n);
    BR_CHK(mskf_ekf_get_pos_var_batch(ctx_, n, streams_.data(), pv.data()));
    for (int i = 0; i < n; ++i) systems_[i]->msckfvio_ptr()->phaseD(&pv[3 * i]);
    lap(PH_POSVAR);
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

int BatchGroup::run_pipelined(int first, int n_frames) {
    const int n = size();
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<FrameBatch>> queue;
    std::vector<std::unique_ptr<FrameBatch>> pool;      // consumed batches, reused by the producer
    bool producer_done = false;
    std::atomic<int> ekf_rc{MSKF_OK};
    std::string ekf_err;
    std::thread consumer([&]() {
        for (;;) {
            std::unique_ptr<FrameBatch> fb;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return !queue.empty() || producer_done; });
                if (queue.empty()) return;
                fb = std::move(queue.front());
                queue.pop_front();
            }
            cv.notify_all();
            if (ekf_rc.load() != MSKF_OK) continue;   // drain
            int rc = feed_imu(fb->frame, false, true);
            if (rc == MSKF_OK) rc = step_ekf(fb.get());
            if (rc != MSKF_OK) ekf_rc.store(rc);
            { std::lock_guard<std::mutex> lk(mu); pool.push_back(std::move(fb)); }
        }
    });
    int rc = MSKF_OK;
    for (int k = first; k < first + n_frames && rc == MSKF_OK && ekf_rc.load() == MSKF_OK; ++k) {
        rc = feed_imu(k, true, false);
        if (rc == MSKF_OK) rc = step_fe(p0_.data(), p1_.data(), seq[0].on_device, t_.data(), false);
        if (rc != MSKF_OK) break;
        // hand-off = a snapshot of every stream's message (live + stale entries + the first tail record).  The batches and
        // their per-stream messages are recycled through `pool` (at most 2 queued + 1 in the filter stage + 1 being
        // filled), so the steady state allocates nothing
        std::unique_ptr<FrameBatch> fb;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!pool.empty()) { fb = std::move(pool.back()); pool.pop_back(); }
        }
        if (!fb) fb.reset(new FrameBatch);
        fb->frame = k;
        fb->msg.resize(n); fb->tail_start.resize(n); fb->total.resize(n);
        for (int i = 0; i < n; ++i) {
            const ImageProcessor &ip = *systems_[i]->imgproc_ptr_;
            const CameraMeasurement &live = *ip.feature_msg_ptr_;
            const size_t total = live.features.size(), start = ip.zeroTailStart();
            const size_t keep = std::min(total, start + 1);
            if (!fb->msg[i] || fb->msg[i].use_count() > 1) fb->msg[i].reset(new CameraMeasurement);
            fb->msg[i]->time_stamp = live.time_stamp;
            fb->msg[i]->features.assign(live.features.begin(), live.features.begin() + keep);
            fb->tail_start[i] = start; fb->total[i] = total;
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return queue.size() < 2; });
            queue.push_back(std::move(fb));
        }
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); producer_done = true; }
    cv.notify_all();
    consumer.join();
    if (rc == MSKF_OK) rc = ekf_rc.load();
    return rc;
}

MultiRunner::MultiRunner(int device, int n_groups, int per_group, const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf,
                         int host_threads)
    : n_groups_(n_groups), per_group_(per_group), off_(n_groups, 0), next_(n_groups, 0) {
    for (int g = 0; g < n_groups; ++g) groups_.emplace_back(new BatchGroup(device, per_group, calib, fe, ekf, host_threads));
}

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
        const int from = off_[g] > 0 ? std::min(next_[g], first + off_[g]) : first;
        const int cnt = first + off_[g] + n - from;
        next_[g] = from + cnt;
        return pipelined ? groups_[g]->run_pipelined(from, cnt) : groups_[g]->run(from, cnt);
    };
    if (!threaded || n_groups_ == 1) {
        for (int g = 0; g < n_groups_; ++g) { rcs[g] = one(g); if (rcs[g] != MSKF_OK) return rcs[g]; }
        return MSKF_OK;
    }
    std::vector<std::thread> th;
    for (int g = 0; g < n_groups_; ++g) th.emplace_back([&, g]() { rcs[g] = one(g); });
    for (auto &t : th) t.join();
    for (int g = 0; g < n_groups_; ++g) if (rcs[g] != MSKF_OK) return rcs[g];
    return MSKF_OK;
}

}  // namespace cg
