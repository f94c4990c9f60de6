// host_capi.cpp — C entry points over the host mirror classes (cg::System, MultiRunner) for the Python
// tests and bench.py.  This is host glue above the C-ABI of include/mskf_hip.h, not part of it.
#include <cstring>
#include "batch_runner.h"
#include "host_prof.h"

using namespace cg;

extern "C" {

void *mskfh_runner_create(int device, int n_groups, int per_group, const mskf_calib *calib, const mskf_fe_cfg *fe, const mskf_ekf_cfg *ekf,
                          int host_threads, int ekf_host_threads, int halves) {
    MultiRunner *r = new MultiRunner(device, n_groups, per_group, *calib, *fe, *ekf, host_threads, ekf_host_threads, halves);
    if (!r->ok()) {
        std::fprintf(stderr, "mskfh_runner_create: %s\n", r->error().c_str());
        delete r;
        return nullptr;
    }
    return r;
}
// parse the three YAML files the reference reads (Q16 paths relative to config_dir); 0 on success
int mskfh_load_configs(const char *config_dir, mskf_calib *calib, mskf_fe_cfg *fe, mskf_ekf_cfg *ekf) {
    try {
        const std::string d(config_dir);
        *calib = calib_from_yaml(YAML::LoadFile(d + "/camchain-imucam-euroc.yaml"));
        *fe = fe_cfg_from_yaml(YAML::LoadFile(d + "/app_imgproc.yaml"));
        *ekf = ekf_cfg_from_yaml(YAML::LoadFile(d + "/app_msckfvio.yaml"));
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "mskfh_load_configs: %s\n", e.what());
        return -1;
    }
}

// where the front-end's bookkeeping runs (ImageProcessor::setFeBooksOnHost): tests compare the two paths
void mskfh_set_fe_books_on_host(int on) { ImageProcessor::setFeBooksOnHost(on); }
void mskfh_runner_destroy(void *h) { delete (MultiRunner *)h; }
int mskfh_runner_num_streams(void *h) { return ((MultiRunner *)h)->n_streams(); }
const char *mskfh_runner_error(void *h) { static thread_local std::string e; e = ((MultiRunner *)h)->error(); return e.c_str(); }

void mskfh_runner_imu(void *h, int stream, const mskf_imu_sample *s) { ((MultiRunner *)h)->imu(stream, *s); }
int mskfh_runner_step(void *h, const uint8_t *const *cam0, const uint8_t *const *cam1, int on_device, const double *t) {
    return ((MultiRunner *)h)->step(cam0, cam1, on_device, t);
}
void mskfh_runner_set_sequence(void *h, int stream, const uint8_t *cam0_base, const uint8_t *cam1_base, int on_device, size_t frame_bytes,
                               int n_static, int n_loop, long long t0_ns, long long frame_dt_ns, const mskf_imu_sample *imu, int n_imu) {
    StreamSequence &q = ((MultiRunner *)h)->sequence(stream);
    q.cam0_base = cam0_base; q.cam1_base = cam1_base; q.on_device = on_device; q.frame_bytes = frame_bytes;
    q.n_static = n_static; q.n_loop = n_loop; q.t0_ns = t0_ns; q.frame_dt_ns = frame_dt_ns; q.imu = imu; q.n_imu = n_imu;
    q.imu_cursor = 0;
}
// threaded: one host thread per group; pipelined: front-end and filter of each group run as a two-stage pipeline
int mskfh_runner_run(void *h, int first, int n, int threaded, int pipelined) { return ((MultiRunner *)h)->run(first, n, threaded != 0, pipelined != 0); }
// `warmup` + `steps` frames of every group in one pipelined run; *elapsed_s covers exactly the `steps` frames (MultiRunner::run_timed)
int mskfh_runner_run_timed(void *h, int first, int warmup, int steps, int max_extra, double *elapsed_s) {
    return ((MultiRunner *)h)->run_timed(first, warmup, steps, max_extra, elapsed_s);
}
int mskfh_runner_frames_done(void *h, int g) { return ((MultiRunner *)h)->frames_done(g); }
// group g's stages in the last timed window: [0] front-end open, [1] front-end close, [2] filter open, [3] filter close (steady
// clock, s), [4] frames the front-end started inside, [5] frames the filter started inside, [6] frames of the run the batch had
// completed when the shared window closed (balanced runner; 0 otherwise)
void mskfh_runner_window(void *h, int g, double out[7]) {
    const TimedWindow &w = ((MultiRunner *)h)->window(g);
    out[0] = w.t_fe_begin; out[1] = w.t_fe_end; out[2] = w.t_ekf_begin; out[3] = w.t_ekf_end; out[4] = w.fe_frames; out[5] = w.ekf_frames;
    out[6] = w.frames_at_close;
}
// wall seconds per phase inside the last timed window, summed over groups (each stage between its own marks)
void mskfh_runner_get_window_phases(void *h, double *out) {
    MultiRunner *r = (MultiRunner *)h;
    for (int k = 0; k < BatchGroup::PH_COUNT; ++k) out[k] = 0;
    for (int g = 0; g < r->n_groups(); ++g)
        for (int k = 0; k < BatchGroup::PH_COUNT; ++k) out[k] += r->group(g).window_phase_s[k];
}
void mskfh_runner_get_window_phases_group(void *h, int g, double *out) {
    MultiRunner *r = (MultiRunner *)h;
    for (int k = 0; k < BatchGroup::PH_COUNT; ++k) out[k] = r->group(g).window_phase_s[k];
}
// state of local stream 0 of group g when its stages closed the window; returns the feature count (-1: no window closed)
int mskfh_runner_mark_dump_size(void *h, int g) {
    const BatchGroup::MarkDump &m = ((MultiRunner *)h)->group(g).mark_dump;
    return (m.fe_valid && m.ekf_valid) ? (int)m.ids.size() : -1;
}
void mskfh_runner_mark_dump(void *h, int g, uint64_t *ids, int32_t *lifetime, mskf_point2f *cam0, mskf_point2f *cam1, double *imu28) {
    const BatchGroup::MarkDump &m = ((MultiRunner *)h)->group(g).mark_dump;
    for (size_t i = 0; i < m.ids.size(); ++i) {
        ids[i] = m.ids[i]; lifetime[i] = m.life[i];
        cam0[i] = mskf_point2f{m.c0[i].x, m.c0[i].y}; cam1[i] = mskf_point2f{m.c1[i].x, m.c1[i].y};
    }
    std::memcpy(imu28, m.imu, sizeof(m.imu));
}
void mskfh_runner_keep_trajectory_stream(void *h, int stream, int keep) { ((MultiRunner *)h)->system(stream).msckfvio_ptr()->keepTrajectory = keep != 0; }
void mskfh_runner_set_stagger(void *h, int delta) { ((MultiRunner *)h)->set_stagger(delta); }
// QR compression of every stream's stacked Jacobian from the next update on (mskf_ekf_set_compression_mode); call between runs
int mskfh_runner_set_compression(void *h, int mode) {
    MultiRunner *r = (MultiRunner *)h;
    for (int i = 0; i < r->n_streams(); ++i) { const int rc = mskf_ekf_set_compression_mode(r->system(i).stream(), mode); if (rc != MSKF_OK) return rc; }
    return MSKF_OK;
}
void mskfh_runner_set_workers(void *h, int fe_workers, int ekf_workers) { ((MultiRunner *)h)->set_workers(fe_workers, ekf_workers); }
int mskfh_runner_group_offset(void *h, int g) { return ((MultiRunner *)h)->group_offset(g); }
void mskfh_runner_keep_trajectory(void *h, int keep) {
    MultiRunner *r = (MultiRunner *)h;
    for (int i = 0; i < r->n_streams(); ++i) r->system(i).msckfvio_ptr()->keepTrajectory = keep != 0;
}

void mskfh_runner_set_timing(void *h, int enable) {
    MultiRunner *r = (MultiRunner *)h;
    for (int g = 0; g < r->n_groups(); ++g)
        for (int h = 0; h < r->group(g).n_halves(); ++h) { mskf_ctx_set_timing(r->group(g).ctx(h), enable); mskf_ctx_set_timing(r->group(g).ekf_ctx(h), enable); }
}
// sums over groups; arrays of MSKF_K_COUNT
void mskfh_runner_get_timing(void *h, double *ms, long long *launches, long long *units, int reset) {
    MultiRunner *r = (MultiRunner *)h;
    for (int k = 0; k < MSKF_K_COUNT; ++k) { ms[k] = 0; launches[k] = 0; units[k] = 0; }
    for (int g = 0; g < r->n_groups(); ++g)
        for (int h = 0; h < r->group(g).n_halves(); ++h) {
            mskf_ctx *cs[2] = {r->group(g).ctx(h), r->group(g).ekf_ctx(h)};
            for (mskf_ctx *c : cs) {
                double m[MSKF_K_COUNT]; long long l[MSKF_K_COUNT], u[MSKF_K_COUNT];
                if (mskf_ctx_get_timing(c, m, l, u, reset) != MSKF_OK) continue;
                for (int k = 0; k < MSKF_K_COUNT; ++k) { ms[k] += m[k]; launches[k] += l[k]; units[k] += u[k]; }
            }
        }
}

// host seconds inside the batched C-ABI calls (packing / unpacking around the device work), summed over groups:
// [0] update pack, [1] update unpack, [2] track pack, [3] track unpack
void mskfh_runner_get_abi_host_time(void *h, double *out, int reset) {
    MultiRunner *r = (MultiRunner *)h;
    for (int k = 0; k < 4; ++k) out[k] = 0;
    for (int g = 0; g < r->n_groups(); ++g)
        for (int h = 0; h < r->group(g).n_halves(); ++h) {
            mskf_ctx *cs[2] = {r->group(g).ctx(h), r->group(g).ekf_ctx(h)};
            for (mskf_ctx *c : cs) { double t[4]; if (mskf_ctx_get_host_time(c, t, reset) == MSKF_OK) for (int k = 0; k < 4; ++k) out[k] += t[k]; }
        }
}

// wall seconds per step() phase summed over groups (BatchGroup::PH_*), optionally reset
void mskfh_runner_get_phases(void *h, double *out, int reset) {
    MultiRunner *r = (MultiRunner *)h;
    for (int k = 0; k < BatchGroup::PH_COUNT; ++k) out[k] = 0;
    for (int g = 0; g < r->n_groups(); ++g)
        for (int k = 0; k < BatchGroup::PH_COUNT; ++k) { out[k] += r->group(g).phase_s[k]; if (reset) r->group(g).phase_s[k] = 0; }
}

// host bookkeeping accounting (host_prof.h): seconds per slot summed over all streams/threads; returns the slot count
int mskfh_get_hostprof(double *out, int capacity, int reset) {
    for (int k = 0; k < cg::hostprof::N_SLOTS && k < capacity; ++k) {
        out[k] = (double)cg::hostprof::counters()[k].load() * 1e-9;
        if (reset) cg::hostprof::counters()[k].store(0);
    }
    return cg::hostprof::N_SLOTS;
}
const char *mskfh_hostprof_name(int slot) { return cg::hostprof::name(slot); }

// ---- per-stream inspection
// ImageProcessor::twoPointRansac on undistorted point pairs (host arithmetic only, no device involved)
void mskfh_two_point_ransac(int n, const mskf_point2f *pts1_und, const mskf_point2f *pts2_und, const double *R_p_c, const double *intrinsics,
                            double inlier_error, double success_probability, unsigned long long *draws, int32_t *markers) {
    std::vector<cg::Point2f> a(n), b(n);
    for (int i = 0; i < n; ++i) { a[i] = cg::Point2f(pts1_und[i].x, pts1_und[i].y); b[i] = cg::Point2f(pts2_und[i].x, pts2_und[i].y); }
    hm::Mat3 R;
    for (int i = 0; i < 9; ++i) R.m[i] = R_p_c[i];
    std::vector<int> m;
    cg::two_point_ransac(a, b, R, intrinsics, inlier_error, success_probability, *draws, m);
    for (int i = 0; i < n; ++i) markers[i] = m[i];
}

int mskfh_num_features(void *h, int stream) {
    std::vector<ImageProcessor::FeatureIDType> ids; std::vector<int> life; std::vector<Point2f> a, b;
    ((MultiRunner *)h)->system(stream).imgproc_ptr_->dumpCurrent(ids, life, a, b);
    return (int)ids.size();
}
void mskfh_get_dump(void *h, int stream, uint64_t *ids, int32_t *lifetime, mskf_point2f *cam0, mskf_point2f *cam1, mskf_tracking_info *info) {
    std::vector<ImageProcessor::FeatureIDType> id; std::vector<int> life; std::vector<Point2f> a, b;
    ImageProcessor &ip = *((MultiRunner *)h)->system(stream).imgproc_ptr_;
    ip.dumpCurrent(id, life, a, b);
    for (size_t i = 0; i < id.size(); ++i) {
        ids[i] = id[i]; lifetime[i] = life[i];
        cam0[i] = mskf_point2f{a[i].x, a[i].y}; cam1[i] = mskf_point2f{b[i].x, b[i].y};
    }
    info->time_stamp = ip.last_tracking_info.time_stamp;
    info->before_tracking = ip.last_tracking_info.before_tracking; info->after_tracking = ip.last_tracking_info.after_tracking;
    info->after_matching = ip.last_tracking_info.after_matching; info->after_ransac = ip.last_tracking_info.after_ransac;
}
// the message as the reference holds it: with the whole Q1 tail of never-written records (kept as a count by the batch runner)
int mskfh_msg_size(void *h, int stream) { return (int)((MultiRunner *)h)->system(stream).imgproc_ptr_->messageSize(); }
void mskfh_get_msg(void *h, int stream, mskf_feature_meas *out) {
    const ImageProcessor &ip = *((MultiRunner *)h)->system(stream).imgproc_ptr_;
    const auto &f = ip.feature_msg_ptr_->features;
    static_assert(sizeof(FeatureMeasurement) == sizeof(mskf_feature_meas), "record layout");
    std::memset(out, 0, ip.messageSize() * sizeof(mskf_feature_meas));
    std::memcpy(out, f.data(), f.size() * sizeof(mskf_feature_meas));
}
int mskfh_num_poses(void *h, int stream) { return (int)((MultiRunner *)h)->system(stream).msckfvio_ptr()->poses().size(); }
void mskfh_get_poses(void *h, int stream, mskf_pose *out) {
    const auto &p = ((MultiRunner *)h)->system(stream).msckfvio_ptr()->poses();
    std::memcpy(out, p.data(), p.size() * sizeof(mskf_pose));
}
int mskfh_state_dim(void *h, int stream) { int d = 0; mskf_ekf_get_dim(((MultiRunner *)h)->system(stream).stream(), &d); return d; }
int mskfh_get_cov(void *h, int stream, double *out, int cap) { return mskf_ekf_get_cov(((MultiRunner *)h)->system(stream).stream(), out, cap); }
void mskfh_get_imu_state(void *h, int stream, double *out) {   // same 28-double layout as the oracle's getter
    const IMUState &s = ((MultiRunner *)h)->system(stream).msckfvio_ptr()->imuState();
    int k = 0;
    for (int i = 0; i < 4; ++i) out[k++] = s.orientation.q[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.position[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.velocity[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.gyro_bias[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.acc_bias[i];
    for (int i = 0; i < 9; ++i) out[k++] = s.R_imu_cam0.m[i];
    for (int i = 0; i < 3; ++i) out[k++] = s.t_cam0_imu[i];
}
long long mskfh_num_device_frames(void *h, int stream) { return ((MultiRunner *)h)->system(stream).imgproc_ptr_->deviceFrames(); }
int mskfh_num_updates(void *h, int stream) { return ((MultiRunner *)h)->system(stream).msckfvio_ptr()->numUpdates(); }
int mskfh_num_tsqr_updates(void *h, int stream) { return ((MultiRunner *)h)->system(stream).msckfvio_ptr()->numTsqrUpdates(); }
int mskfh_num_uncompressed_updates(void *h, int stream) { return ((MultiRunner *)h)->system(stream).msckfvio_ptr()->numUncompressedUpdates(); }
long long mskfh_stacked_rows(void *h, int stream) { return ((MultiRunner *)h)->system(stream).msckfvio_ptr()->stackedRows(); }
long long mskfh_num_resets(void *h, int stream) { return ((MultiRunner *)h)->system(stream).msckfvio_ptr()->numResets(); }
int mskfh_num_clones(void *h, int stream) { return ((MultiRunner *)h)->system(stream).msckfvio_ptr()->numClones(); }
void *mskfh_group_hip_stream(void *h, int stream) { int l; return mskf_ctx_hip_stream(((MultiRunner *)h)->group_of(stream, l).ctx()); }

}  // extern "C"
