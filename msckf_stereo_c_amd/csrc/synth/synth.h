// synth.h — seeded EuRoC-shaped synthetic stereo + IMU stream generator.
//
// Test / bench input generator only: no part of the VIO pipeline.  Scene: a textured box room
// rendered through the radtan stereo calibration by inverse warp; trajectory: smooth periodic
// sinusoids with a static prefix (so MsckfVio::initializeGravityAndBias, msckf_vio.cpp:198, sees
// gravity only); IMU: analytic omega / specific force at 200 Hz + white noise + bias walk, values
// rounded through float32 like the reference's std::stof parsing
// (apps/run_euroc_single_thread.cpp:220,225).  Timestamps are integer nanoseconds
// converted the way the reference app does (:166, :192, :230).
//
// The image sequence is periodic after the static prefix: frame k >= n_static shows pose
// ((k - n_static) mod n_loop), so an arbitrarily long run needs only n_static + n_loop rendered
// stereo pairs (the trajectory and its velocity are continuous across the wrap).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../../include/mskf_types.h"

namespace synth {

struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed = 42u, uint64_t seq = 54u) {
        state = 0; inc = (seq << 1u) | 1u;
        next(); state += seed; next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((-rot) & 31));
    }
    double uniform() { return (next() >> 8) * (1.0 / 16777216.0); }
    double gauss() {  // Box-Muller
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-12) u1 = 1e-12;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

// Functions a pixel's value is made of: the same source is compiled for the host renderer (Stream::render) and for the HIP
// kernel of synth_render.hip (bench.py renders its sequences on the device); both with -ffp-contract=off, every operation
// a single IEEE double / integer operation, so the two produce identical bytes.
#if defined(__HIPCC__)
#define SYNTH_FN __host__ __device__ inline
#else
#define SYNTH_FN static inline
#endif
SYNTH_FN uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
SYNTH_FN uint32_t hash3(int32_t a, int32_t b, uint32_t c) {
    return hash32((uint32_t)a * 0x9E3779B1U ^ hash32((uint32_t)b * 0x85EBCA77U ^ hash32(c)));
}

// smooth value noise, one octave
SYNTH_FN double vnoise(double u, double v, uint32_t salt) {
    const double fu = floor(u), fv = floor(v);
    const int iu = (int)fu, iv = (int)fv;
    double a = u - fu, b = v - fv;
    a = a * a * (3 - 2 * a); b = b * b * (3 - 2 * b);
    const double h00 = (hash3(iu, iv, salt) & 0xFFFF) * (1.0 / 65535.0), h10 = (hash3(iu + 1, iv, salt) & 0xFFFF) * (1.0 / 65535.0);
    const double h01 = (hash3(iu, iv + 1, salt) & 0xFFFF) * (1.0 / 65535.0), h11 = (hash3(iu + 1, iv + 1, salt) & 0xFFFF) * (1.0 / 65535.0);
    return (h00 * (1 - a) + h10 * a) * (1 - b) + (h01 * (1 - a) + h11 * a) * b;
}
SYNTH_FN double texture(uint32_t seed, double u, double v, uint32_t plane) {
    const uint32_t s = seed * 31u + plane * 1013u;
    double t = 0.30 * vnoise(u / 0.64, v / 0.64, s + 1) + 0.30 * vnoise(u / 0.32, v / 0.32, s + 2) +
               0.45 * vnoise(u / 0.16, v / 0.16, s + 3) + 0.55 * vnoise(u / 0.08, v / 0.08, s + 4) +
               0.30 * vnoise(u / 0.04, v / 0.04, s + 5);
    t = (t - 0.95) * 2.2 + 0.5;  // contrast stretch around the mean
    // sparse high-contrast blobs (blocks) so a corner detector fires in every region
    const int bu = (int)floor(u / 0.24), bv = (int)floor(v / 0.24);
    const uint32_t hb = hash3(bu, bv, s + 9);
    if ((hb & 7u) == 0u) {
        const double cu = (bu + 0.5) * 0.24, cv = (bv + 0.5) * 0.24;
        if (fabs(u - cu) < 0.07 && fabs(v - cv) < 0.07) t = (hb & 8u) ? 0.95 : 0.05;
    }
    return t < 0 ? 0 : (t > 1 ? 1 : t);
}

// What one image of a sequence is rendered from: camera rotation (world <- camera) and centre of the frame's pose, the
// stream's seed, k2c = 2 * frame key + camera (the pixel-noise stream), and which camera's ray table its pixels read.
struct RenderImg { double R_wc[9]; double o[3]; double pixel_sigma; uint32_t seed; int32_t k2c; int32_t cam; int32_t pad; };

// value of pixel `idx` whose undistorted viewing ray is (rx, ry, 1): the textured box room seen along the ray, plus noise
SYNTH_FN uint8_t shade_pixel(const RenderImg &I, float rx, float ry, uint32_t idx) {
    const double cx = rx, cy = ry, cz = 1.0;
    const double rv[3] = {I.R_wc[0] * cx + I.R_wc[1] * cy + I.R_wc[2] * cz, I.R_wc[3] * cx + I.R_wc[4] * cy + I.R_wc[5] * cz,
                          I.R_wc[6] * cx + I.R_wc[7] * cy + I.R_wc[8] * cz};
    // box room: x in [-3, 5.5], y in [-4, 4], z in [-1.6, 2.4]
    const double lo[3] = {-3.0, -4.0, -1.6}, hi[3] = {5.5, 4.0, 2.4};
    double tbest = 1e30; int pbest = 0;
    for (int a = 0; a < 3; ++a) {
        if (rv[a] > 1e-9) { const double t = (hi[a] - I.o[a]) / rv[a]; if (t < tbest) { tbest = t; pbest = 2 * a; } }
        else if (rv[a] < -1e-9) { const double t = (lo[a] - I.o[a]) / rv[a]; if (t < tbest) { tbest = t; pbest = 2 * a + 1; } }
    }
    const double hx = I.o[0] + tbest * rv[0], hy = I.o[1] + tbest * rv[1], hz = I.o[2] + tbest * rv[2];
    double tu, tv;
    if (pbest < 2) { tu = hy; tv = hz; } else if (pbest < 4) { tu = hx; tv = hz; } else { tu = hx; tv = hy; }
    double val = 20.0 + 215.0 * texture(I.seed, tu, tv, (uint32_t)pbest);
    // pixel noise ~ N(0, sigma^2): sum of 4 uniforms (Irwin-Hall), deterministic in (seed, cam, frame, pixel)
    const uint32_t h = hash3((int32_t)idx, I.k2c, I.seed ^ 0xC0FFEEu);
    const double n4 = ((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + ((h >> 24) & 0xFF)) * (1.0 / 255.0) - 2.0;
    val += I.pixel_sigma * n4 * 1.7320508;  // var of sum of 4 U(0,1) = 1/3
    const int iv = (int)floor(val + 0.5);
    return (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

struct Cfg {
    uint32_t seed = 0x5EED0000u;
    int width = 752, height = 480;
    mskf_calib calib;
    int n_static = 25;       // frames with the rig at rest
    int n_loop = 200;        // frames per trajectory period (10 s at 20 Hz)
    int imu_per_frame = 10;  // 200 Hz IMU / 20 Hz camera
    int64_t t0_ns = 1403715273262142976LL;  // EuRoC V1_01-like epoch
    int64_t frame_dt_ns = 50000000LL;
    double sigma_gyro = 0.005, sigma_acc = 0.05, sigma_bg = 0.001, sigma_ba = 0.01;  // app_msckfvio.yaml:13-16
    double pixel_sigma = 2.0;
    double motion_scale = 1.0;
};

// scale the EuRoC calibration (config/camchain-imucam-euroc.yaml) to a W x H sensor
inline mskf_calib euroc_calib(int width, int height) {
    mskf_calib c;
    std::memset(&c, 0, sizeof(c));
    c.width = width; c.height = height;
    const double sx = width / 752.0, sy = height / 480.0;
    const double k0[4] = {458.654, 457.296, 367.215, 248.375};
    const double k1[4] = {457.587, 456.134, 379.999, 255.238};
    const double d0[4] = {-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05};
    const double d1[4] = {-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05};
    for (int i = 0; i < 4; ++i) {
        c.cam0_intrinsics[i] = k0[i] * ((i & 1) ? sy : sx);
        c.cam1_intrinsics[i] = k1[i] * ((i & 1) ? sy : sx);
        c.cam0_distortion[i] = d0[i];
        c.cam1_distortion[i] = d1[i];
    }
    c.cam0_model = c.cam1_model = MSKF_MODEL_RADTAN;
    const double Tci[16] = {0.014865542981794, 0.999557249008346, -0.025774436697440, 0.065222909535531,
                            -0.999880929698575, 0.014967213324719, 0.003756188357967, -0.020706385492719,
                            0.004140296794224, 0.025715529947966, 0.999660727177902, -0.008054602460030,
                            0, 0, 0, 1};
    const double T10[16] = {0.999997256477881, 0.002312067192424, 0.000376008102415, -0.110073808127187,
                            -0.002317135723281, 0.999898048506644, 0.014089835846648, 0.000399121547014,
                            -0.000343393120525, -0.014090668452714, 0.999900662637729, -0.000853702503357,
                            0, 0, 0, 1};
    const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(c.T_cam0_imu, Tci, sizeof(Tci));
    std::memcpy(c.T_cam1_cam0, T10, sizeof(T10));
    std::memcpy(c.T_imu_body, I4, sizeof(I4));
    return c;
}

struct Vec3 { double x, y, z; };
struct Mat3 { double m[9]; };
static inline Vec3 mul(const Mat3 &A, const Vec3 &v) {
    return Vec3{A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
                A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
static inline Mat3 mul(const Mat3 &A, const Mat3 &B) {
    Mat3 C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0; for (int k = 0; k < 3; ++k) s += A.m[3 * i + k] * B.m[3 * k + j];
        C.m[3 * i + j] = s;
    }
    return C;
}
static inline Mat3 transpose(const Mat3 &A) {
    Mat3 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * j + i];
    return C;
}

struct Pose { Mat3 R_wi; Vec3 p; };  // world <- imu rotation, imu position in world

class Stream {
  public:
    explicit Stream(const Cfg &cfg) : cfg_(cfg) {
        period_ = cfg.n_loop * (cfg.frame_dt_ns * 1e-9);
        gen_imu_noise_seed_ = cfg.seed ^ 0xA5A5A5A5u;
    }
    const Cfg &cfg() const { return cfg_; }

    // ---- timestamps, reference app conversion (apps/run_euroc_single_thread.cpp:164-166,192)
    static double ns_to_sec(int64_t ns) {
        const int64_t sec = ns / 1000000000LL, nsec = ns % 1000000000LL;
        const double stamp_ns = (double)(int)sec * 1e9 + (double)(int)nsec;
        return stamp_ns * 1e-9;
    }
    int64_t frame_ns(int k) const { return cfg_.t0_ns + (int64_t)k * cfg_.frame_dt_ns; }
    double frame_time(int k) const { return ns_to_sec(frame_ns(k)); }
    int64_t imu_ns(int j) const { return cfg_.t0_ns + (int64_t)j * (cfg_.frame_dt_ns / cfg_.imu_per_frame); }

    // motion phase time tau (s) for absolute sample time offset (s since t0)
    double tau_of(double t_rel) const {
        const double ts = cfg_.n_static * (cfg_.frame_dt_ns * 1e-9);
        return t_rel <= ts ? 0.0 : (t_rel - ts);
    }

    Pose pose_at_tau(double tau) const {
        const double w = 6.283185307179586 / period_;
        const double s = cfg_.motion_scale;
        const double c1 = std::cos(w * tau), s1 = std::sin(w * tau), c2 = std::cos(2 * w * tau), s2 = std::sin(2 * w * tau);
        const double c3 = std::cos(3 * w * tau), s3 = std::sin(3 * w * tau);
        Pose P;
        P.p = Vec3{s * (0.45 * (1 - c1) + 0.12 * (1 - c2)), s * (0.50 * (s1 - 0.5 * s2)), s * (0.18 * (1 - c2) + 0.08 * (1 - c3))};
        const double roll = s * 0.10 * (s1 - s3 / 3.0), pitch = s * 0.12 * (1 - c2) * 0.5 - s * 0.05 * (1 - c1),
                     yaw = s * 0.30 * (s1 - 0.5 * s2);
        // world <- imu: R0 * Rz(yaw) Ry(pitch) Rx(roll) expressed about world axes applied to the nominal attitude
        const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
        Mat3 Rx{{1, 0, 0, 0, cr, -sr, 0, sr, cr}}, Ry{{cp, 0, sp, 0, 1, 0, -sp, 0, cp}}, Rz{{cy, -sy, 0, sy, cy, 0, 0, 0, 1}};
        // nominal attitude: imu x = world up, imu z = world +x (camera looks along world +x), imu y = world -y
        Mat3 R0{{0, 0, 1, 0, -1, 0, 1, 0, 0}};
        P.R_wi = mul(mul(Rz, mul(Ry, Rx)), R0);
        return P;
    }
    Pose pose_of_frame(int k) const {
        if (k < cfg_.n_static) return pose_at_tau(0.0);
        const int kk = (k - cfg_.n_static) % cfg_.n_loop;
        return pose_at_tau(kk * (cfg_.frame_dt_ns * 1e-9));
    }
    // body(=imu, T_imu_body = I) pose at frame k for ground truth: position + Hamilton quaternion of R_wi
    mskf_pose gt_pose(int k) const {
        const double ts = cfg_.n_static * (cfg_.frame_dt_ns * 1e-9);
        (void)ts;
        Pose P = (k < cfg_.n_static) ? pose_at_tau(0.0) : pose_at_tau((k - cfg_.n_static) * (cfg_.frame_dt_ns * 1e-9));
        mskf_pose o;
        o.time_stamp = frame_time(k);
        o.p[0] = P.p.x; o.p[1] = P.p.y; o.p[2] = P.p.z;
        rot_to_quat(P.R_wi, o.q);
        return o;
    }

    // ---- IMU sample j (absolute index), deterministic given (seed, j): noise drawn from a per-sample hash stream
    mskf_imu_sample imu_sample(int j) {
        const double dt = (cfg_.frame_dt_ns / cfg_.imu_per_frame) * 1e-9;
        const double t_rel = j * dt;
        const double tau = tau_of(t_rel);
        mskf_imu_sample s;
        s.time_stamp = ns_to_sec(imu_ns(j));
        Vec3 w{0, 0, 0}, a_w{0, 0, 0};
        Pose P = pose_at_tau(tau);
        if (tau > 0) {
            const double h = 1e-4;
            Pose Pm = pose_at_tau(tau - h), Pp = pose_at_tau(tau + h);
            // omega_body from R(t-h)^T R(t+h) ~ I + 2h [w]x
            Mat3 dR = mul(transpose(Pm.R_wi), Pp.R_wi);
            w = Vec3{(dR.m[7] - dR.m[5]) / (4 * h), (dR.m[2] - dR.m[6]) / (4 * h), (dR.m[3] - dR.m[1]) / (4 * h)};
            a_w = Vec3{(Pp.p.x - 2 * P.p.x + Pm.p.x) / (h * h), (Pp.p.y - 2 * P.p.y + Pm.p.y) / (h * h), (Pp.p.z - 2 * P.p.z + Pm.p.z) / (h * h)};
        }
        // specific force in body: R^T (a_w - g), g = (0,0,-9.81)
        Vec3 f = mul(transpose(P.R_wi), Vec3{a_w.x, a_w.y, a_w.z + 9.81});
        // bias walk: advance lazily and cache
        while ((int)bias_.size() <= j) {
            const int n = (int)bias_.size();
            Bias b = n ? bias_.back() : Bias{{0.002, -0.001, 0.0015}, {0.02, -0.015, 0.01}};
            Pcg32 r(cfg_.seed * 2654435761u + (uint32_t)n, 77);
            const double sq = std::sqrt(dt);
            for (int i = 0; i < 3; ++i) { b.g[i] += cfg_.sigma_bg * sq * r.gauss(); b.a[i] += cfg_.sigma_ba * sq * r.gauss(); }
            bias_.push_back(b);
        }
        Pcg32 r(gen_imu_noise_seed_ * 40503u + (uint32_t)j, 99);
        const double wv[3] = {w.x, w.y, w.z}, fv[3] = {f.x, f.y, f.z};
        for (int i = 0; i < 3; ++i) {
            const double g = wv[i] + bias_[j].g[i] + cfg_.sigma_gyro * r.gauss();
            const double a = fv[i] + bias_[j].a[i] + cfg_.sigma_acc * r.gauss();
            s.angular_velocity[i] = (double)(float)g;      // Q9: std::stof
            s.linear_acceleration[i] = (double)(float)a;
        }
        return s;
    }

    // what the image of camera c at frame k is rendered from (also the input of the device renderer, synth_render.hip)
    RenderImg render_params(const Pose &P, int c, int k) const {
        Mat3 R_cw; Vec3 o;
        cam_pose(P, c, R_cw, o);
        const Mat3 R_wc = transpose(R_cw);
        RenderImg I;
        for (int i = 0; i < 9; ++i) I.R_wc[i] = R_wc.m[i];
        I.o[0] = o.x; I.o[1] = o.y; I.o[2] = o.z;
        I.pixel_sigma = cfg_.pixel_sigma; I.seed = cfg_.seed; I.k2c = k * 2 + c; I.cam = c; I.pad = 0;
        return I;
    }
    RenderImg render_params(int k, int c) const { return render_params(pose_of_frame(k), c, k); }
    const float *ray_table(int c) const { need_rays(); return rays_[c].data(); }      // width x height x (x, y): undistorted viewing rays

    // ---- render stereo pair for frame k (row-major u8, pitch = width)
    void render(int k, uint8_t *cam0, uint8_t *cam1) const {
        Pose P = pose_of_frame(k);
        render_cam(P, 0, k, cam0);
        render_cam(P, 1, k, cam1);
    }

    // project a world point into cam `c` (distorted pixels); returns false if behind the camera
    bool project(const Pose &P, int c, const Vec3 &pw, double &u, double &v) const {
        Mat3 R_cw; Vec3 t_wc;
        cam_pose(P, c, R_cw, t_wc);
        Vec3 d{pw.x - t_wc.x, pw.y - t_wc.y, pw.z - t_wc.z};
        Vec3 pc = mul(R_cw, d);
        if (pc.z <= 0.05) return false;
        const double *K = c ? cfg_.calib.cam1_intrinsics : cfg_.calib.cam0_intrinsics;
        const double *D = c ? cfg_.calib.cam1_distortion : cfg_.calib.cam0_distortion;
        const double x = pc.x / pc.z, y = pc.y / pc.z, r2 = x * x + y * y;
        const double cd = 1 + D[0] * r2 + D[1] * r2 * r2;
        const double xd = x * cd + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x);
        const double yd = y * cd + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
        u = K[0] * xd + K[2]; v = K[1] * yd + K[3];
        return true;
    }

    static void rot_to_quat(const Mat3 &R, double q[4]) {  // Hamilton x y z w
        const double *m = R.m;
        double t = m[0] + m[4] + m[8];
        if (t > 0) {
            double s = std::sqrt(t + 1.0) * 2;
            q[3] = 0.25 * s; q[0] = (m[7] - m[5]) / s; q[1] = (m[2] - m[6]) / s; q[2] = (m[3] - m[1]) / s;
        } else if (m[0] > m[4] && m[0] > m[8]) {
            double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
            q[3] = (m[7] - m[5]) / s; q[0] = 0.25 * s; q[1] = (m[1] + m[3]) / s; q[2] = (m[2] + m[6]) / s;
        } else if (m[4] > m[8]) {
            double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
            q[3] = (m[2] - m[6]) / s; q[0] = (m[1] + m[3]) / s; q[1] = 0.25 * s; q[2] = (m[5] + m[7]) / s;
        } else {
            double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
            q[3] = (m[3] - m[1]) / s; q[0] = (m[2] + m[6]) / s; q[1] = (m[5] + m[7]) / s; q[2] = 0.25 * s;
        }
    }

  private:
    struct Bias { double g[3], a[3]; };

    // world <- cam c rotation transposed (R_cw maps world to cam) and camera centre in world
    void cam_pose(const Pose &P, int c, Mat3 &R_cw, Vec3 &t_wc) const {
        // T_cam0_imu: p_cam0 = R_ci p_imu + t_ci
        Mat3 R_ci; Vec3 t_ci;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R_ci.m[3 * i + j] = cfg_.calib.T_cam0_imu[4 * i + j];
        t_ci = Vec3{cfg_.calib.T_cam0_imu[3], cfg_.calib.T_cam0_imu[7], cfg_.calib.T_cam0_imu[11]};
        if (c == 1) {
            Mat3 R10; Vec3 t10;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R10.m[3 * i + j] = cfg_.calib.T_cam1_cam0[4 * i + j];
            t10 = Vec3{cfg_.calib.T_cam1_cam0[3], cfg_.calib.T_cam1_cam0[7], cfg_.calib.T_cam1_cam0[11]};
            Vec3 rt = mul(R10, t_ci);
            t_ci = Vec3{rt.x + t10.x, rt.y + t10.y, rt.z + t10.z};
            R_ci = mul(R10, R_ci);
        }
        // p_cam = R_ci R_iw (p_w - p) + t_ci  => R_cw = R_ci R_wi^T ; centre: p_w = p - R_wi R_ci^T t_ci
        R_cw = mul(R_ci, transpose(P.R_wi));
        Vec3 ci = mul(transpose(R_ci), t_ci);
        Vec3 cw = mul(P.R_wi, ci);
        t_wc = Vec3{P.p.x - cw.x, P.p.y - cw.y, P.p.z - cw.z};
    }

    // the ray tables are built on first use (a generator that only feeds IMU samples, or whose images are rendered on the device
    // from another generator's tables - they depend on the calibration alone - never needs them: 0.2 s per generator at 752 x 480)
    void need_rays() const { std::call_once(rays_once_, [this]() { const_cast<Stream *>(this)->build_ray_tables(); }); }
    void build_ray_tables() {
        for (int c = 0; c < 2; ++c) {
            const double *K = c ? cfg_.calib.cam1_intrinsics : cfg_.calib.cam0_intrinsics;
            const double *D = c ? cfg_.calib.cam1_distortion : cfg_.calib.cam0_distortion;
            std::vector<float> &tab = rays_[c];
            tab.resize((size_t)cfg_.width * cfg_.height * 2);
            for (int v = 0; v < cfg_.height; ++v)
                for (int u = 0; u < cfg_.width; ++u) {
                    double x0 = (u - K[2]) / K[0], y0 = (v - K[3]) / K[1], x = x0, y = y0;
                    for (int it = 0; it < 8; ++it) {
                        const double r2 = x * x + y * y;
                        const double icd = 1.0 / (1.0 + (D[1] * r2 + D[0]) * r2);
                        const double dx = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x);
                        const double dy = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
                        x = (x0 - dx) * icd; y = (y0 - dy) * icd;
                    }
                    tab[2 * ((size_t)v * cfg_.width + u)] = (float)x;
                    tab[2 * ((size_t)v * cfg_.width + u) + 1] = (float)y;
                }
        }
    }

    void render_cam(const Pose &P, int c, int k, uint8_t *out) const {
        const RenderImg I = render_params(P, c, k);
        need_rays();
        const std::vector<float> &tab = rays_[c];
        const size_t n = (size_t)cfg_.width * cfg_.height;
        for (size_t idx = 0; idx < n; ++idx) out[idx] = shade_pixel(I, tab[2 * idx], tab[2 * idx + 1], (uint32_t)idx);
    }

    Cfg cfg_;
    double period_;
    std::vector<float> rays_[2];
    mutable std::once_flag rays_once_;
    std::vector<Bias> bias_;
    uint32_t gen_imu_noise_seed_;
};

}  // namespace synth
