// host_prof.h — coarse wall-clock accounting of the host-side bookkeeping (where the CPU time of a frame goes),
// summed over all streams and threads.  ~40 ns per scope; read through mskfh_get_hostprof (bench.py --host-prof).
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>

namespace cg {
namespace hostprof {
enum Slot {
    FE_TRACK_TAIL = 0, FE_OCCUPANCY, FE_DETECT, FE_SIEVE, FE_NEW_TAIL, FE_PRUNE, FE_PUBLISH, FE_PREPARE,
    EKF_IMU, EKF_AUGMENT, EKF_ADD_OBS, EKF_BUILD_LOST, EKF_APPLY1, EKF_ERASE_LOST, EKF_BUILD_PRUNE, EKF_TAIL_PRUNE, EKF_PUBLISH,
    N_SLOTS
};
inline const char *name(int s) {
    static const char *n[N_SLOTS] = {"fe_track_tail", "fe_occupancy", "fe_detect", "fe_sieve", "fe_new_tail", "fe_prune", "fe_publish",
                                     "fe_prepare", "ekf_imu", "ekf_augment", "ekf_add_obs", "ekf_build_lost", "ekf_apply1",
                                     "ekf_erase_lost", "ekf_build_prune", "ekf_tail_prune", "ekf_publish"};
    return (s >= 0 && s < N_SLOTS) ? n[s] : "";
}
inline std::atomic<uint64_t> *counters() { static std::atomic<uint64_t> c[N_SLOTS]; return c; }
// per-thread accounting gate (default on): a pipeline stage opens and closes its own measurement window
inline bool &enabled() { static thread_local bool e = true; return e; }
struct Scope {
    int slot; std::chrono::steady_clock::time_point t0;
    explicit Scope(int s) : slot(s), t0(std::chrono::steady_clock::now()) {}
    ~Scope() {
        if (!enabled()) return;
        const auto dt = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        counters()[slot].fetch_add((uint64_t)dt, std::memory_order_relaxed);
    }
};
}  // namespace hostprof
}  // namespace cg
