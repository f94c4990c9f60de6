"""Python driver of the host mirror classes (cg::System / MultiRunner, csrc/host) — the product path.

Everything below the binding is C++ above the C-ABI and HIP kernels below it; nothing here computes.
"""
import ctypes as C
import os

import numpy as np

from .capi import MskfError
from .ctypes_types import FEATURE_MEAS, POINT2F, POSE, Calib, EkfCfg, FeCfg, ImuSample, TrackingInfo

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

IMU_SAMPLE = np.dtype([("t", "<f8"), ("w", "<f8", 3), ("a", "<f8", 3)])


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "_build", "libmskf_host.so")
        if not os.path.exists(p):
            raise MskfError("libmskf_host.so is not built (run python -m msckf_stereo_c_amd.build); there is no CPU fallback")
        L = C.CDLL(p)
        L.mskfh_runner_create.restype = C.c_void_p
        L.mskfh_runner_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Calib), C.POINTER(FeCfg), C.POINTER(EkfCfg), C.c_int, C.c_int, C.c_int]
        L.mskfh_runner_set_workers.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mskfh_runner_set_compression.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_destroy.argtypes = [C.c_void_p]
        L.mskfh_runner_error.restype = C.c_char_p
        L.mskfh_runner_error.argtypes = [C.c_void_p]
        L.mskfh_runner_imu.argtypes = [C.c_void_p, C.c_int, C.POINTER(ImuSample)]
        L.mskfh_runner_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mskfh_runner_set_sequence.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                                C.c_longlong, C.c_longlong, C.c_void_p, C.c_int]
        L.mskfh_runner_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mskfh_runner_keep_trajectory.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_keep_trajectory_stream.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mskfh_runner_run_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.mskfh_runner_frames_done.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_window.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mskfh_runner_get_window_phases.argtypes = [C.c_void_p, C.c_void_p]
        L.mskfh_runner_mark_dump_size.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_mark_dump.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.mskfh_runner_set_stagger.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_group_offset.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_group_offset.restype = C.c_int
        for name in ("mskfh_num_features", "mskfh_msg_size", "mskfh_num_poses", "mskfh_state_dim", "mskfh_num_updates",
                     "mskfh_num_clones"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            getattr(L, name).restype = C.c_int
        L.mskfh_num_tsqr_updates.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_num_tsqr_updates.restype = C.c_int
        L.mskfh_num_uncompressed_updates.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_num_uncompressed_updates.restype = C.c_int
        L.mskfh_stacked_rows.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_stacked_rows.restype = C.c_longlong
        L.mskfh_num_resets.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_num_resets.restype = C.c_longlong
        L.mskfh_num_device_frames.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_num_device_frames.restype = C.c_longlong
        L.mskfh_get_dump.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.POINTER(TrackingInfo)]
        L.mskfh_get_msg.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mskfh_get_poses.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mskfh_get_cov.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.mskfh_get_imu_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mskfh_runner_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_runner_get_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mskfh_runner_get_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mskfh_group_hip_stream.argtypes = [C.c_void_p, C.c_int]
        L.mskfh_group_hip_stream.restype = C.c_void_p
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_fe_books_on_host(on):
    """Process-wide: 1 = the front-end's bookkeeping stays on the host for every frame (the phased mskf_fe_track path), 0 = whole
    frames on the device wherever the configuration allows (default), -1 = back to the MSKF_FE_BOOKS environment variable."""
    lib().mskfh_set_fe_books_on_host(int(on))


class Runner:
    """n_groups x per_group independent VIO streams on one GPU."""

    def __init__(self, calib, fe_cfg, ekf_cfg, n_groups=1, per_group=1, device=0, host_threads=1, ekf_host_threads=0, halves=1):
        """host_threads / ekf_host_threads: threads sharing the per-stream host phases of a group's front-end / filter stage
        (0 = as host_threads); halves = 2: two staggered half-batches per stage (BatchGroup)."""
        self.L = lib()
        self.calib, self.fe_cfg, self.ekf_cfg = calib, fe_cfg, ekf_cfg
        self.n = n_groups * per_group
        self.n_groups, self.per_group = n_groups, per_group
        self.h = self.L.mskfh_runner_create(device, n_groups, per_group, C.byref(calib), C.byref(fe_cfg), C.byref(ekf_cfg), host_threads, ekf_host_threads, halves)
        if not self.h:
            raise MskfError("could not create the runner (no GPU / HIP library?): see stderr")
        self._keep = []

    def close(self):
        if self.h:
            self.L.mskfh_runner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MskfError("runner status %d: %s" % (rc, self.L.mskfh_runner_error(self.h).decode()))

    def set_compression(self, mode):
        """QR compression of every stream from its next update on: 0 auto, 1 Gram + Cholesky, 2 Householder TSQR (call between runs)."""
        self._chk(self.L.mskfh_runner_set_compression(self.h, int(mode)))

    def set_workers(self, fe_workers=0, ekf_workers=0):
        """Workers of the balanced runner (MultiRunner::run_balanced): front-end / filter workers that serve the n_groups
        batches; 0 = one of a kind per batch."""
        self.L.mskfh_runner_set_workers(self.h, int(fe_workers), int(ekf_workers))

    def imu(self, stream, sample):
        self.L.mskfh_runner_imu(self.h, stream, C.byref(sample))

    def step(self, cam0_list, cam1_list, times):
        """One frame for every stream from host images (numpy u8 arrays)."""
        a = [np.ascontiguousarray(x, dtype=np.uint8) for x in cam0_list]
        b = [np.ascontiguousarray(x, dtype=np.uint8) for x in cam1_list]
        pa = (C.c_void_p * self.n)(*[x.ctypes.data for x in a])
        pb = (C.c_void_p * self.n)(*[x.ctypes.data for x in b])
        t = np.ascontiguousarray(times, dtype=np.float64)
        self._chk(self.L.mskfh_runner_step(self.h, pa, pb, 0, _p(t)))

    def set_sequence(self, stream, cam0_base_ptr, cam1_base_ptr, on_device, frame_bytes, n_static, n_loop, t0_ns, frame_dt_ns, imu):
        imu = np.ascontiguousarray(imu, dtype=IMU_SAMPLE)
        self._keep.append(imu)
        self.L.mskfh_runner_set_sequence(self.h, stream, cam0_base_ptr, cam1_base_ptr, int(on_device), frame_bytes, n_static, n_loop,
                                         t0_ns, frame_dt_ns, _p(imu), len(imu))

    def run(self, first, n, threaded=True, pipelined=False):
        """Frames [first, first+n) of the attached sequences.  pipelined: front-end and filter of every group run
        as a two-stage pipeline on two HIP streams (identical results: the front-end never reads filter state)."""
        self._chk(self.L.mskfh_runner_run(self.h, first, n, int(threaded), int(pipelined)))

    def run_timed(self, first, warmup, steps, max_extra=8):
        """ONE pipelined run of every group over at least `warmup` + `steps` frames (no fill / drain of the group pipelines at
        the boundary).  Returns the seconds in which the groups together completed frames n_groups * warmup + 1 ..
        n_groups * (warmup + steps) of the run: exactly `steps` steps' worth of stream-frames, with every group busy from
        before that window opens until after it closes (MultiRunner::run_timed)."""
        el = C.c_double(0.0)
        self._chk(self.L.mskfh_runner_run_timed(self.h, first, warmup, steps, max_extra, C.byref(el)))
        return el.value

    def frames_done(self, group=0):
        """Next frame index of a group (absolute, its stagger offset included)."""
        return self.L.mskfh_runner_frames_done(self.h, group)

    def window(self, group=0):
        out = np.zeros(7)
        self.L.mskfh_runner_window(self.h, group, _p(out))
        return dict(zip(("fe_open", "fe_close", "ekf_open", "ekf_close", "fe_frames", "ekf_frames", "frames_at_close"), (float(x) for x in out)))

    def mark_dump(self, group=0):
        """(ids, lifetimes, cam0, cam1, imu_state[28]) of local stream 0 of a group at the close of its timed window."""
        n = self.L.mskfh_runner_mark_dump_size(self.h, group)
        if n < 0:
            raise MskfError("no timed window has been closed")
        ids = np.zeros(n, np.uint64)
        life = np.zeros(n, np.int32)
        c0 = np.zeros(n, POINT2F)
        c1 = np.zeros(n, POINT2F)
        imu = np.zeros(28)
        self.L.mskfh_runner_mark_dump(self.h, group, _p(ids), _p(life), _p(c0), _p(c1), _p(imu))
        return ids, life, c0, c1, imu

    def set_stagger(self, delta):
        """Group g works g * delta frames ahead of the index passed to run() (MultiRunner::set_stagger)."""
        self.L.mskfh_runner_set_stagger(self.h, int(delta))

    def group_offset(self, g):
        return self.L.mskfh_runner_group_offset(self.h, g)

    def keep_trajectory(self, keep, stream=None):
        if stream is None:
            self.L.mskfh_runner_keep_trajectory(self.h, int(keep))
        else:
            self.L.mskfh_runner_keep_trajectory_stream(self.h, int(stream), int(keep))

    KERNELS = ["k_pyr_down", "k_detect_cells", "k_track4", "k_ekf_propagate", "k_ekf_augment", "k_ekf_feature_blocks",
               "k_ekf_tsqr", "k_ekf_gemm", "k_ekf_chol_lds", "k_ekf_trsm", "k_ekf_small", "k_ekf_remove_clone", "k_pt_geom", "k_fe_book"]

    def set_timing(self, enable):
        self.L.mskfh_runner_set_timing(self.h, int(enable))

    def get_timing(self, reset=True):
        """Per-kernel HIP-event timing accumulated on the groups' own streams: {kernel: (ms, launches, units)}."""
        k = len(self.KERNELS)
        ms = np.zeros(k)
        launches = np.zeros(k, np.int64)
        units = np.zeros(k, np.int64)
        self.L.mskfh_runner_get_timing(self.h, _p(ms), _p(launches), _p(units), int(reset))
        return {n: (float(ms[i]), int(launches[i]), int(units[i])) for i, n in enumerate(self.KERNELS)}

    PHASES = ["push", "fe_prepare1", "track1", "fe_after1", "track2", "fe_after2", "ekf_A", "update1", "ekf_B", "update2",
              "ekf_C", "posvar", "imu_feed", "handoff", "fe_queue_wait", "ekf_queue_wait", "imu_feed_ekf"]
    FE_THREAD_PHASES = ["imu_feed", "push", "fe_prepare1", "track1", "fe_after1", "track2", "fe_after2", "handoff", "fe_queue_wait"]
    EKF_THREAD_PHASES = ["ekf_queue_wait", "imu_feed_ekf", "ekf_A", "update1", "ekf_B", "update2", "ekf_C", "posvar"]

    def get_window_phases_group(self, g):
        """The same for one group."""
        out = np.zeros(len(self.PHASES))
        self.L.mskfh_runner_get_window_phases_group.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.L.mskfh_runner_get_window_phases_group(self.h, int(g), _p(out))
        return {n: float(out[i]) for i, n in enumerate(self.PHASES)}

    def get_window_phases(self):
        """Wall seconds per phase inside the last timed window, summed over groups."""
        out = np.zeros(len(self.PHASES))
        self.L.mskfh_runner_get_window_phases(self.h, _p(out))
        return {n: float(out[i]) for i, n in enumerate(self.PHASES)}

    def get_phases(self, reset=True):
        """Wall seconds per BatchGroup::step phase, summed over groups (host bookkeeping vs device calls)."""
        out = np.zeros(len(self.PHASES))
        self.L.mskfh_runner_get_phases(self.h, _p(out), int(reset))
        return {n: float(out[i]) for i, n in enumerate(self.PHASES)}

    def get_abi_host_time(self, reset=True):
        """Host seconds inside the batched C-ABI calls, outside the device waits, summed over groups."""
        out = np.zeros(4)
        self.L.mskfh_runner_get_abi_host_time.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.mskfh_runner_get_abi_host_time(self.h, _p(out), int(reset))
        return dict(zip(("update_pack", "update_unpack", "track_pack", "track_unpack"), (float(x) for x in out)))

    def get_hostprof(self, reset=True):
        """Seconds of host bookkeeping per slot of csrc/host/host_prof.h, summed over all streams and threads."""
        out = np.zeros(64)
        self.L.mskfh_get_hostprof.restype = C.c_int
        self.L.mskfh_hostprof_name.restype = C.c_char_p
        n = self.L.mskfh_get_hostprof(_p(out), 64, int(reset))
        return {self.L.mskfh_hostprof_name(i).decode(): float(out[i]) for i in range(n)}

    def hip_stream(self, stream=0):
        return self.L.mskfh_group_hip_stream(self.h, stream)

    # ---- inspection
    def dump(self, stream=0):
        n = self.L.mskfh_num_features(self.h, stream)
        ids = np.zeros(n, np.uint64)
        life = np.zeros(n, np.int32)
        c0 = np.zeros(n, POINT2F)
        c1 = np.zeros(n, POINT2F)
        info = TrackingInfo()
        self.L.mskfh_get_dump(self.h, stream, _p(ids), _p(life), _p(c0), _p(c1), C.byref(info))
        return ids, life, c0, c1, info

    def msg(self, stream=0):
        n = self.L.mskfh_msg_size(self.h, stream)
        out = np.zeros(n, FEATURE_MEAS)
        if n:
            self.L.mskfh_get_msg(self.h, stream, _p(out))
        return out

    def poses(self, stream=0):
        n = self.L.mskfh_num_poses(self.h, stream)
        out = np.zeros(n, POSE)
        if n:
            self.L.mskfh_get_poses(self.h, stream, _p(out))
        return out

    def cov(self, stream=0):
        d = self.L.mskfh_state_dim(self.h, stream)
        P = np.zeros((d, d))
        self._chk(self.L.mskfh_get_cov(self.h, stream, _p(P), P.size))
        return P

    def imu_state(self, stream=0):
        out = np.zeros(28)
        self.L.mskfh_get_imu_state(self.h, stream, _p(out))
        return out

    def num_updates(self, stream=0):
        return self.L.mskfh_num_updates(self.h, stream)

    def num_device_frames(self, stream=0):
        """Front-end frames of the stream that ran as ONE device call (mskf_fe_frame_batch_*), books and RANSAC included."""
        return int(self.L.mskfh_num_device_frames(self.h, stream))

    def num_tsqr_updates(self, stream=0):
        """Updates of the stream whose QR compression ran as Householder TSQR (the rest: Gram + Cholesky)."""
        return self.L.mskfh_num_tsqr_updates(self.h, stream)

    def num_uncompressed_updates(self, stream=0):
        """Updates whose stack had no more rows than active columns and was used as it is (msckf_vio.cpp:818-821)."""
        return self.L.mskfh_num_uncompressed_updates(self.h, stream)

    def stacked_rows(self, stream=0):
        return self.L.mskfh_stacked_rows(self.h, stream)

    def num_resets(self, stream=0):
        return self.L.mskfh_num_resets(self.h, stream)

    def num_clones(self, stream=0):
        return self.L.mskfh_num_clones(self.h, stream)


class StreamView:
    """Adapter so that oracle_py.Synth.feed() can drive one stream of a Runner with the reference
    harness call order (imu_callback xN, stereo_callback, backend_callback)."""

    def __init__(self, runner):
        assert runner.n == 1
        self.r = runner
        self._pending = None

    def imu(self, s):
        self.r.imu(0, s)

    def stereo(self, cam0, cam1, t):
        self._pending = (cam0, cam1, t)

    def backend(self):
        cam0, cam1, t = self._pending
        self.r.step([cam0], [cam1], [t])
