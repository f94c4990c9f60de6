"""ctypes loader for the CPU oracle and the synthetic generator (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from msckf_stereo_c_amd.ctypes_types import (CORNER, FEATURE_MEAS, POINT2F, POSE, Calib, EkfCfg, FeCfg, ImuSample,
                                              TrackingInfo)

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


_built = False


def build(force=False):
    """make decides what is stale (the libraries under _build/ travel with the tree, the sources may be newer)."""
    global _built
    if _built and not force:
        return
    import fcntl
    os.makedirs(_BUILD, exist_ok=True)
    with open(os.path.join(_BUILD, ".lock"), "w") as lk:      # several ranks may get here at once
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
        fcntl.flock(lk, fcntl.LOCK_UN)
    _built = True


_lib = None
_synth = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_BUILD, "liboracle.so"))
        L.orc_system_create.restype = C.c_void_p
        L.orc_system_create.argtypes = [C.POINTER(Calib), C.POINTER(FeCfg), C.POINTER(EkfCfg)]
        for name in ("orc_system_destroy", "orc_system_backend"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_system_imu.argtypes = [C.c_void_p, C.POINTER(ImuSample)]
        L.orc_system_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
        for name in ("orc_system_num_features", "orc_system_msg_size", "orc_system_num_poses", "orc_system_state_dim",
                     "orc_system_num_updates", "orc_system_map_size"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.orc_system_num_resets.argtypes = [C.c_void_p]
        L.orc_system_num_resets.restype = C.c_longlong
        L.orc_system_get_dump.argtypes = [C.c_void_p] * 5 + [C.POINTER(TrackingInfo)]
        L.orc_system_get_msg.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_system_get_poses.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_system_get_cov.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_system_get_imu_state.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_lk_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_cell_maxima.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_detect.restype = C.c_int
        L.orc_undistort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_distort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_stereo_match.argtypes = [C.POINTER(Calib), C.POINTER(FeCfg), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        L.orc_ekf_update_problem.restype = C.c_int
        L.orc_ekf_update_problem.argtypes = [C.POINTER(Calib), C.POINTER(EkfCfg), C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_triangulate.argtypes = [C.POINTER(Calib), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def synth_lib():
    global _synth
    if _synth is None:
        build()
        S = C.CDLL(os.path.join(_BUILD, "libmskf_synth.so"))
        S.synth_create.restype = C.c_void_p
        S.synth_create.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        S.synth_destroy.argtypes = [C.c_void_p]
        S.synth_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        S.synth_frame_time.argtypes = [C.c_void_p, C.c_int]
        S.synth_frame_time.restype = C.c_double
        S.synth_imu.argtypes = [C.c_void_p, C.c_int, C.POINTER(ImuSample)]
        S.synth_gt_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        S.synth_euroc_calib.argtypes = [C.c_int, C.c_int, C.POINTER(Calib)]
        S.synth_render_params.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        S.synth_ray_table.argtypes = [C.c_void_p, C.c_int]
        S.synth_ray_table.restype = C.POINTER(C.c_float)
        _synth = S
    return _synth


from msckf_stereo_c_amd.ctypes_types import RENDER_IMG   # synth::RenderImg


def euroc_calib(w, h):
    c = Calib()
    synth_lib().synth_euroc_calib(w, h, C.byref(c))
    return c


class Synth:
    """Seeded EuRoC-shaped stereo + IMU stream (msckf_stereo_c_amd/csrc/synth/synth.h)."""

    def __init__(self, seed=0x5EED0000, width=752, height=480, n_static=25, n_loop=200, pixel_sigma=2.0,
                 motion_scale=1.0):
        self.S = synth_lib()
        self.w, self.h = width, height
        self.n_static, self.n_loop = n_static, n_loop
        self.h_ = self.S.synth_create(seed, width, height, n_static, n_loop, pixel_sigma, motion_scale)
        self.calib = euroc_calib(width, height)
        self.imu_per_frame = 10
        self._cache = {}

    def __del__(self):
        try:
            self.S.synth_destroy(self.h_)
        except Exception:
            pass

    def frame_key(self, k):
        return k if k < self.n_static else self.n_static + (k - self.n_static) % self.n_loop

    def render(self, k):
        key = self.frame_key(k)
        if key not in self._cache:
            a = np.empty((self.h, self.w), np.uint8)
            b = np.empty((self.h, self.w), np.uint8)
            self.S.synth_render(self.h_, key, _p(a), _p(b))
            self._cache[key] = (a, b)
        return self._cache[key]

    def frame_time(self, k):
        return self.S.synth_frame_time(self.h_, k)

    def render_params(self, k, cam):
        """Inputs of the device renderer (csrc/synth/synth_render.hip) for camera `cam` of frame k: one RENDER_IMG record."""
        out = np.zeros(1, RENDER_IMG)
        self.S.synth_render_params(self.h_, self.frame_key(k), cam, _p(out))
        return out[0]

    def ray_table(self, cam):
        """Undistorted viewing rays of camera `cam`, (h, w, 2) float32 (the same for every seed: it depends on the calibration only)."""
        p = self.S.synth_ray_table(self.h_, cam)
        return np.ctypeslib.as_array(p, shape=(self.h, self.w, 2)).copy()

    def imu(self, j):
        s = ImuSample()
        self.S.synth_imu(self.h_, j, C.byref(s))
        return s

    def gt_pose(self, k):
        out = np.zeros(1, POSE)
        self.S.synth_gt_pose(self.h_, k, _p(out))
        return out[0]

    def feed(self, system, n_frames, on_frame=None, start=0):
        """Reference harness call order (apps/run_euroc_single_thread.cpp:189-254, Q10): per image,
        feed IMU samples do-while t_imu <= t_img, then stereo_callback, then backend_callback."""
        j = getattr(system, "_imu_cursor", 0)
        for k in range(start, start + n_frames):
            t_img = self.frame_time(k)
            while True:
                s = self.imu(j)
                j += 1
                system.imu(s)
                if not (s.time_stamp <= t_img):
                    break
            a, b = self.render(k)
            system.stereo(a, b, t_img)
            system.backend()
            if on_frame:
                on_frame(k, system)
        system._imu_cursor = j


class OracleSystem:
    """cg::System restated on the CPU (oracle/o_capi.cpp OSystem)."""

    def __init__(self, calib, fe_cfg, ekf_cfg):
        self.L = lib()
        self.calib, self.fe_cfg, self.ekf_cfg = calib, fe_cfg, ekf_cfg
        self.h = self.L.orc_system_create(C.byref(calib), C.byref(fe_cfg), C.byref(ekf_cfg))

    def __del__(self):
        try:
            self.L.orc_system_destroy(self.h)
        except Exception:
            pass

    def imu(self, s):
        self.L.orc_system_imu(self.h, C.byref(s))

    def stereo(self, cam0, cam1, t):
        cam0 = np.ascontiguousarray(cam0)
        cam1 = np.ascontiguousarray(cam1)
        self.L.orc_system_stereo(self.h, _p(cam0), _p(cam1), cam0.shape[1], cam0.shape[0], t)

    def backend(self):
        self.L.orc_system_backend(self.h)

    def dump(self):
        n = self.L.orc_system_num_features(self.h)
        ids = np.zeros(n, np.uint64)
        life = np.zeros(n, np.int32)
        c0 = np.zeros(n, POINT2F)
        c1 = np.zeros(n, POINT2F)
        info = TrackingInfo()
        self.L.orc_system_get_dump(self.h, _p(ids), _p(life), _p(c0), _p(c1), C.byref(info))
        return ids, life, c0, c1, info

    def msg(self):
        n = self.L.orc_system_msg_size(self.h)
        out = np.zeros(n, FEATURE_MEAS)
        if n:
            self.L.orc_system_get_msg(self.h, _p(out))
        return out

    def poses(self):
        n = self.L.orc_system_num_poses(self.h)
        out = np.zeros(n, POSE)
        if n:
            self.L.orc_system_get_poses(self.h, _p(out))
        return out

    def cov(self):
        d = self.L.orc_system_state_dim(self.h)
        P = np.zeros((d, d))
        self.L.orc_system_get_cov(self.h, _p(P))
        return P

    def imu_state(self):
        out = np.zeros(28)
        self.L.orc_system_get_imu_state(self.h, _p(out))
        return out

    def num_updates(self):
        return self.L.orc_system_num_updates(self.h)

    def num_resets(self):
        return self.L.orc_system_num_resets(self.h)


# ---- kernel-level entry points
def pyr_down(img):
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(_p(np.ascontiguousarray(img)), w, h, _p(out))
    return out


def build_pyramid(img):
    pyr = [np.ascontiguousarray(img)]
    for _ in range(3):
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def lk_track(imgA, imgB, ptsA, ptsB_init):
    h, w = imgA.shape
    a = np.ascontiguousarray(ptsA, dtype=np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(ptsB_init, dtype=np.float32).reshape(-1, 2).copy()
    st = np.zeros(len(a), np.uint8)
    lib().orc_lk_track(_p(np.ascontiguousarray(imgA)), _p(np.ascontiguousarray(imgB)), w, h, len(a), _p(a), _p(b), _p(st))
    return b, st


def cell_maxima(img, det_rows=30, det_cols=47):
    h, w = img.shape
    out = np.zeros(det_rows * det_cols, CORNER)
    lib().orc_cell_maxima(_p(np.ascontiguousarray(img)), w, h, det_rows, det_cols, _p(out))
    return out


def detect(img, det_rows=30, det_cols=47, thr=10, occupancy=None):
    h, w = img.shape
    pts = np.zeros((det_rows * det_cols, 2), np.float32)
    resp = np.zeros(det_rows * det_cols, np.float64)
    occ_arr = None if occupancy is None else np.ascontiguousarray(occupancy, dtype=np.uint8)
    occ = None if occ_arr is None else _p(occ_arr)
    n = lib().orc_detect(_p(np.ascontiguousarray(img)), w, h, det_rows, det_cols, thr, occ, _p(pts), _p(resp))
    return pts[:n], resp[:n]


def undistort(K, D, pts, R=None, Pn=None, model=0):
    K = np.ascontiguousarray(K, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    R = np.eye(3) if R is None else np.ascontiguousarray(R, dtype=np.float64)
    Pn = np.array([1.0, 1.0, 0.0, 0.0]) if Pn is None else np.ascontiguousarray(Pn, dtype=np.float64)
    a = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    out = np.zeros_like(a)
    lib().orc_undistort(_p(K), _p(D), model, _p(R), _p(Pn), len(a), _p(a), _p(out))
    return out


def distort(K, D, pts, model=0):
    K = np.ascontiguousarray(K, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    a = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    out = np.zeros_like(a)
    lib().orc_distort(_p(K), _p(D), model, len(a), _p(a), _p(out))
    return out


def stereo_match(calib, fe_cfg, cam0, cam1, pts0):
    a = np.ascontiguousarray(pts0, dtype=np.float32).reshape(-1, 2)
    b = np.zeros_like(a)
    m = np.zeros(len(a), np.uint8)
    cam0 = np.ascontiguousarray(cam0)
    cam1 = np.ascontiguousarray(cam1)
    lib().orc_stereo_match(C.byref(calib), C.byref(fe_cfg), _p(cam0), _p(cam1), len(a), _p(a), _p(b), _p(m))
    return b, m


def two_point_ransac(calib, fe_cfg, cam, pts1, pts2, R_p_c, inlier_error=3.0, success_probability=0.99, draws=0):
    """ImageProcessor::twoPointRansac (image_processor.cpp:911-1135) -> (markers, draw counter afterwards)."""
    a = np.ascontiguousarray(pts1, dtype=np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(pts2, dtype=np.float32).reshape(-1, 2)
    R = np.ascontiguousarray(R_p_c, dtype=np.float64).reshape(9)
    m = np.zeros(len(a), np.int32)
    d = C.c_ulonglong(draws)
    f = lib().orc_two_point_ransac
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    f(C.byref(calib), C.byref(fe_cfg), cam, len(a), _p(a), _p(b), _p(R), inlier_error, success_probability, C.byref(d), _p(m))
    return m, d.value


def ekf_update_problem(calib, ekf_cfg, gravity, clones, P, positions, obs_start, obs_clone, obs_z, dof_offset):
    clones = np.ascontiguousarray(clones, dtype=np.float64)
    n_clones = clones.shape[0]
    d = 21 + 6 * n_clones
    P = np.ascontiguousarray(P, dtype=np.float64).copy()
    positions = np.ascontiguousarray(positions, dtype=np.float64)
    obs_start = np.ascontiguousarray(obs_start, dtype=np.int32)
    obs_clone = np.ascontiguousarray(obs_clone, dtype=np.int32)
    obs_z = np.ascontiguousarray(obs_z, dtype=np.float64)
    n_feat = len(obs_start) - 1
    gamma = np.zeros(n_feat)
    passed = np.zeros(n_feat, np.uint8)
    dx = np.zeros(d)
    g = np.ascontiguousarray(gravity, dtype=np.float64)
    rows = lib().orc_ekf_update_problem(C.byref(calib), C.byref(ekf_cfg), _p(g), n_clones, _p(clones), _p(P), n_feat,
                                        _p(positions), _p(obs_start), _p(obs_clone), _p(obs_z), dof_offset, _p(gamma),
                                        _p(passed), _p(dx))
    return dict(rows=rows, gamma=gamma, passed=passed, delta_x=dx, P=P)


def triangulate(calib, clones, obs_start, obs_clone, obs_z):
    clones = np.ascontiguousarray(clones, dtype=np.float64)
    obs_start = np.ascontiguousarray(obs_start, dtype=np.int32)
    obs_clone = np.ascontiguousarray(obs_clone, dtype=np.int32)
    obs_z = np.ascontiguousarray(obs_z, dtype=np.float64)
    n_feat = len(obs_start) - 1
    pos = np.zeros((n_feat, 3))
    valid = np.zeros(n_feat, np.uint8)
    lib().orc_triangulate(C.byref(calib), clones.shape[0], _p(clones), n_feat, _p(obs_start), _p(obs_clone), _p(obs_z),
                          _p(pos), _p(valid))
    return pos, valid
