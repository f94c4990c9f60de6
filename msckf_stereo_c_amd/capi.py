"""ctypes binding of the C-ABI (include/mskf_hip.h).  Thin: every call goes straight to the HIP
library; there is no Python/CPU fallback and a missing library or device raises."""
import ctypes as C
import os

import numpy as np

from .ctypes_types import CORNER, POINT2F, Calib, EkfCfg, FeCfg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MskfError(RuntimeError):
    pass


class TrackArgs(C.Structure):
    _fields_ = [("n", C.c_int32), ("do_temporal", C.c_int32), ("in_pts", C.c_void_p), ("Hpred", C.c_double * 9),
                ("out0", C.c_void_p), ("out1", C.c_void_p), ("und0", C.c_void_p), ("und1", C.c_void_p),
                ("status", C.c_void_p)]


class EkfFeature(C.Structure):
    _fields_ = [("obs_start", C.c_int32), ("n_obs", C.c_int32), ("needs_init", C.c_int32), ("init_start", C.c_int32),
                ("n_init", C.c_int32), ("_pad", C.c_int32), ("position", C.c_double * 3)]


EKF_FEATURE = np.dtype([("obs_start", "<i4"), ("n_obs", "<i4"), ("needs_init", "<i4"), ("init_start", "<i4"),
                        ("n_init", "<i4"), ("_pad", "<i4"), ("position", "<f8", 3)])
CLONE_STATE = np.dtype([("q", "<f8", 4), ("p", "<f8", 3), ("q_null", "<f8", 4), ("p_null", "<f8", 3)])


class EkfUpdateArgs(C.Structure):
    _fields_ = [("n_clones", C.c_int32), ("n_feat", C.c_int32), ("n_obs", C.c_int32), ("dof_offset", C.c_int32),
                ("apply_row_cap", C.c_int32), ("_pad", C.c_int32), ("gravity", C.c_double * 3),
                ("clones", C.c_void_p), ("features", C.c_void_p), ("obs_clone", C.c_void_p), ("obs_z", C.c_void_p),
                ("delta_x", C.c_void_p), ("feat_status", C.c_void_p), ("gamma", C.c_void_p), ("rows_out", C.c_void_p),
                ("diag_out", C.c_void_p), ("pos_var_out", C.c_void_p)]


class FeFrameArgs(C.Structure):   # mskf_fe_frame_args (include/mskf_hip.h)
    _fields_ = [("Hpred", C.c_double * 9), ("capacity", C.c_int32), ("n", C.c_int32),
                ("id", C.c_void_p), ("lifetime", C.c_void_p), ("cam0", C.c_void_p), ("cam1", C.c_void_p), ("und0", C.c_void_p), ("und1", C.c_void_p),
                ("before_tracking", C.c_int32), ("after_tracking", C.c_int32), ("after_matching", C.c_int32), ("after_ransac", C.c_int32),
                ("n_candidates", C.c_int32), ("n_new", C.c_int32), ("next_feature_id", C.c_uint64),
                ("R_p_c", (C.c_double * 9) * 2), ("ransac_draws", C.c_uint64)]


EXPORTS = [
    "mskf_last_error", "mskf_abi_version", "mskf_ctx_create", "mskf_ctx_create_prio", "mskf_ctx_create_shared", "mskf_ctx_destroy", "mskf_ctx_sync", "mskf_ctx_hip_stream",
    "mskf_stream_create", "mskf_stream_destroy", "mskf_fe_push_stereo", "mskf_fe_push_stereo_device",
    "mskf_fe_push_stereo_batch", "mskf_fe_set_detect_floor", "mskf_fe_get_cell_maxima", "mskf_fe_get_cell_candidates", "mskf_fe_track", "mskf_fe_track_batch", "mskf_fe_swap",
    "mskf_fe_get_level", "mskf_ekf_reset", "mskf_ekf_propagate", "mskf_ekf_augment", "mskf_ekf_update",
    "mskf_ekf_update_batch", "mskf_ekf_remove_clone", "mskf_ekf_remove_clones_batch", "mskf_ekf_predict_batch", "mskf_ekf_propagate_imu",
    "mskf_ekf_get_pos_var", "mskf_ekf_get_pos_var_batch", "mskf_ctx_set_timing", "mskf_ctx_get_timing", "mskf_stream_ctx",
    "mskf_ekf_get_dim", "mskf_ekf_get_cov", "mskf_ekf_set_cov", "mskf_ekf_debug_read", "mskf_ctx_get_host_time", "mskf_fe_track_batch_begin", "mskf_fe_track_batch_end",
    "mskf_ekf_update_batch_begin", "mskf_ekf_update_batch_end", "mskf_ekf_get_pos_var_batch_begin", "mskf_ekf_get_pos_var_batch_end", "mskf_ctx_timing_gate",
    "mskf_fe_grid_capacity", "mskf_fe_set_grid", "mskf_fe_frame_batch_begin", "mskf_fe_frame_batch_end", "mskf_ctx_set_wait_mode",
    "mskf_stream_rebind", "mskf_ctx_record_point", "mskf_ctx_wait_point", "mskf_point_destroy", "mskf_ekf_set_compression_mode",
]


def lib_path():
    return os.path.join(_HERE, "_build", "libmskf_hip.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise MskfError("libmskf_hip.so is not built (run python -m msckf_stereo_c_amd.build); there is no CPU fallback")
        L = C.CDLL(p)
        L.mskf_last_error.restype = C.c_char_p
        L.mskf_ctx_hip_stream.restype = C.c_void_p
        L.mskf_ctx_hip_stream.argtypes = [C.c_void_p]
        L.mskf_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mskf_ctx_destroy.argtypes = [C.c_void_p]
        L.mskf_ctx_sync.argtypes = [C.c_void_p]
        L.mskf_stream_create.argtypes = [C.c_void_p, C.POINTER(Calib), C.POINTER(FeCfg), C.POINTER(EkfCfg), C.POINTER(C.c_void_p)]
        L.mskf_stream_destroy.argtypes = [C.c_void_p]
        L.mskf_fe_push_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        L.mskf_fe_get_cell_maxima.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.mskf_fe_track.argtypes = [C.c_void_p, C.POINTER(TrackArgs)]
        L.mskf_fe_swap.argtypes = [C.c_void_p]
        L.mskf_fe_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mskf_ekf_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.mskf_ekf_propagate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mskf_ekf_augment.argtypes = [C.c_void_p, C.c_void_p]
        L.mskf_ekf_update.argtypes = [C.c_void_p, C.POINTER(EkfUpdateArgs)]
        L.mskf_ekf_remove_clone.argtypes = [C.c_void_p, C.c_int]
        L.mskf_ekf_get_dim.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.mskf_ekf_get_cov.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mskf_ekf_set_cov.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != 0:
        raise MskfError("mskf status %d: %s" % (rc, lib().mskf_last_error().decode()))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.mskf_ctx_create(device, C.byref(self.h)))
        self.streams = []

    def close(self):
        if self.h:
            for s in list(self.streams):
                s.close()
            self.L.mskf_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        _chk(self.L.mskf_ctx_sync(self.h))

    def ekf_update_batch(self, streams, problems):
        """One mskf_ekf_update_batch over several streams of this context; problems[i] = kwargs of Stream.ekf_update."""
        n = len(streams)
        built = [Stream._update_args(**pr) for pr in problems]
        args = (EkfUpdateArgs * n)(*[b[0] for b in built])
        hs = (C.c_void_p * n)(*[s.h for s in streams])
        self.L.mskf_ekf_update_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(EkfUpdateArgs)]
        _chk(self.L.mskf_ekf_update_batch(self.h, n, hs, args))
        return [b[2]() for b in built]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """One VIO stream: device-resident pyramids + covariance behind the C-ABI."""

    def __init__(self, ctx, calib, fe_cfg, ekf_cfg):
        self.ctx, self.L = ctx, ctx.L
        self.calib, self.fe_cfg, self.ekf_cfg = calib, fe_cfg, ekf_cfg
        self.h = C.c_void_p()
        _chk(self.L.mskf_stream_create(ctx.h, C.byref(calib), C.byref(fe_cfg), C.byref(ekf_cfg), C.byref(self.h)))
        ctx.streams.append(self)

    def close(self):
        if self.h:
            self.L.mskf_stream_destroy(self.h)
            self.h = None
            if self in self.ctx.streams:
                self.ctx.streams.remove(self)

    # ---- front-end
    def push_stereo(self, cam0, cam1, t=0.0, pitch=None):
        """cam0 / cam1: h x w images, or (pitch given) h x pitch buffers whose first w = calib.width columns are the image."""
        cam0 = np.ascontiguousarray(cam0, dtype=np.uint8)
        cam1 = np.ascontiguousarray(cam1, dtype=np.uint8)
        h, w = cam0.shape
        if pitch is not None:
            assert cam0.shape[1] == pitch and cam1.shape == cam0.shape
            w = self.calib.width
        _chk(self.L.mskf_fe_push_stereo(self.h, _p(cam0), _p(cam1), w, h, w if pitch is None else pitch, t))

    def set_detect_floor(self, min_score):
        _chk(self.L.mskf_fe_set_detect_floor(self.h, int(min_score)))

    def cell_maxima(self):
        n = self.fe_cfg.det_rows * self.fe_cfg.det_cols
        out = np.zeros(n, CORNER)
        got = C.c_int()
        _chk(self.L.mskf_fe_get_cell_maxima(self.h, _p(out), n, C.byref(got)))
        return out[:got.value]

    def cell_candidates(self, min_score):
        """Per-cell maxima whose score exceeds min_score (1/256 units), in cell order."""
        n = self.fe_cfg.det_rows * self.fe_cfg.det_cols
        out = np.zeros(n, CORNER)
        got = C.c_int()
        self.L.mskf_fe_get_cell_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _chk(self.L.mskf_fe_get_cell_candidates(self.h, int(min_score), _p(out), n, C.byref(got)))
        return out[:got.value]

    def track(self, pts, do_temporal, Hpred=None):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        n = len(pts)
        out0, out1, und0, und1 = (np.zeros((n, 2), np.float32) for _ in range(4))
        status = np.zeros(n, np.uint8)
        a = TrackArgs()
        a.n, a.do_temporal = n, int(do_temporal)
        a.in_pts = pts.ctypes.data
        H = np.eye(3) if Hpred is None else np.ascontiguousarray(Hpred, dtype=np.float64)
        a.Hpred[:] = list(H.reshape(-1))
        a.out0, a.out1, a.und0, a.und1, a.status = (x.ctypes.data for x in (out0, out1, und0, und1, status))
        _chk(self.L.mskf_fe_track(self.h, C.byref(a)))
        return dict(out0=out0, out1=out1, und0=und0, und1=und1, status=status)

    def swap(self):
        _chk(self.L.mskf_fe_swap(self.h))

    def get_level(self, role, level):
        w, h = C.c_int(), C.c_int()
        buf = np.zeros(self.calib.width * self.calib.height, np.uint8)
        _chk(self.L.mskf_fe_get_level(self.h, role, level, _p(buf), buf.size, C.byref(w), C.byref(h)))
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    # ---- EKF
    def ekf_reset(self, P0):
        P0 = np.ascontiguousarray(P0, dtype=np.float64)
        assert P0.shape == (21, 21)
        _chk(self.L.mskf_ekf_reset(self.h, _p(P0)))

    def ekf_set_cov(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64)
        _chk(self.L.mskf_ekf_set_cov(self.h, _p(P), P.shape[0]))

    def ekf_dim(self):
        d = C.c_int()
        _chk(self.L.mskf_ekf_get_dim(self.h, C.byref(d)))
        return d.value

    def ekf_get_cov(self):
        d = self.ekf_dim()
        P = np.zeros((d, d))
        _chk(self.L.mskf_ekf_get_cov(self.h, _p(P), P.size))
        return P

    def ekf_propagate(self, Phi, Q):
        Phi = np.ascontiguousarray(Phi, dtype=np.float64).reshape(-1, 21, 21)
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(-1, 21, 21)
        _chk(self.L.mskf_ekf_propagate(self.h, len(Phi), _p(Phi), _p(Q)))

    def ekf_augment(self, J):
        J = np.ascontiguousarray(J, dtype=np.float64)
        assert J.shape == (6, 21)
        _chk(self.L.mskf_ekf_augment(self.h, _p(J)))

    def ekf_remove_clone(self, idx):
        _chk(self.L.mskf_ekf_remove_clone(self.h, idx))

    @staticmethod
    def _update_args(gravity, clones, positions, obs_start, obs_clone, obs_z, dof_offset, apply_row_cap, needs_init=None, init_ranges=None):
        """(mskf_ekf_update_args, buffers it points into, result builder) for one stream."""
        clones = np.ascontiguousarray(clones, dtype=np.float64).reshape(-1, 14)
        n_clones = len(clones)
        obs_start = np.asarray(obs_start, dtype=np.int32)
        n_feat = len(obs_start) - 1
        obs_clone = np.ascontiguousarray(obs_clone, dtype=np.int32)
        obs_z = np.ascontiguousarray(obs_z, dtype=np.float64).reshape(-1, 4)
        feats = np.zeros(n_feat, EKF_FEATURE)
        feats["obs_start"] = obs_start[:-1]
        feats["n_obs"] = np.diff(obs_start)
        if positions is not None:
            feats["position"] = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        if needs_init is not None:
            feats["needs_init"] = np.asarray(needs_init, dtype=np.int32)
            if init_ranges is None:
                feats["init_start"] = feats["obs_start"]
                feats["n_init"] = feats["n_obs"]
            else:
                feats["init_start"] = [r[0] for r in init_ranges]
                feats["n_init"] = [r[1] for r in init_ranges]
        d = 21 + 6 * n_clones
        dx = np.zeros(d)
        status = np.zeros(max(n_feat, 1), np.uint8)
        gamma = np.zeros(max(n_feat, 1))
        rows = np.zeros(1, np.int32)
        diag = np.zeros(2, np.int32)
        a = EkfUpdateArgs()
        a.n_clones, a.n_feat, a.n_obs = n_clones, n_feat, len(obs_clone)
        a.dof_offset, a.apply_row_cap = dof_offset, int(apply_row_cap)
        a.gravity[:] = list(np.asarray(gravity, dtype=np.float64))
        a.clones, a.features, a.obs_clone, a.obs_z = clones.ctypes.data, feats.ctypes.data, obs_clone.ctypes.data, obs_z.ctypes.data
        a.delta_x, a.feat_status, a.gamma, a.rows_out = dx.ctypes.data, status.ctypes.data, gamma.ctypes.data, rows.ctypes.data
        a.diag_out = diag.ctypes.data
        keep = (clones, feats, obs_clone, obs_z, dx, status, gamma, rows, diag)

        def result():
            return dict(delta_x=dx, status=status[:n_feat], gamma=gamma[:n_feat], rows=int(rows[0]),
                        positions=feats["position"].copy(), used_qr=int(diag[0]), tiny_pivots=int(diag[1]))
        return a, keep, result

    def ekf_update(self, gravity, clones, positions, obs_start, obs_clone, obs_z, dof_offset, apply_row_cap,
                   needs_init=None, init_ranges=None):
        """clones: (n,14) [q p q_null p_null]; obs_start: n_feat+1 offsets; returns dict."""
        a, keep, result = self._update_args(gravity, clones, positions, obs_start, obs_clone, obs_z, dof_offset, apply_row_cap, needs_init, init_ranges)
        _chk(self.L.mskf_ekf_update(self.h, C.byref(a)))
        return result()
