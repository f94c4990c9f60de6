"""twoPointRansac (image_processor.cpp:911-1135, dead code in the reference — SURVEY.md §8f-4): the CPU oracle's
restatement against a known answer, and the product's host implementation against the oracle, marker for marker.

parity unpinned: the reference's draws come from cg::uniform_integer (vikit_cg, absent); both sides use the same
counter-based generator instead, so the hypotheses are identical by construction and everything else is arithmetic."""
import ctypes as C
import os

import numpy as np
import pytest

from msckf_stereo_c_amd.ctypes_types import default_fe_cfg


def _scene(oracle, seed, n=240, n_out=40, translation=(0.04, -0.01, 0.02), rot=(0.004, -0.006, 0.003), radtan=True, out_px=(6, 25)):
    """Pixel pairs of static 3-D points seen before / after a small camera motion, the last n_out of them corrupted."""
    syn = oracle.Synth(seed=0x5EED0000, width=752, height=480)
    calib = syn.calib
    if not radtan:
        calib.cam0_distortion[:] = [0.0, 0.0, 0.0, 0.0]
    K, D = np.array(calib.cam0_intrinsics), np.array(calib.cam0_distortion)
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 9, n)], axis=1)
    th = np.linalg.norm(rot)
    k = np.array(rot) / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R_c_p = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx       # previous -> current camera frame
    Xc = X @ R_c_p.T + np.array(translation)
    n1 = (X[:, :2] / X[:, 2:]).astype(np.float32)
    n2 = (Xc[:, :2] / Xc[:, 2:]).astype(np.float32)
    px1 = oracle.distort(K, D, n1)
    px2 = oracle.distort(K, D, n2)
    px2[n - n_out:] += rng.uniform(out_px[0], out_px[1], (n_out, 2)).astype(np.float32) * rng.choice([-1, 1], (n_out, 2)).astype(np.float32)
    return calib, px1, px2, R_c_p


def test_oracle_ransac_known_answer(oracle):
    """General motion: every corrupted pair is rejected and (almost) every clean one kept."""
    calib, px1, px2, R_c_p = _scene(oracle, 1)
    markers, draws = oracle.two_point_ransac(calib, default_fe_cfg(), 0, px1, px2, R_c_p)
    assert draws == 14                                     # 7 hypotheses (success probability 0.99), two draws each
    assert markers[-40:].sum() <= 3                        # a corruption along the epipolar line is invisible to this test
    assert markers[:-40].sum() >= 195


def test_oracle_ransac_degenerate_and_small_inputs(oracle):
    """Pure rotation (no translation): the degenerate branch thresholds the residual flow; < 3 candidates reject all."""
    calib, px1, px2, R_c_p = _scene(oracle, 2, n_out=20, translation=(0.0, 0.0, 0.0), out_px=(3.5, 5.0))
    markers, draws = oracle.two_point_ransac(calib, default_fe_cfg(), 0, px1, px2, R_c_p)
    assert draws == 0                                      # mean flow < 1 px: no hypothesis is drawn in the degenerate branch
    assert markers[-20:].sum() == 0 and markers[:-20].sum() == 220
    m2, _ = oracle.two_point_ransac(calib, default_fe_cfg(), 0, px1[:2], px2[:2], R_c_p)
    assert m2.tolist() == [0, 0]
    m0, _ = oracle.two_point_ransac(calib, default_fe_cfg(), 0, px1[:0], px2[:0], R_c_p)
    assert len(m0) == 0


@pytest.mark.parametrize("seed,translation", [(3, (0.04, -0.01, 0.02)), (4, (0.0, 0.0, 0.0)), (5, (0.2, 0.1, -0.05)), (6, (0.002, 0.0, 0.001))])
def test_host_ransac_equals_oracle(oracle, seed, translation):
    """Product (host mirror, fed the undistorted points the device returns with every track) == oracle, exactly."""
    from msckf_stereo_c_amd import build
    _, host = build.build_all()
    L = C.CDLL(host)
    f = L.mskfh_two_point_ransac
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    f.restype = None
    calib, px1, px2, R_c_p = _scene(oracle, seed, translation=translation)
    K, D = np.array(calib.cam0_intrinsics), np.array(calib.cam0_distortion)
    ref, ref_draws = oracle.two_point_ransac(calib, default_fe_cfg(), 0, px1, px2, R_c_p, draws=100 * seed)
    u1 = np.ascontiguousarray(oracle.undistort(K, D, px1), dtype=np.float32)
    u2 = np.ascontiguousarray(oracle.undistort(K, D, px2), dtype=np.float32)
    R = np.ascontiguousarray(R_c_p, dtype=np.float64)
    got = np.zeros(len(px1), np.int32)
    draws = C.c_ulonglong(100 * seed)
    f(len(px1), u1.ctypes.data, u2.ctypes.data, R.ctypes.data, K.ctypes.data, 3.0, 0.99, C.byref(draws), got.ctypes.data)
    assert draws.value == ref_draws
    assert np.array_equal(got, ref)
    assert 0 < got.sum() < len(got)


@pytest.mark.parametrize("seed,translation", [(3, (0.04, -0.01, 0.02)), (4, (0.0, 0.0, 0.0)), (5, (0.2, 0.1, -0.05)), (6, (0.002, 0.0, 0.001)), (7, (0.01, 0.03, 0.0))])
def test_device_book_ransac_equals_oracle(oracle, tmp_path, seed, translation):
    """The DEVICE source of the RANSAC (fb_two_point_ransac in csrc/hip/fe_book.h: what the bookkeeping kernel of a device frame
    runs between the track calls when MSKF_COMPAT_Q5_NO_RANSAC is cleared) executed on the CPU == oracle, marker for marker,
    draw counter included: general motion, pure rotation (degenerate branch), large and tiny translations, and the small inputs
    (0, 2 pairs).  On the device the per-pair and per-hypothesis phases run in parallel; the two index-order sums
    (rescalePoints, mean length) are summed by one item in that order."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libfe_book_ransac.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I", root, "-o", so,
                           os.path.join(root, "tests", "cpp", "fe_book_ransac.cpp")])
    f = C.CDLL(so).fb_ransac_run
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = None
    calib, px1, px2, R_c_p = _scene(oracle, seed, translation=translation)
    K, D = np.array(calib.cam0_intrinsics), np.array(calib.cam0_distortion)
    R = np.ascontiguousarray(R_c_p, dtype=np.float64)
    for n in (len(px1), 57, 2, 0):
        ref, ref_draws = oracle.two_point_ransac(calib, default_fe_cfg(), 0, px1[:n], px2[:n], R_c_p, draws=1000 * seed + n)
        u1 = np.ascontiguousarray(oracle.undistort(K, D, px1[:n]), dtype=np.float32).reshape(-1, 2)
        u2 = np.ascontiguousarray(oracle.undistort(K, D, px2[:n]), dtype=np.float32).reshape(-1, 2)
        got = np.zeros(max(n, 1), np.int32)
        draws = C.c_ulonglong(1000 * seed + n)
        f(n, u1.ctypes.data, u2.ctypes.data, R.ctypes.data, float(K[0]), float(K[1]), 3.0, 7, C.byref(draws), got.ctypes.data)
        assert draws.value == ref_draws, (n, draws.value, ref_draws)
        assert np.array_equal(got[:n], ref), (n, int(got[:n].sum()), int(ref.sum()))
