"""Known-answer tests that pin the CPU oracle: the reference ships no golden vectors,
so these are analytic KATs (plus tests/golden/ regression fixtures in test_golden.py)."""
import numpy as np
import pytest

from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg

import ekf_problems


# ------------------------------------------------------------------ pyr_down
def test_pyr_down_constant_ramp_impulse(oracle):
    c = np.full((48, 64), 137, np.uint8)
    assert np.all(oracle.pyr_down(c) == 137)
    yy, xx = np.mgrid[0:48, 0:64]
    ramp = (2 * xx + yy).astype(np.uint8)
    out = oracle.pyr_down(ramp)
    assert out.shape == (24, 32)
    # interior of a linear ramp is reproduced exactly at the sampled positions (kernel is symmetric, sums to 256)
    assert np.array_equal(out[2:-2, 2:-2], ramp[::2, ::2][2:-2, 2:-2])
    imp = np.zeros((33, 41), np.uint8)
    imp[16, 20] = 255
    out = oracle.pyr_down(imp)
    assert out.shape == (17, 21)
    k = np.array([1, 4, 6, 4, 1])
    exp = (np.outer(k, k)[::2, ::2] * 255 + 128) >> 8   # taps that land on even source offsets
    assert np.array_equal(out[7:10, 9:12], exp)
    assert out.sum() == exp.sum()


def test_pyr_down_matches_numpy_reference(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")   # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    ref = np.zeros(((h + 1) // 2, (w + 1) // 2), np.int64)
    for y in range(ref.shape[0]):
        for x in range(ref.shape[1]):
            ref[y, x] = (k[:, None] * k[None, :] * pad[2 * y:2 * y + 5, 2 * x:2 * x + 5]).sum()
    ref = ((ref + 128) >> 8).astype(np.uint8)
    assert np.array_equal(oracle.pyr_down(img), ref)


# ------------------------------------------------------------------ LK
def smooth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w))
    for _ in range(40):
        fx, fy, ph = rng.uniform(0.02, 0.25), rng.uniform(0.02, 0.25), rng.uniform(0, 6.28)
        img += rng.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + ph)
    return img


def shifted_pair(h, w, dx, dy, seed=3):
    """Two renderings of the same analytic image, the second shifted by a known sub-pixel amount."""
    rng = np.random.default_rng(seed)
    params = [(rng.uniform(0.05, 0.5), rng.uniform(0.05, 0.5), rng.uniform(0, 6.28), rng.uniform(0.3, 1.0)) for _ in range(40)]

    def render(ox, oy):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = np.zeros((h, w))
        for fx, fy, ph, a in params:
            img += a * np.sin(fx * (xx - ox) + fy * (yy - oy) + ph)
        return np.round(np.clip((img + 12) / 24, 0, 1) * 255).astype(np.uint8)
    return render(0, 0), render(dx, dy)


@pytest.mark.parametrize("dx,dy", [(0.0, 0.0), (1.3, -0.7), (3.6, 2.2), (-5.25, 4.5)])
def test_lk_recovers_known_shift(oracle, dx, dy):
    a, b = shifted_pair(240, 320, dx, dy)
    pts = np.array([[x, y] for y in range(40, 200, 20) for x in range(40, 280, 24)], np.float32)
    out, st = oracle.lk_track(a, b, pts, pts.copy())
    assert st.all()
    err = out - (pts + np.array([dx, dy], np.float32))
    assert np.median(np.abs(err)) < 0.05 and np.abs(err).max() < 0.25, (np.median(np.abs(err)), np.abs(err).max())


def test_lk_status_rules(oracle):
    a, b = shifted_pair(120, 160, 0.5, 0.25)
    flat = np.full_like(a, 90)
    pts = np.array([[80, 60], [-40, 60], [80, 400]], np.float32)
    _, st = oracle.lk_track(a, b, pts, pts.copy())
    assert list(st) == [1, 0, 0]                      # far outside -> level-0 bounds rule
    _, st = oracle.lk_track(flat, flat, pts[:1], pts[:1].copy())
    assert st[0] == 0                                 # no texture -> min-eigenvalue rule


# ------------------------------------------------------------------ detector
def test_detector_single_corner_and_occupancy(oracle):
    img = np.full((240, 376), 40, np.uint8)
    img[100:140, 200:260] = 220          # bright rectangle: 4 strong corners
    pts, resp = oracle.detect(img, thr=10)
    assert len(pts) >= 4
    corners = np.array([[200, 100], [259, 100], [200, 139], [259, 139]], np.float32)
    for c in corners:
        assert np.min(np.abs(pts - c).sum(1)) <= 8   # the 8x8 box puts the maximum a few pixels inside the corner
    # occupancy masks exactly the marked cells
    mx = oracle.cell_maxima(img)
    cell_of_first = int(mx["cell"][np.argmax(mx["score"])])
    occ = np.zeros(30 * 47, np.uint8)
    occ[cell_of_first] = 1
    pts2, _ = oracle.detect(img, occupancy=occ)
    assert len(pts2) == len(pts) - 1
    # flat image: nothing
    assert len(oracle.detect(np.full((240, 376), 128, np.uint8))[0]) == 0
    # score is integer Shi-Tomasi: response = score / 256
    assert np.allclose(resp, mx["score"][mx["score"] > 2560] / 256.0)


# ------------------------------------------------------------------ point math
def test_undistort_distort_round_trip(oracle):
    calib = oracle.euroc_calib(752, 480)
    K, D = np.array(calib.cam0_intrinsics), np.array(calib.cam0_distortion)
    rng = np.random.default_rng(1)
    px = np.stack([rng.uniform(20, 730, 500), rng.uniform(20, 460, 500)], 1).astype(np.float32)
    n = oracle.undistort(K, D, px)
    back = oracle.distort(K, D, n)
    # cv::undistortPoints' 5 fixed-point iterations converge geometrically: tight near the centre, ~0.3 px in the corners
    central = (np.abs(px[:, 0] - K[2]) < 200) & (np.abs(px[:, 1] - K[3]) < 150)
    assert np.abs(back - px)[central].max() < 5e-3
    assert np.abs(back - px).max() < 0.5
    # identity distortion: undistort is the pinhole normalisation
    n0 = oracle.undistort(K, np.zeros(4), px)
    assert np.allclose(n0, (px - K[2:]) / K[:2], atol=1e-6)
    # rectification rotation is applied after undistortion
    th = 0.01
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    nr = oracle.undistort(K, D, px, R=Rz)
    assert np.allclose(nr, (Rz[:2, :2] @ n.T.astype(np.float64)).T, atol=1e-6)


# ------------------------------------------------------------------ EKF algebra vs numpy
def numpy_update(calib, pr, sigma2, chi2, dof_offset):
    """Dense numpy/scipy restatement of featureJacobian + gatingTest + measurementUpdate with an SVD null space
    (the form the reference's commented-out Eigen code uses, msckf_vio.cpp:735-736)."""
    from ekf_problems import quat_to_rot
    T01 = np.array(calib.T_cam1_cam0).reshape(4, 4)
    R01, t01 = T01[:3, :3], T01[:3, 3]
    g = pr["gravity"]
    n_clones = len(pr["clones"])
    d = 21 + 6 * n_clones
    P = pr["P"].copy()

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    Hs, rs, gammas, passed = [], [], [], []
    for j in range(len(pr["obs_start"]) - 1):
        obs = range(pr["obs_start"][j], pr["obs_start"][j + 1])
        pw = pr["positions"][j]
        Hx, Hf, r = [], [], []
        for o in obs:
            c = pr["clones"][pr["obs_clone"][o]]
            Rw0 = quat_to_rot(c[0:4])
            Rw1 = R01 @ Rw0
            t1 = c[4:7] - Rw1.T @ t01
            p0 = Rw0 @ (pw - c[4:7])
            p1 = Rw1 @ (pw - t1)
            dz0 = np.zeros((4, 3)); dz1 = np.zeros((4, 3))
            dz0[0, 0] = dz0[1, 1] = 1 / p0[2]; dz0[0, 2] = -p0[0] / p0[2] ** 2; dz0[1, 2] = -p0[1] / p0[2] ** 2
            dz1[2, 0] = dz1[3, 1] = 1 / p1[2]; dz1[2, 2] = -p1[0] / p1[2] ** 2; dz1[3, 2] = -p1[1] / p1[2] ** 2
            d0 = np.hstack([skew(p0), -Rw0]); d1 = np.hstack([R01 @ skew(p0), -Rw1])
            A = dz0 @ d0 + dz1 @ d1
            u = np.concatenate([quat_to_rot(c[7:11]) @ g, skew(pw - c[11:14]) @ g])
            hx = A - np.outer(A @ u, u) / (u @ u)
            hf = -hx[:, 3:6]
            z = pr["obs_z"][o]
            row = np.zeros((4, d)); row[:, 21 + 6 * pr["obs_clone"][o]: 27 + 6 * pr["obs_clone"][o]] = hx
            Hx.append(row); Hf.append(hf)
            r.append(z - np.array([p0[0] / p0[2], p0[1] / p0[2], p1[0] / p1[2], p1[1] / p1[2]]))
        Hx, Hf, r = np.vstack(Hx), np.vstack(Hf), np.concatenate(r)
        U = np.linalg.svd(Hf, full_matrices=True)[0]
        A = U[:, 3:]
        Ho, ro = A.T @ Hx, A.T @ r
        S = Ho @ P @ Ho.T + sigma2 * np.eye(len(ro))
        gamma = ro @ np.linalg.solve(S, ro)
        gammas.append(gamma)
        ok = gamma < chi2[len(list(obs)) + dof_offset - 1]
        passed.append(ok)
        if ok:
            Hs.append(Ho); rs.append(ro)
    H, r = np.vstack(Hs), np.concatenate(rs)
    if H.shape[0] > H.shape[1]:
        import scipy.linalg
        Q, Rm = scipy.linalg.qr(H, mode="economic")
        H, r = Rm, Q.T @ r
    S = H @ P @ H.T + sigma2 * np.eye(H.shape[0])
    K = np.linalg.solve(S, H @ P).T
    dx = K @ r
    Pn = (np.eye(d) - K @ H) @ P
    return dict(gamma=np.array(gammas), passed=np.array(passed), delta_x=dx, P=(Pn + Pn.T) / 2, rows=sum(len(x) for x in rs))


@pytest.mark.parametrize("n_clones,n_feat,seed", [(6, 6, 1), (12, 25, 2)])
def test_ekf_update_oracle_vs_numpy(oracle, n_clones, n_feat, seed):
    from scipy.stats import chi2 as chi2dist
    calib = oracle.euroc_calib(376, 240)
    cfg = default_ekf_cfg(max_cam_state_size=n_clones)
    pr = ekf_problems.make_problem(calib, seed=seed, n_clones=n_clones, n_feat=n_feat)
    ref = numpy_update(calib, pr, cfg.noise_feature ** 2, chi2dist.ppf(0.05, np.arange(1, 100)), -1)
    got = oracle.ekf_update_problem(calib, cfg, pr["gravity"], pr["clones"], pr["P"], pr["positions"], pr["obs_start"],
                                    pr["obs_clone"], pr["obs_z"], -1)
    assert np.allclose(got["gamma"], ref["gamma"], rtol=1e-8)       # null-space basis invariance (SVD vs Householder)
    assert np.array_equal(got["passed"].astype(bool), ref["passed"])
    assert got["rows"] == ref["rows"]
    assert np.allclose(got["delta_x"], ref["delta_x"], rtol=1e-6, atol=1e-10)
    assert np.allclose(got["P"], ref["P"], rtol=1e-8, atol=1e-13)


def test_triangulation_recovers_points(oracle):
    calib = oracle.euroc_calib(376, 240)
    pr = ekf_problems.make_problem(calib, seed=4, n_clones=10, n_feat=30, noise=0.0)
    # ground truth positions are pr["positions"] minus the perturbation: regenerate exactly
    pr0 = ekf_problems.make_problem(calib, seed=4, n_clones=10, n_feat=30, noise=0.0)
    pos, valid = oracle.triangulate(calib, pr["clones"], pr["obs_start"], pr["obs_clone"], pr["obs_z"])
    assert valid.all()
    assert np.abs(pos - pr0["positions"]).max() < 0.05   # positions carry a 1 cm sigma perturbation in the generator


# ------------------------------------------------------------------ end to end vs ground truth
def align_rigid(A, B):
    ca, cb = A.mean(0), B.mean(0)
    U, _, Vt = np.linalg.svd((A - ca).T @ (B - cb))
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    Rm = Vt.T @ D @ U.T
    return (Rm @ (A - ca).T).T + cb


def test_oracle_tracks_ground_truth(oracle):
    """The oracle is a working VIO: on the synthetic stream its trajectory stays within centimetres of ground truth."""
    syn = oracle.Synth(width=376, height=240)
    sysm = oracle.OracleSystem(syn.calib, default_fe_cfg(), default_ekf_cfg())
    n = 110
    syn.feed(sysm, n)
    poses = sysm.poses()
    assert len(poses) > 80 and sysm.num_updates() > 20 and sysm.num_resets() == 0
    k0 = n - len(poses)
    gt = np.array([syn.gt_pose(k)["p"] for k in range(k0, n)])
    err = np.linalg.norm(align_rigid(poses["p"], gt) - gt, axis=1)
    assert np.sqrt((err ** 2).mean()) < 0.03
    ids, life, c0, c1, info = sysm.dump()
    assert 55 <= len(ids) <= 85 and info.after_matching >= 50       # 4x5 grid, 3..4 features per cell
    # Q1: the feature message is never cleared
    assert len(sysm.msg()) > 60 * (n - 5)


def test_deterministic_trig_and_fisheye_round_trip(oracle):
    """The equidistant model uses a deterministic atan / tan (oracle/o_math.h, same operation sequence on the device):
    distort(undistort(p)) returns p, and against numpy's libm the model functions agree to ~1e-7 px / 1e-15 relative."""
    K = np.array([458.654, 457.296, 367.215, 248.375])
    D = np.array([-0.013, 0.021, -0.008, 0.0015])
    rng = np.random.default_rng(8)
    pts = np.stack([rng.uniform(0, 752, 4000), rng.uniform(0, 480, 4000)], 1).astype(np.float32)
    und = oracle.undistort(K, D, pts, model=1)
    back = oracle.distort(K, D, und, model=1)
    assert np.abs(back - pts).max() < 2e-3                      # float32 in/out around 1e3 px
    # reference evaluation with libm (cv::fisheye::distortPoints)
    x, y = und[:, 0].astype(np.float64), und[:, 1].astype(np.float64)
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    thd = th * (1 + D[0] * th ** 2 + D[1] * th ** 4 + D[2] * th ** 6 + D[3] * th ** 8)
    sc = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
    ref = np.stack([x * sc * K[0] + K[2], y * sc * K[1] + K[3]], 1)
    assert np.abs(ref - back.astype(np.float64)).max() < 1e-4   # float32 output rounding only
    # wide angles: rays up to ~80 degrees off axis
    wide = np.stack([np.linspace(-5.5, 5.5, 401), np.linspace(5.0, -5.0, 401)], 1).astype(np.float32)
    d = oracle.distort(K, D, wide, model=1).astype(np.float64)
    x, y = wide[:, 0].astype(np.float64), wide[:, 1].astype(np.float64)
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    thd = th * (1 + D[0] * th ** 2 + D[1] * th ** 4 + D[2] * th ** 6 + D[3] * th ** 8)
    sc = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
    ref = np.stack([x * sc * K[0] + K[2], y * sc * K[1] + K[3]], 1)
    assert np.abs(ref - d).max() < 1e-3 * (1 + np.abs(ref).max() / 1e4)
