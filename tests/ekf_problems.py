"""Seeded synthetic EKF update problems (clone poses, features, observations, covariance) for parity tests."""
import numpy as np


def quat_to_rot(q):
    """JPL quaternion [x y z w] -> R (world -> body), SURVEY Appendix C."""
    x, y, z, w = q
    qv = np.array([x, y, z])
    sk = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return (2 * w * w - 1) * np.eye(3) - 2 * w * sk + 2 * np.outer(qv, qv)


def small_quat(rng, scale):
    v = rng.normal(size=3) * scale
    q = np.array([v[0] / 2, v[1] / 2, v[2] / 2, 1.0])
    return q / np.linalg.norm(q)


def make_problem(calib, seed=0, n_clones=10, n_feat=12, min_obs=3, noise=0.002, null_perturb=1e-3, p_scale=1e-3, pair=None,
                 baseline_scale=1.0, depth_scale=1.0):
    """pair = (ka, kb): every feature is observed by exactly these two clones (the pruning update's shape).
    baseline_scale < 1 shrinks the camera motion between clones, depth_scale < 1 moves the features closer: both drive
    the stacked Jacobian towards ill conditioning / large entries (condition sweep of the QR-compression tests)."""
    rng = np.random.default_rng(seed)
    T01 = np.array(calib.T_cam1_cam0).reshape(4, 4)
    R01, t01 = T01[:3, :3], T01[:3, 3]
    clones = np.zeros((n_clones, 14))
    for i in range(n_clones):
        q = small_quat(rng, 0.15)
        p = (np.array([0.05 * i, 0.02 * np.sin(i), 0.03 * np.cos(i)]) + rng.normal(size=3) * 0.01) * baseline_scale
        qn = q + rng.normal(size=4) * null_perturb
        qn /= np.linalg.norm(qn)
        pn = p + rng.normal(size=3) * null_perturb
        clones[i] = np.concatenate([q, p, qn, pn])
    positions, obs_start, obs_clone, obs_z = [], [0], [], []
    for j in range(n_feat):
        pw = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(3.0, 8.0)]) * depth_scale
        m = int(rng.integers(min_obs, n_clones + 1))
        start = int(rng.integers(0, n_clones - m + 1))
        for ci in (range(start, start + m) if pair is None else pair):
            R = quat_to_rot(clones[ci, 0:4])
            pc0 = R @ (pw - clones[ci, 4:7])
            pc1 = R01 @ pc0 + t01
            z = np.array([pc0[0] / pc0[2], pc0[1] / pc0[2], pc1[0] / pc1[2], pc1[1] / pc1[2]]) + rng.normal(size=4) * noise
            obs_clone.append(ci)
            obs_z.append(z)
        obs_start.append(len(obs_clone))
        positions.append(pw + rng.normal(size=3) * 0.01 * depth_scale)
    d = 21 + 6 * n_clones
    A = rng.normal(size=(d, d))
    P = p_scale * (A @ A.T / d + np.eye(d))
    P = (P + P.T) / 2
    return dict(clones=clones, positions=np.array(positions), obs_start=np.array(obs_start, np.int32),
                obs_clone=np.array(obs_clone, np.int32), obs_z=np.array(obs_z), P=P, gravity=np.array([0.0, 0.0, -9.81]))
