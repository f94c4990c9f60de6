"""Dev aid (CPU only): accuracy of the covariance update formulas on one synthetic problem, against a
long-double (80-bit) evaluation of P' = P - P H^T (H P H^T + s2 I)^-1 H P on the oracle's stacked (H, r)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from oracle import oracle_py as O
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg
import ekf_problems

LD = np.longdouble

def chol_ld(A):
    A = A.copy(); n = A.shape[0]
    for j in range(n):
        A[j, j] = np.sqrt(A[j, j] - np.dot(A[j, :j], A[j, :j]))
        if j + 1 < n:
            A[j + 1:, j] = (A[j + 1:, j] - A[j + 1:, :j] @ A[j, :j]) / A[j, j]
    return np.tril(A)

def fwd_ld(L, B):
    Y = B.copy(); n = L.shape[0]
    for i in range(n):
        Y[i] = (Y[i] - L[i, :i] @ Y[:i]) / L[i, i]
    return Y

n_clones = int(sys.argv[1]) if len(sys.argv) > 1 else 29
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 4
calib = O.euroc_calib(376, 240)
cfg = default_ekf_cfg(max_cam_state_size=max(n_clones, 4))
pr = ekf_problems.make_problem(calib, seed=100 + n_clones, n_clones=n_clones, n_feat=n_feat, min_obs=3)
ref = O.ekf_update_problem(calib, cfg, pr["gravity"], pr["clones"], pr["P"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], -1)
L = O.lib()
L.orc_ekf_last_system.restype = C.c_int
rows = L.orc_ekf_last_system(None, None, 0)
d = 21 + 6 * n_clones
H = np.zeros((rows, d)); r = np.zeros(rows)
L.orc_ekf_last_system(H.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), rows)
s2 = cfg.noise_feature ** 2
print("rows", rows, "d", d, "sigma2", s2, "cond(H) nonzero sv:", end=" ")
sv = np.linalg.svd(H, compute_uv=False); print(sv[0], sv[sv > sv[0] * 1e-14][-1], (sv > sv[0] * 1e-12).sum())
P = pr["P"]
# truth in long double (measurement space, no compression)
Hl, Pl = H.astype(LD), P.astype(LD)
T = Hl @ Pl
S = T @ Hl.T + LD(s2) * np.eye(rows, dtype=LD)
Ls = chol_ld(S)
Y = fwd_ld(Ls, T)
Pt = (Pl - Y.T @ Y).astype(np.float64)
# (a) reference/oracle formula result
eo = np.abs(ref["P"] - Pt).max() / np.abs(Pt).max()
# (b) Gram + semidefinite Cholesky + square-root downdate in float64 (the GPU algebra)
G = H.T @ H
n = d
Lg = np.zeros((n, n)); A = G.copy(); tol = A.diagonal().max() * n * 2.220446049250313e-16
for j in range(n):
    piv = A[j, j]
    if not piv > tol:
        A[j:, j] = 0; continue
    l = np.sqrt(piv); Lg[j, j] = l; Lg[j + 1:, j] = A[j + 1:, j] / l
    A[j + 1:, j + 1:] -= np.outer(Lg[j + 1:, j], Lg[j + 1:, j])
R = Lg.T
Tg = R @ P
Sg = Tg @ R.T + s2 * np.eye(n)
Lsg = np.linalg.cholesky(Sg)
Yg = np.linalg.solve(Lsg, Tg)      # not triangular-aware but accurate enough
Pg = P - Yg.T @ Yg
eg = np.abs(Pg - Pt).max() / np.abs(Pt).max()
print(f"vs long-double truth: oracle (I-KH)P {eo:.2e}   Gram/Cholesky float64 {eg:.2e}   oracle-vs-gram {np.abs(Pg-ref['P']).max()/np.abs(Pt).max():.2e}")

# (c) information form with float64 Gram: P' = P - P (s2 I + G P)^-1 G P
GP = G @ P
Pi = P - P @ np.linalg.solve(s2 * np.eye(n) + GP, GP)
Pi = (Pi + Pi.T) / 2
print(f"information form (float64 G): {np.abs(Pi - Pt).max() / np.abs(Pt).max():.2e}")
# (d) symmetric information form through chol(P): P' = Lp (I + Lp^T G Lp / s2)^-1 Lp^T
Lp = np.linalg.cholesky(P)
M = np.eye(n) + Lp.T @ G @ Lp / s2
Lm = np.linalg.cholesky(M)
Z = np.linalg.solve(Lm, Lp.T)
Pd = Z.T @ Z
print(f"chol(P) information form: {np.abs(Pd - Pt).max() / np.abs(Pt).max():.2e}")
# (e) Gram/Cholesky with different pivot tolerances
for fac in (1.0, 1e2, 1e4, 1e6):
    Lg = np.zeros((n, n)); A = G.copy(); tol2 = tol * fac; kept = 0
    for j in range(n):
        piv = A[j, j]
        if not piv > tol2:
            A[j:, j] = 0; continue
        kept += 1
        l = np.sqrt(piv); Lg[j, j] = l; Lg[j + 1:, j] = A[j + 1:, j] / l
        A[j + 1:, j + 1:] -= np.outer(Lg[j + 1:, j], Lg[j + 1:, j])
    R = Lg.T; Tg = R @ P; Sg = Tg @ R.T + s2 * np.eye(n)
    Yg = np.linalg.solve(np.linalg.cholesky(Sg), Tg)
    Pg2 = P - Yg.T @ Yg
    print(f"  tol x{fac:g}: kept {kept} err {np.abs(Pg2 - Pt).max() / np.abs(Pt).max():.2e}")
# (f) regularised Cholesky: R^T R = G + lam I (no pivot skipping at all)
g = G.diagonal().max()
for rel in (1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10):
    lam = rel * g * n
    try:
        Lr = np.linalg.cholesky(G + lam * np.eye(n))
    except np.linalg.LinAlgError:
        print(f"  lam {rel:g}*n*g: not PD"); continue
    R = Lr.T; Tg = R @ P; Sg = Tg @ R.T + s2 * np.eye(n)
    Yg = np.linalg.solve(np.linalg.cholesky(Sg), Tg)
    Pg3 = P - Yg.T @ Yg
    # delta_x
    w = np.linalg.solve(np.linalg.cholesky(Sg), np.linalg.solve(Lr, H.T @ r))
    dx = Yg.T @ w
    Kt = np.linalg.solve((Hl @ Pl @ Hl.T + LD(s2) * np.eye(rows, dtype=LD)).astype(np.float64), (H @ P))
    dxt = Kt.T @ r
    print(f"  lam {rel:g}*n*g: P err {np.abs(Pg3 - Pt).max() / np.abs(Pt).max():.2e}  dx err {np.abs(dx - dxt).max() / np.abs(dxt).max():.2e}")
