#!/usr/bin/env python3
"""Absolute trajectory error (ATE RMSE) between two TUM-format trajectories — the evaluation the reference's README
points to (rgbd_benchmark_tools' evaluate_ate: associate by time stamp, Horn/Umeyama rigid alignment, RMSE of the
translational residuals).  pose_out.txt as written by MsckfVio::publish (msckf_vio.cpp:1256-1258) is TUM format:
    t  px py pz  qx qy qz qw

usage: tools/ate_rmse.py estimated.txt reference.txt [--max-dt 0.01] [--scale]
Also importable: ate_rmse(est_t, est_p, ref_t, ref_p, max_dt=0.01, with_scale=False) -> dict.
"""
import argparse
import sys

import numpy as np


def read_tum(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        v = line.replace(",", " ").split()
        if len(v) >= 4:
            rows.append([float(x) for x in v[:4]])
    a = np.array(rows, dtype=np.float64).reshape(-1, 4)
    return a[:, 0], a[:, 1:4]


def associate(t_est, t_ref, max_dt):
    """Greedy nearest-stamp association (one-to-one, |dt| <= max_dt), like associate.py of the TUM tools."""
    cand = []
    j0 = 0
    order = np.argsort(t_ref)
    tr = t_ref[order]
    for i, t in enumerate(t_est):
        j = np.searchsorted(tr, t)
        for jj in (j - 1, j):
            if 0 <= jj < len(tr) and abs(tr[jj] - t) <= max_dt:
                cand.append((abs(tr[jj] - t), i, int(order[jj])))
    cand.sort()
    used_e, used_r, pairs = set(), set(), []
    for _, i, j in cand:
        if i in used_e or j in used_r:
            continue
        used_e.add(i); used_r.add(j); pairs.append((i, j))
    pairs.sort()
    return np.array(pairs, dtype=np.int64).reshape(-1, 2)


def umeyama(A, B, with_scale=False):
    """R, t, s minimising sum |s R a + t - b|^2 (Umeyama 1991; s = 1 gives Horn's rigid alignment)."""
    ca, cb = A.mean(0), B.mean(0)
    Ac, Bc = A - ca, B - cb
    U, S, Vt = np.linalg.svd(Bc.T @ Ac / len(A))
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    s = float((S * np.diag(D)).sum() / (Ac ** 2).sum() * len(A)) if with_scale else 1.0
    t = cb - s * R @ ca
    return R, t, s


def ate_rmse(est_t, est_p, ref_t, ref_p, max_dt=0.01, with_scale=False):
    pairs = associate(np.asarray(est_t), np.asarray(ref_t), max_dt)
    if len(pairs) < 3:
        raise ValueError("fewer than 3 associated poses")
    A, B = np.asarray(est_p)[pairs[:, 0]], np.asarray(ref_p)[pairs[:, 1]]
    R, t, s = umeyama(A, B, with_scale)
    err = np.linalg.norm((s * (R @ A.T)).T + t - B, axis=1)
    return {"pairs": int(len(pairs)), "rmse": float(np.sqrt((err ** 2).mean())), "mean": float(err.mean()),
            "median": float(np.median(err)), "max": float(err.max()), "scale": s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("estimated")
    ap.add_argument("reference")
    ap.add_argument("--max-dt", type=float, default=0.01)
    ap.add_argument("--scale", action="store_true")
    a = ap.parse_args()
    te, pe = read_tum(a.estimated)
    tr, pr = read_tum(a.reference)
    r = ate_rmse(te, pe, tr, pr, a.max_dt, a.scale)
    print("compared_pose_pairs %d\nabsolute_translational_error.rmse %.6f m\n.mean %.6f m\n.median %.6f m\n.max %.6f m\nscale %.6f"
          % (r["pairs"], r["rmse"], r["mean"], r["median"], r["max"], r["scale"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
