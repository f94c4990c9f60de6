#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference ships no golden vectors and cannot be built here, so these fixtures pin
the ORACLE (and through it the HIP path) against regressions; they are not outputs of the reference.
Run from the repo root: python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O  # noqa: E402
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg  # noqa: E402
import ekf_problems  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# ---- front-end kernels on a small synthetic pair (188x120 keeps the fixture small)
W, H = 188, 120
syn = O.Synth(seed=0x5EED00AA, width=W, height=H)
a0, b0 = syn.render(30)
a1, b1 = syn.render(31)
pyr = O.build_pyramid(a1)
pts, resp = O.detect(a0, det_rows=15, det_cols=24)
lk_out, lk_st = O.lk_track(a0, a1, pts, pts.copy())
fe = default_fe_cfg()
fe.det_rows, fe.det_cols = 15, 24
sm_out, sm_in = O.stereo_match(syn.calib, fe, a1, b1, pts)
mx = O.cell_maxima(a0, 15, 24)
np.savez_compressed(os.path.join(OUT, "frontend_188x120.npz"), a0=a0, b0=b0, a1=a1, b1=b1, l1=pyr[1], l2=pyr[2], l3=pyr[3],
                    det_pts=pts, det_resp=resp, cell_score=mx["score"], cell_x=mx["x"], cell_y=mx["y"],
                    lk_out=lk_out, lk_status=lk_st, stereo_out=sm_out, stereo_inlier=sm_in)

# ---- EKF update problem
calib = O.euroc_calib(376, 240)
cfg = default_ekf_cfg(max_cam_state_size=10)
pr = ekf_problems.make_problem(calib, seed=77, n_clones=10, n_feat=16)
ref = O.ekf_update_problem(calib, cfg, pr["gravity"], pr["clones"], pr["P"], pr["positions"], pr["obs_start"], pr["obs_clone"],
                           pr["obs_z"], -1)
pos, valid = O.triangulate(calib, pr["clones"], pr["obs_start"], pr["obs_clone"], pr["obs_z"])
np.savez_compressed(os.path.join(OUT, "ekf_update_10x16.npz"), clones=pr["clones"], P=pr["P"], positions=pr["positions"],
                    obs_start=pr["obs_start"], obs_clone=pr["obs_clone"], obs_z=pr["obs_z"], gravity=pr["gravity"],
                    gamma=ref["gamma"], passed=ref["passed"], delta_x=ref["delta_x"], P_new=ref["P"], rows=ref["rows"],
                    tri_pos=pos, tri_valid=valid)

# ---- end-to-end: 60 frames of one 188x120 stream (ids, pixels of the last frame, all poses)
syn = O.Synth(seed=0x5EED00AB, width=W, height=H)
fe = default_fe_cfg()
fe.det_rows, fe.det_cols = 15, 24
osys = O.OracleSystem(syn.calib, fe, default_ekf_cfg())
ids_per_frame, n_per_frame = [], []


def cb(k, s):
    ids, life, c0, c1, info = s.dump()
    ids_per_frame.append(ids.copy())
    n_per_frame.append(len(ids))


syn.feed(osys, 60, cb)
ids, life, c0, c1, info = osys.dump()
poses = osys.poses()
np.savez_compressed(os.path.join(OUT, "system_188x120_60f.npz"), ids_concat=np.concatenate(ids_per_frame), n_per_frame=np.array(n_per_frame),
                    last_c0=np.stack([c0["x"], c0["y"]], 1), last_c1=np.stack([c1["x"], c1["y"]], 1), pose_t=poses["t"],
                    pose_p=poses["p"], pose_q=poses["q"], n_updates=osys.num_updates())
print("golden fixtures written to", OUT, [(f, os.path.getsize(os.path.join(OUT, f))) for f in sorted(os.listdir(OUT))])
