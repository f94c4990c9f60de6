"""Dev aid: EKF update parity (GPU vs oracle) over clone counts; prints the worst relative errors per size."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from oracle import oracle_py as O
from msckf_stereo_c_amd import capi
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
import ekf_problems

ctx = capi.Context(0)
calib = O.euroc_calib(376, 240)
nf_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for n_clones in range(3, 31):
    for dof in (-1, 0):
        cfg = default_ekf_cfg(max_cam_state_size=max(n_clones, 4))
        s = capi.Stream(ctx, calib, default_fe_cfg(), cfg)
        pr = ekf_problems.make_problem(calib, seed=100 + n_clones, n_clones=n_clones, n_feat=(nf_arg or 3 * n_clones), min_obs=min(3, n_clones))
        ref = O.ekf_update_problem(calib, cfg, pr["gravity"], pr["clones"], pr["P"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], dof)
        s.ekf_set_cov(pr["P"])
        got = s.ekf_update(pr["gravity"], pr["clones"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], dof, apply_row_cap=(dof < 0))
        Pg = s.ekf_get_cov()
        eP = np.abs(Pg - ref["P"]).max() / np.abs(ref["P"]).max()
        edx = np.abs(got["delta_x"] - ref["delta_x"]).max() / max(np.abs(ref["delta_x"]).max(), 1e-12)
        flag = "  <<<<" if (eP > 1e-6 or edx > 1e-5 or got["rows"] != ref["rows"]) else ""
        print(f"clones {n_clones:2d} dof {dof:2d} rows {got['rows']:5d}/{ref['rows']:5d} eP {eP:.2e} edx {edx:.2e}{flag}")
        s.close()
