import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from msckf_stereo_c_amd import capi
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
from oracle import oracle_py as O
import ekf_problems
O.build()
ctx = capi.Context(0)
def run(n_clones, n_feat, seed, mode, dof=-1, cap=True, **kw):
    calib = O.euroc_calib(376, 240)
    cfg = default_ekf_cfg(max_cam_state_size=max(n_clones,4), compression_mode=mode)
    s = capi.Stream(ctx, calib, default_fe_cfg(), cfg)
    pr = ekf_problems.make_problem(calib, seed=seed, n_clones=n_clones, n_feat=n_feat, **kw)
    ref = O.ekf_update_problem(calib, cfg, pr["gravity"], pr["clones"], pr["P"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], dof)
    s.ekf_set_cov(pr["P"])
    got = s.ekf_update(pr["gravity"], pr["clones"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], dof, cap)
    Pg = s.ekf_get_cov()
    eP = np.abs(Pg - ref["P"]).max() / np.abs(ref["P"]).max()
    ex = np.abs(got["delta_x"] - ref["delta_x"]).max() / max(np.abs(ref["delta_x"]).max(), 1e-30)
    s.close()
    return got["rows"], ref["rows"], got["used_qr"], got["tiny_pivots"], eP, ex
for (nc, nf, seed, kw) in [(6,5,1,{}),(20,30,2,{}),(30,60,3,{}),(29,4,129,dict(min_obs=3)),(13,4,113,dict(min_obs=3)),(10,2,110,dict(min_obs=3)),(19,3,119,dict(min_obs=3)),(50,40,21,dict(min_obs=20)),(30,400,31,dict(pair=(3,4),noise=0.004)),(12,3,5,dict(pair=(0,1)))]:
    for mode in (1, 2, 0):
        dof, cap = (0, False) if 'pair' in kw else (-1, True)
        print(nc, nf, kw, "mode", mode, "rows %d/%d qr %d tiny %d  errP %.2e errdx %.2e" % run(nc, nf, seed, mode, dof, cap, **kw), flush=True)
