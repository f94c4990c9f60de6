import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from msckf_stereo_c_amd import capi
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
from oracle import oracle_py as O
import ekf_problems
O.build()
ctx = capi.Context(0)
def run(n_clones, n_feat, seed, mode, **kw):
    calib = O.euroc_calib(376, 240)
    cfg = default_ekf_cfg(max_cam_state_size=max(n_clones,4), compression_mode=mode)
    s = capi.Stream(ctx, calib, default_fe_cfg(), cfg)
    pr = ekf_problems.make_problem(calib, seed=seed, n_clones=n_clones, n_feat=n_feat, **kw)
    ref = O.ekf_update_problem(calib, cfg, pr["gravity"], pr["clones"], pr["P"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], -1)
    s.ekf_set_cov(pr["P"])
    got = s.ekf_update(pr["gravity"], pr["clones"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], -1, True)
    Pg = s.ekf_get_cov()
    eP = np.abs(Pg - ref["P"]).max() / np.abs(ref["P"]).max()
    st = (got["status"]>>1)&1
    nobs = np.diff(pr["obs_start"])
    clones_used = sorted(set(int(c) for j in range(n_feat) if st[j] for c in pr["obs_clone"][pr["obs_start"][j]:pr["obs_start"][j+1]]))
    s.close()
    return got["rows"], 6*len(clones_used), got["used_qr"], eP, list(nobs), list(st)
for nc in ():
    for nf in (2, 4, 8):
        for seed in (1, 2, 3):
            r = run(nc, nf, 100*nc+seed, 2, min_obs=3)
            print(nc, nf, seed, "rows %d na %d qr %d errP %.2e nobs %s passed %s" % r, flush=True)
print("---- seed sweep")
nfail = 0
for nc, nf, kw in ((13,4,dict(min_obs=3)),(13,3,dict(min_obs=3)),(12,4,dict(min_obs=3)),(14,5,dict(min_obs=3)),(30,6,dict(min_obs=3)),(50,40,dict(min_obs=20)),(60,30,dict(min_obs=20)),(64,30,dict(min_obs=20)),(30,60,{})):
    for seed in range(100, 140 if nc < 30 else 104):
        r = run(nc, nf, seed, 2, **kw)
        if r[3] > 1e-9 or r[3] != r[3]:
            nfail += 1
            print("FAIL", nc, nf, seed, "rows %d na %d qr %d errP %.2e" % (r[0], r[1], r[2], r[3]), flush=True)
print("done, failures:", nfail)
