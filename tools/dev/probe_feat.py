"""Per-feature latency of k_ekf_feature_blocks: single-stream updates whose features all have M observations (one
round: 64 features on 256 CUs), with and without triangulation.  Run under rocprofv3 --kernel-trace and read the trace."""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from msckf_stereo_c_amd import capi
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
from oracle import oracle_py as O
import ekf_problems
O.build()
ctx = capi.Context(0)
calib = O.euroc_calib(376, 240)
for M in (3, 8, 16, 17, 29):
    for init in (0, 1):
        cfg = default_ekf_cfg(max_cam_state_size=max(M, 4))
        s = capi.Stream(ctx, calib, default_fe_cfg(), cfg)
        pr = ekf_problems.make_problem(calib, seed=M, n_clones=M, n_feat=64, min_obs=M)
        s.ekf_set_cov(pr["P"])
        for rep in range(3):
            got = s.ekf_update(pr["gravity"], pr["clones"], pr["positions"], pr["obs_start"], pr["obs_clone"], pr["obs_z"], -1, True,
                               needs_init=np.full(64, init, np.int32))
        print("M", M, "init", init, "rows", got["rows"], flush=True)
        s.close()
