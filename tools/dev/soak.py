"""Dev aid: long lockstep run of one stream (GPU path vs CPU oracle) — ids/pixels every frame, pose drift, covariance health."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle_py as O
from msckf_stereo_c_amd import runner as R
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 800
compat = int(sys.argv[2]) if len(sys.argv) > 2 else 15
syn = O.Synth(seed=0x5EED0077, width=376, height=240)
fe, ekf = default_fe_cfg(compat=compat), default_ekf_cfg()
osys = O.OracleSystem(syn.calib, fe, ekf)
run = R.Runner(syn.calib, fe, ekf, 1, 1)
view = R.StreamView(run)
j = 0
t0 = time.time()
for k in range(n_frames):
    t_img = syn.frame_time(k)
    while True:
        s = syn.imu(j); j += 1
        osys.imu(s); view.imu(s)
        if not (s.time_stamp <= t_img):
            break
    a, b = syn.render(k)
    osys.stereo(a, b, t_img); view.stereo(a, b, t_img)
    osys.backend(); view.backend()
    o, g = osys.dump(), run.dump(0)
    if not (np.array_equal(o[0], g[0]) and np.array_equal(o[2], g[2]) and np.array_equal(o[3], g[3])):
        print("frame", k, "ids/pixels differ"); break
    if k % 100 == 99:
        op, gp = osys.poses(), run.poses(0)
        print("frame %d: %d features, %d updates, max |dp| %.3e m (%.0f s)" % (k, len(g[0]), run.num_updates(0), np.abs(op["p"] - gp["p"]).max(), time.time() - t0), flush=True)
P = run.cov(0)
w = np.linalg.eigvalsh((P + P.T) / 2)
print("cov: dim %d, asymmetry %.3e, min eig %.3e, max eig %.3e" % (P.shape[0], np.abs(P - P.T).max(), w.min(), w.max()))
Po = osys.cov()
print("cov vs oracle: %.3e relative" % (np.abs(P - Po).max() / np.abs(Po).max()))
print("resets", run.num_resets(0), osys.num_resets())
