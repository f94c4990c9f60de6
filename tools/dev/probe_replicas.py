"""Dev aid: determinism probe — many replicated streams over several pipelined groups; reports streams whose filter
state differs bit-wise from the first replica of their sequence."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle_py as O
from msckf_stereo_c_amd import runner as R
from msckf_stereo_c_amd.runner import IMU_SAMPLE
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg

n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per_group = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 110
pipelined = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
w, h = 752, 480
fe = default_fe_cfg(grid_row=8, grid_col=10, grid_min=4, grid_max=5)
ekf = default_ekf_cfg(max_cam_state_size=30)
uniq = [O.Synth(seed=0x5EED0050 + i, width=w, height=h) for i in range(2)]
run = R.Runner(uniq[0].calib, fe, ekf, n_groups, per_group, host_threads=1)
packs = []
for syn in uniq:
    n_keys = syn.n_static + syn.n_loop
    frames = np.empty((2, n_keys, syn.h, syn.w), np.uint8)
    for k in range(min(n_keys, n_frames + 1)):
        a, b = syn.render(k); frames[0, k], frames[1, k] = a, b
    imu = np.zeros((n_frames + 3) * 10 + 20, IMU_SAMPLE)
    for j in range(len(imu)):
        s = syn.imu(j); imu[j] = (s.time_stamp, tuple(s.angular_velocity), tuple(s.linear_acceleration))
    packs.append((frames, imu, n_keys, syn))
N = n_groups * per_group
for s in range(N):
    frames, imu, n_keys, syn = packs[s % 2]
    fb = syn.w * syn.h
    run.set_sequence(s, frames.ctypes.data, frames.ctypes.data + n_keys * fb, 0, fb, syn.n_static, syn.n_loop, 1403715273262142976, 50000000, imu)
step = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if step:
    seen = set()
    for f0 in range(0, n_frames, step):
        run.run(f0, step, threaded=True, pipelined=pipelined)
        for s in range(2, N):
            if s in seen: continue
            a, b = run.cov(s % 2), run.cov(s)
            if a.shape != b.shape or not np.array_equal(a, b):
                seen.add(s)
                d = np.abs(a - b)
                rows = np.nonzero(d.max(axis=1) > 0)[0]
                blk = sorted(set(((rows[rows >= 21] - 21) // 6).tolist()))
                print("frame", f0 + step - 1, "stream", s, "dim", a.shape[0], "max", float(d.max()), "rel", float(d.max() / np.abs(a).max()),
                      "nonzero", int((d > 0).sum()), "of", d.size, "imu rows differ", int((rows < 21).sum()), "clone blocks", blk[:40],
                      "updates", run.num_updates(s), run.num_updates(s % 2))
else:
    run.run(0, n_frames, threaded=True, pipelined=pipelined)
bad = []
for s in range(2, N):
    ref = s % 2
    a, b = run.cov(ref), run.cov(s)
    if a.shape != b.shape or not np.array_equal(a, b):
        bad.append((s, s // per_group, float(np.abs(a - b).max()) if a.shape == b.shape else -1.0))
print("DIVERGED %d of %d:" % (len(bad), N - 2), bad[:12])
for s, g, _ in bad[:6]:
    a, b = run.cov(s % 2), run.cov(s)
    if a.shape != b.shape: print("  stream", s, "dims", a.shape, b.shape); continue
    d = np.abs(a - b)
    rows = np.nonzero(d.max(axis=1) > 0)[0]
    blk = sorted(set(((rows[rows >= 21] - 21) // 6).tolist()))
    i, j = np.unravel_index(np.argmax(d), d.shape)
    pa, pb = run.poses(s % 2), run.poses(s)
    k0 = int(np.argmax(np.any(pa["p"] != pb["p"], axis=1))) if len(pa) == len(pb) else -1
    print("  stream", s, "max at", (int(i), int(j)), "nonzero", int((d > 0).sum()), "of", d.size, "imu rows", int((rows < 21).sum()),
          "clone blocks", len(blk), "first differing pose index", k0, "of", len(pa), "updates", run.num_updates(s), run.num_updates(s % 2),
          "dpos max", float(np.abs(pa["p"] - pb["p"]).max()) if len(pa) == len(pb) else None)
