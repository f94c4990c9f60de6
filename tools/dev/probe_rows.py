"""How many rows do the updates of a steady-state C2 stream stack, and how often is m <= active columns?"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from msckf_stereo_c_amd import runner as R
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
from oracle import oracle_py as O
O.build()
syn = O.Synth(seed=0x5EED0000, width=752, height=480, n_static=25, n_loop=100)
fe = default_fe_cfg(grid_row=8, grid_col=10, grid_min=5, grid_max=6)
ekf = default_ekf_cfg(max_cam_state_size=30)
run = R.Runner(syn.calib, fe, ekf, 1, 1)
view = R.StreamView(run)
prev = (0, 0, 0)
for k0 in range(0, 140, 1):
    syn.feed(view, 1, start=k0)
    cur = (run.num_updates(0), run.num_tsqr_updates(0), run.stacked_rows(0))
    if k0 >= 80:
        print(k0, "updates +%d tsqr +%d rows +%d" % tuple(c - p for c, p in zip(cur, prev)))
    prev = cur
