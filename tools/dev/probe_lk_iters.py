"""LK lock-step cost: per-point, per-level iteration counts of the temporal and the stereo track (debug build of the
library with -DLK_ITER_DBG, which returns them in the und1 words), against what a wave of four points pays (the
maximum of its four points at every level), in feature order and for better groupings."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from msckf_stereo_c_amd import capi
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
from oracle import oracle_py as O
O.build()
ctx = capi.Context(0)
w, h = 752, 480
syn = O.Synth(seed=0x5EED0000, width=w, height=h, n_static=25, n_loop=100)
calib = O.euroc_calib(w, h)
s = capi.Stream(ctx, calib, default_fe_cfg(grid_row=8, grid_col=10, grid_min=5, grid_max=6), default_ekf_cfg())
prev_cnt = None
for k in range(40, 46):
    a, b = syn.render(k)
    s.push_stereo(a, b)
    if k == 40:
        cand = s.cell_candidates(10 * 256)
        pts = np.stack([cand["x"], cand["y"]], 1).astype(np.float32)[::3][:440]
        got = s.track(pts, do_temporal=False)
        pts = pts[(got["status"] & 2) != 0]
    else:
        got = s.track(pts, do_temporal=True)
        raw = got["und1"].view(np.uint32).reshape(-1, 2)
        for name, col in (("temporal", 0), ("stereo", 1)):
            lv = np.stack([(raw[:, col] >> (8 * l)) & 255 for l in range(4)], 1).astype(np.int64)   # [pt, level]
            n = len(lv) // 4 * 4
            lv = lv[:n]
            per_point = lv.sum(1).mean()
            def cost(order):
                g = lv[order].reshape(-1, 4, 4)
                return g.max(1).sum(1).mean()
            ident = cost(np.arange(n))
            ideal = cost(np.argsort(lv.sum(1), kind="stable"))
            line = f"frame {k} {name}: points {n}, iterations per point {per_point:.2f} (levels {lv.mean(0).round(2)}), per wave slot {ident:.2f}, sorted by own total {ideal:.2f}"
            if prev_cnt is not None and name in prev_cnt and len(prev_cnt[name]) == len(lv):
                line += f", sorted by last frame's total {cost(np.argsort(prev_cnt[name], kind='stable')):.2f}"
            print(line, flush=True)
            prev_cnt = prev_cnt or {}
            prev_cnt[name] = lv.sum(1)
        keep = (got["status"] & 3) == 3
        # keep the feature set fixed in size for the "last frame" comparison: only report when nothing was lost
        if not keep.all():
            pts = got["out0"][keep]; prev_cnt = None
            s.swap(); continue
        pts = got["out0"]
    s.swap()
