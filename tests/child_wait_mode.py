"""Child process of tests/test_gpu_system.py::test_nap_wait_mode_in_a_child_process: the wait mode of the library is read
once per process (MSKF_WAIT), so a run under another mode needs a process of its own.  Three batches of one stream each
through the balanced runner, compared with the CPU oracle; prints OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import oracle_py as oracle
    oracle.build()
    from msckf_stereo_c_amd import runner as R
    from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
    from test_gpu_system import _attach_sequences, POS_TOL
    w, h, n_frames, delta = 376, 240, 36, 3
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syn = oracle.Synth(seed=0x5EED0064, width=w, height=h)
    keep = []
    run = R.Runner(syn.calib, fe, ekf, 3, 1, host_threads=1)
    _attach_sequences(oracle, run, [syn, syn, syn], n_frames + 2 * delta + 4, keep)
    run.set_stagger(delta)
    run.run(0, n_frames, threaded=True, pipelined=True)
    for g in range(3):
        osys = oracle.OracleSystem(syn.calib, fe, ekf)
        syn.feed(osys, n_frames + g * delta)
        for x, y in zip(osys.dump()[:4], run.dump(g)[:4]):
            assert np.array_equal(x, y), "ids / pixels differ in group %d" % g
        op, gp = osys.poses(), run.poses(g)
        assert len(op) == len(gp) and np.abs(op["p"] - gp["p"]).max() < POS_TOL
    run.close()
    print("OK wait=%s" % os.environ.get("MSKF_WAIT", "spin"))


if __name__ == "__main__":
    main()
