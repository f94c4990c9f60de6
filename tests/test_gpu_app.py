"""The headless run_euroc_single_thread binary (drop-in harness: cg::System from YAML files, EuRoC mav0 directory
layout, PNG images, CSV timestamps parsed like the reference app) against the CPU oracle on the same files."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_euroc_single_thread_on_synthetic_mav0(tmp_path, oracle):
    from PIL import Image
    from msckf_stereo_c_amd import build
    build.build_all()
    n_frames = 40
    # (a short static start and a fast trajectory: the tenth published frame then has lost features to linearise, which the
    # debug_msckfvio.txt check below REQUIRES — found with the oracle: frame 29 of this stream has a lost-feature update)
    syn = oracle.Synth(seed=0x5EED0042, width=752, height=480, n_static=21, motion_scale=3.0)
    mav0 = tmp_path / "mav0"
    for c in (0, 1):
        (mav0 / ("cam%d" % c) / "data").mkdir(parents=True)
    (mav0 / "imu0").mkdir()
    t0_ns, dt_ns = 1403715273262142976, 50000000
    rows = []
    for k in range(n_frames):
        a, b = syn.render(k)
        name = "%d.png" % (t0_ns + k * dt_ns)
        Image.fromarray(a).save(mav0 / "cam0" / "data" / name)
        Image.fromarray(b).save(mav0 / "cam1" / "data" / name)
        rows.append("%d,%s\r" % (t0_ns + k * dt_ns, name))
    for c in (0, 1):
        (mav0 / ("cam%d" % c) / "data.csv").write_text("#timestamp [ns],filename\r\n" + "\n".join(rows) + "\n")
    n_imu = (n_frames + 2) * 10
    lines = ["#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z"]
    for j in range(n_imu):
        s = syn.imu(j)
        vals = list(s.angular_velocity) + list(s.linear_acceleration)
        lines.append("%d,%s" % (t0_ns + j * (dt_ns // 10), ",".join("%.9g" % v for v in vals)))
    (mav0 / "imu0" / "data.csv").write_text("\n".join(lines) + "\n")
    shutil.copytree(os.path.join(ROOT, "config"), tmp_path / "config")
    work = tmp_path / "build"
    work.mkdir()
    exe = os.path.join(ROOT, "msckf_stereo_c_amd", "_build", "run_euroc_single_thread")
    res = subprocess.run([exe, str(mav0)], cwd=work, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    got = np.loadtxt(work / "pose_out.txt").reshape(-1, 8)
    # oracle on the same stream, reference harness order
    osys = oracle.OracleSystem(syn.calib, default_fe_cfg(), default_ekf_cfg())
    syn.feed(osys, n_frames)
    ref = osys.poses()
    assert len(got) == len(ref) > 10
    assert np.abs(got[:, 0] - ref["t"]).max() < 2e-6
    assert np.abs(got[:, 1:4] - ref["p"]).max() < 1e-4 + 1e-6          # 6-decimal text output (Q15)
    assert np.abs(got[:, 4:8] - ref["q"]).max() < 1e-4 + 1e-6
    dbg = (work / "debug_imageprocessor.txt").read_text().strip().splitlines()
    assert len(dbg) == n_frames
    # debug_msckfvio.txt (msckf_vio.cpp:169-171, 719-723): un-projected Jacobians of the features linearised in frame n_pub == 9
    # (the file is created at start-up; frame 9 writes one block triple per linearised feature)
    assert (work / "debug_msckfvio.txt").exists()
    jac = (work / "debug_msckfvio.txt").read_text().split("featureJacobian ")
    assert len(jac) > 3, "frame n_pub == 9 of this stream linearises features: the dump must not be empty"
    assert jac[1].startswith("H_xj:") and jac[2].startswith("H_fj:") and jac[3].startswith("r_j:")
    hx = np.array([[float(v) for v in line.split()] for line in jac[1].splitlines()[1:] if line.strip()])
    hf = np.array([[float(v) for v in line.split()] for line in jac[2].splitlines()[1:] if line.strip()])
    assert hx.shape[0] == hf.shape[0] and hx.shape[0] % 4 == 0 and hf.shape[1] == 3 and (hx.shape[1] - 21) % 6 == 0
    assert np.all(hx[:, :21] == 0) and np.abs(hx).max() > 0          # clone columns only (:713)
