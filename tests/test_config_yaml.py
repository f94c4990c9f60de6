"""The YAML-subset reader of the host mirror (yaml-cpp is absent) parses the three config files into the
same values the defaults encode."""
import ctypes as C
import os

import numpy as np

from msckf_stereo_c_amd.ctypes_types import Calib, EkfCfg, FeCfg, default_ekf_cfg, default_fe_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yaml_configs_match_defaults(oracle):
    from msckf_stereo_c_amd import build
    build.build_all()
    L = C.CDLL(os.path.join(ROOT, "msckf_stereo_c_amd", "_build", "libmskf_host.so"))
    calib, fe, ekf = Calib(), FeCfg(), EkfCfg()
    rc = L.mskfh_load_configs(os.path.join(ROOT, "config").encode(), C.byref(calib), C.byref(fe), C.byref(ekf))
    assert rc == 0
    ref = oracle.euroc_calib(752, 480)
    for f in ("cam0_intrinsics", "cam0_distortion", "cam1_intrinsics", "cam1_distortion", "T_cam0_imu", "T_cam1_cam0", "T_imu_body"):
        assert np.allclose(np.array(getattr(calib, f)), np.array(getattr(ref, f)), rtol=0, atol=1e-15), f
    assert (calib.width, calib.height, calib.cam0_model, calib.cam1_model) == (752, 480, 0, 0)
    dfe, dek = default_fe_cfg(), default_ekf_cfg()
    for name, _ in FeCfg._fields_:
        if name != "_pad":
            assert getattr(fe, name) == getattr(dfe, name), name
    for name, _ in EkfCfg._fields_:
        if name in ("_pad",):
            continue
        a, b = getattr(ekf, name), getattr(dek, name)
        if name == "init_velocity":
            assert list(a) == list(b)
        else:
            assert a == b, name
