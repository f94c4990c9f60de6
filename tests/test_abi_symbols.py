"""The C-ABI shared library loads and exports every symbol include/mskf_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mskf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mskf_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from msckf_stereo_c_amd import build
    lib = build.build_hip()
    L = ctypes.CDLL(lib)
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert L.mskf_abi_version() == 4


def test_no_cpu_fallback_without_device():
    """Without a GPU the product path fails loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from msckf_stereo_c_amd import capi
    with pytest.raises(capi.MskfError):
        capi.Context(0)


def test_product_does_not_link_oracle():
    """Nothing under msckf_stereo_c_amd/ may include, import or link anything under oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "msckf_stereo_c_amd")):
        if "_build" in root or "__pycache__" in root:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r'#include\s+"[^"]*oracle/|from oracle|import oracle|liboracle', txt):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_feature_measurement_layout():
    from msckf_stereo_c_amd.ctypes_types import FEATURE_MEAS, POSE
    assert FEATURE_MEAS.itemsize == 40      # cg::FeatureMeasurement, data_msg.h:30-36
    assert POSE.itemsize == 64
