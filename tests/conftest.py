import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_ctx():
    """A C-ABI context on cuda:0; the HIP library must be present (no CPU fallback)."""
    from msckf_stereo_c_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
