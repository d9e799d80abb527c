import json
import os
import sys

import pytest
import torch  # noqa: F401  (before the HIP library: torch bundles its own HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi.Oracle()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "survey_8c.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cm():
    """The product package (ctypes bindings over libcoolmic-dsp-hip.so)."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu(cm):
    if cm.device_count() < 1:
        pytest.fail("gpu-marked test on a machine without a HIP device")
    return cm
