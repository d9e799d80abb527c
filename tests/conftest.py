import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The engine FIRST, before any test module can import torch: whatever is loaded first decides which HIP
# runtime the engine is bound to (the system's ROCm 7.2 in /opt/rocm, as under bench.py -- or, torch first,
# the ROCm 7.0 copy inside the torch wheel, whose direct dispatch DESIGN 5 measured behaving differently).
# The GPU tests use no torch at all; tests/test_multirank_gloo.py imports it for gloo on the CPU only.
import __graft_entry__ as _ge  # noqa: E402

_ge.load_package()
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
