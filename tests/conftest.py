import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: reads /root/reference (skipped where it is absent)")


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "rene-shader"))


@pytest.fixture(scope="session")
def hip_lib():
    """librene_hip.so, built in-tree if missing (hipcc cross-compiles gfx950 without a GPU)."""
    from rene_amd import api
    if not os.path.exists(api.LIB_PATH):
        api.build()
    return api.lib()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.lib()
    return oracle


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
