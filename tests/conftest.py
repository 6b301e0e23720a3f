import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    lib = os.path.join(ROOT, "merian-quake_amd", "lib", "libmqhip.so")
    orc = os.path.join(ROOT, "oracle", "libmqoracle.so")
    if not os.path.exists(lib) or not os.path.exists(orc):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return lib, orc


@pytest.fixture(scope="session")
def built():
    return _ensure_built()


@pytest.fixture(scope="session")
def mqlib(built):
    import mqhip
    return mqhip.load_library()


@pytest.fixture(scope="session")
def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
