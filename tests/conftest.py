import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def built_lib():
    """libctxtrans.so, cross-compiled for gfx950 if it is not there yet (hipcc needs no GPU)."""
    from imitation_from_observation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "imitation_from_observation_amd", "csrc"), "-j", "8"], check=True)
    return _lib.load()
