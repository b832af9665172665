import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle binding (tests may use it as the checker; the product never does)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def built_lib():
    """libphyx_amd.so, built in-tree with hipcc if the snapshot does not already carry it."""
    from phyx_amd import build, _lib
    build.build()
    return _lib.load()
