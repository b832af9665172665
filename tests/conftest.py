import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_present():
    try:
        from phyx_amd import build, _lib
        build.build()
        return _lib.load().phx_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests are the parity tests proper and need an MI355X: without a device they are skipped with a reason
    instead of failing in the first handle constructor (a plain `pytest` on a CPU box stays green)."""
    if not any("gpu" in item.keywords for item in items):
        return
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device: `gpu` tests run on the MI355X box (pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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


@pytest.fixture(autouse=True)
def _oracle_arith_matches_library(request):
    """`gpu` tests compare the library with the oracle bit for bit: the oracle sweeps in the arithmetic form the library was
    built with (include/phyx_amd.h phx_arith_mode; oracle/phx_oracle.c mul_add).  Everything else keeps the source form."""
    if "gpu" not in request.keywords:
        yield
        return
    from oracle import binding
    from phyx_amd import _lib
    prev = binding.set_arith(_lib.load().phx_arith_mode())
    yield
    binding.set_arith(prev)
