"""The C-ABI shared library loads and exports exactly what include/phyx_amd.h declares (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "phyx_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(phx_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(built_lib):
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(built_lib, n), "include/phyx_amd.h declares %s but libphyx_amd.so does not export it" % n


def test_python_binding_covers_header():
    from phyx_amd import _lib
    assert sorted(_lib.declared_symbols()) == _declared()


def test_no_oracle_in_product():
    """The product library and package never link, import or call the oracle."""
    out = subprocess.check_output(["nm", "-D", os.path.join(ROOT, "phyx_amd", "libphyx_amd.so")], text=True)
    assert "phxo_" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "phyx_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "phxo_" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_abi_version_and_error_string(built_lib):
    assert built_lib.phx_abi_version() == 2
    assert isinstance(built_lib.phx_last_error(), bytes)


def test_compute_fails_loudly_without_gpu(built_lib):
    """On a machine without a GPU every compute entry point reports PHX_ERR_NO_DEVICE — there is no fallback."""
    if built_lib.phx_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    for create in (built_lib.phx_solver_create, built_lib.phx_broadphase_create, built_lib.phx_world_create):
        assert create(C.byref(h), 0) == -2
        assert not h.value
        assert b"no CPU fallback" in built_lib.phx_last_error()
    from phyx_amd import World, PhxError
    with pytest.raises(PhxError):
        World()
