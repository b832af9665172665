"""examples/drop_in.c — the C ABI used from plain C (gcc, no Python, no HIP headers on the caller's side)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, built_lib):
    exe = str(tmp_path / "drop_in")
    lib_dir = os.path.join(ROOT, "phyx_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "drop_in.c"), "-L" + lib_dir, "-lphyx_amd", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_c_example_compiles_and_fails_loudly_without_a_gpu(tmp_path, built_lib):
    import phyx_amd
    exe = _build(tmp_path, built_lib)
    try:
        have_gpu = phyx_amd.device_count() > 0
    except phyx_amd.PhxError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe, "2", "5", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_runs_and_drop_in_call_matches_the_world_step(tmp_path, built_lib):
    exe = _build(tmp_path, built_lib)
    r = subprocess.run([exe, "24", "40", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "drop-in call vs world step: identical" in r.stdout
    assert "961 bodies" in r.stdout
