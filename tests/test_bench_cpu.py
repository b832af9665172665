"""bench.py pieces that run without a GPU: the cpu_baseline leg (the oracle's impulse loop timed on the host) and the
contract that the bench itself refuses to run without a device instead of falling back."""
import importlib.util
import os
import subprocess
import sys

from helpers import presolve_state
from phyx_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cpu_baseline_leg(oracle):
    bodies, cps, joints = presolve_state(scenes.stack(8, 60), 3, iters=20)
    out = _bench_module().cpu_baseline(bodies, cps, joints, 20, 3.0)
    assert out["unit"] == "joint-visits/s" and out["kind"] == "port"
    assert out["value"] > 1e6 and out["single_thread"]["value"] > 1e6 and out["scalar_port_single_thread_value"] > 1e6
    assert 1 <= out["cores"] <= (os.cpu_count() or 1)
    assert "impulse sweeps" in out["sample"]
    for phases in (out["phases_ms"], out["single_thread"]["phases_ms"]):           # the reference's scopes, BASELINE.md §3
        assert set(phases) >= {"refresh", "prestep", "impulse", "displacement", "prepare_indices"} and phases["impulse"] > 0
    assert out["broadphase_ms"]["threads_1"]["UpdatePairs"] > 0 and out["broadphase_ms"]["candidate_tests"] > 0


def test_bench_fails_loudly_without_gpu(built_lib):
    if built_lib.phx_device_count() > 0:
        import pytest
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "no usable HIP device" in (p.stderr + p.stdout)
    assert not p.stdout.strip().startswith("{")          # no JSON line, i.e. no silent CPU number
