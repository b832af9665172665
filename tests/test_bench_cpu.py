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
    assert 1 <= out["cores"] <= out["host_threads"] <= (os.cpu_count() or 1)
    assert out["all_host_threads"]["threads"] == out["host_threads"] and out["all_host_threads"]["value"] > 0      # every host thread, reported either way
    assert out["value"] >= out["all_host_threads"]["value"]
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


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_multi_rank_bench_reports_a_failure_as_json_and_does_not_hang(built_lib):
    """N > 1 must never hang its launcher: whatever goes wrong in a rank ends as ONE JSON line with an "error" field from rank 0 and
    a non-zero exit status, and the ranks that are still alive are taken down.  Here (no GPU) every rank fails at the device."""
    import json
    import pytest
    if built_lib.phx_device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1",
                        "--columns", "4", "--rows", "4"], capture_output=True, text=True, timeout=300, env=_clean_env(PHX_COMM_TIMEOUT_S="20"))
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2 and "error" in out and out["error"]


def test_a_rank_whose_peer_never_arrives_gives_up(built_lib):
    """Rank 0 of 2 started alone (its peer crashed before the rendezvous): the bounded rendezvous (PHX_COMM_TIMEOUT_S) turns the
    wait into an error line and a non-zero exit status within seconds — not torch's half hour, not a driver timeout."""
    import json
    import socket
    import time
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300,
                       env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PHX_COMM_TIMEOUT_S="5"))
    assert p.returncode != 0 and time.time() - t0 < 120
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    assert json.loads(lines[0])["error"]


def test_self_launch_takes_the_survivors_down(tmp_path):
    """dist.self_launch: a rank that dies must not leave the others waiting for it — they are terminated and the status is non-zero."""
    import time
    from phyx_amd import dist as pdist
    script = tmp_path / "ranks.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(600)\n")
    t0 = time.time()
    rc = pdist.self_launch(3, argv=[str(script)], timeout_s=60)
    assert rc == 7 and time.time() - t0 < 30
    script.write_text("import time\ntime.sleep(600)\n")           # nobody fails, nobody finishes: the launcher's own bound
    t0 = time.time()
    assert pdist.self_launch(2, argv=[str(script)], timeout_s=2) == 124 and time.time() - t0 < 30
