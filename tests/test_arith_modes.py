"""The two stated arithmetic forms of the sweeps (include/phyx_amd.h phx_arith_mode; oracle/phx_oracle.c mul_add / mul_sub).

The reference writes `dV -= projector * velocity` / `velocity += compMass * dImpulse` and ships -ffast-math -mfma
(ref: Makefile:11, 17-24), so whether such a pair is fused is its compiler's choice.  The library states its choice (fused
by default) and is bit-exact against the oracle's matching form (the `gpu` tests, tests/conftest.py).  Here, on the CPU: the two
oracle forms are different roundings of the same computation and stay within SURVEY.md section 8(c)'s tiers of each other —
T0 (one impulse sweep, same inputs and order: 1e-5 abs / rel) and T1 (one SolveJoints: |dvel| <= 1e-3, |dpos| <= 1e-4 after
integration) — and the fused form really is a different rounding (a test that could not tell the forms apart would pin nothing).
"""
import numpy as np
import pytest

from helpers import SMALL_SCENES, presolve_state, oracle_world
from phyx_amd import scenes

VEL = ("velocity", "displacing_velocity")


def _solve_in(oracle, form, state, solve_mode, island_mode, ci, pi):
    b, cp, j = (a.copy() for a in state)
    prev = oracle.set_arith(form)
    try:
        oracle.solver_solve(b, cp, j, solve_mode, island_mode, ci, pi)
    finally:
        oracle.set_arith(prev)
    return b, j


def _max_abs(a, b, fields):
    out = 0.0
    for f in fields:
        x, y = a[f], b[f]
        if x.dtype.names:
            for n in x.dtype.names:
                out = max(out, float(np.max(np.abs(x[n].astype(np.float64) - y[n].astype(np.float64)))) if len(x) else 0.0)
        else:
            out = max(out, float(np.max(np.abs(x.astype(np.float64) - y.astype(np.float64)))) if len(x) else 0.0)
    return out


@pytest.fixture(scope="module", params=list(SMALL_SCENES))
def state(request):
    make, warm = SMALL_SCENES[request.param]
    return presolve_state(make(), warm)


def test_default_form_is_source(oracle):
    assert oracle.get_arith() == oracle.ARITH_SOURCE


def test_t0_one_impulse_sweep(oracle, state):
    """One impulse sweep (PreStep + one pass over the joints), identical inputs and order: 1e-5 abs + 1e-5 rel (of the body's speed)."""
    for mode in (oracle.SOLVE_SCALAR, oracle.SOLVE_AVX2):
        bs, js = _solve_in(oracle, oracle.ARITH_SOURCE, state, mode, oracle.ISLAND_SINGLE, 1, 0)
        bf, jf = _solve_in(oracle, oracle.ARITH_FUSED, state, mode, oracle.ISLAND_SINGLE, 1, 0)
        # per body, relative to the body's speed: a component near zero is a difference of terms of that size (ulp(150) = 1.5e-5)
        comps = [(bs["velocity"]["x"], bf["velocity"]["x"]), (bs["velocity"]["y"], bf["velocity"]["y"]), (bs["angular_velocity"], bf["angular_velocity"])]
        scale = np.max(np.abs(np.stack([a.astype(np.float64) for a, _ in comps])), axis=0)
        for a, b in comps:
            assert np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= 1e-5 + 1e-5 * scale)
        for f in ("normal_acc", "friction_acc"):
            a, b = js[f].astype(np.float64), jf[f].astype(np.float64)
            assert np.all(np.abs(a - b) <= 1e-5 + 1e-5 * np.abs(a))


def test_t1_one_solve(oracle, state):
    """One full SolveJoints (15 + 15 sweeps), same order: |dvel| <= 1e-3; positions one IntegratePosition later <= 1e-4."""
    dt = 1.0 / 60.0
    for mode, island in ((oracle.SOLVE_SCALAR, oracle.ISLAND_SINGLE), (oracle.SOLVE_AVX2, oracle.ISLAND_MULTIPLE)):
        bs, _ = _solve_in(oracle, oracle.ARITH_SOURCE, state, mode, island, 15, 15)
        bf, _ = _solve_in(oracle, oracle.ARITH_FUSED, state, mode, island, 15, 15)
        assert _max_abs(bs, bf, VEL) <= 1e-3
        assert _max_abs(bs, bf, ("angular_velocity", "displacing_angular_velocity")) <= 1e-3
        for n in ("x", "y"):      # ref: World.cpp:57-70: pos += displacingVelocity + velocity * dt
            ps = bs["pos"][n].astype(np.float64) + bs["displacing_velocity"][n] + bs["velocity"][n].astype(np.float64) * dt
            pf = bf["pos"][n].astype(np.float64) + bf["displacing_velocity"][n] + bf["velocity"][n].astype(np.float64) * dt
            assert np.max(np.abs(ps - pf)) <= 1e-4


def test_forms_really_differ(oracle):
    """On a scene with thousands of joints some result rounds differently (else the form switch would be dead code)."""
    st = presolve_state(scenes.stack(10, 100), 3)
    bs, _ = _solve_in(oracle, oracle.ARITH_SOURCE, st, oracle.SOLVE_SCALAR, oracle.ISLAND_SINGLE, 15, 15)
    bf, _ = _solve_in(oracle, oracle.ARITH_FUSED, st, oracle.SOLVE_SCALAR, oracle.ISLAND_SINGLE, 15, 15)
    assert bs.tobytes() != bf.tobytes()


def test_k_steps_stay_within_the_modes_own_spread(oracle):
    """K = 10 world steps of the 1k-box scene (BASELINE config 1): the fused-vs-source distance is a rounding-noise distance — no
    larger than the reference's own scalar-vs-AVX2 distance at the same K (<= 2x that spread)."""
    scene = scenes.stack(10, 100)

    def run(form, mode):
        prev = oracle.set_arith(form)
        try:
            w = oracle_world(scene)
            for _ in range(10):
                w.update(1.0 / 60.0, mode, oracle.ISLAND_SINGLE, 20, 20)
            return w.bodies().copy()
        finally:
            oracle.set_arith(prev)

    a = run(oracle.ARITH_SOURCE, oracle.SOLVE_SCALAR)
    b = run(oracle.ARITH_FUSED, oracle.SOLVE_SCALAR)
    c = run(oracle.ARITH_SOURCE, oracle.SOLVE_AVX2)

    def mean_dpos(x, y):
        return float(np.mean(np.hypot(x["pos"]["x"].astype(np.float64) - y["pos"]["x"], x["pos"]["y"].astype(np.float64) - y["pos"]["y"])))

    spread = mean_dpos(a, c)
    assert np.isfinite(b["pos"]["x"]).all() and np.isfinite(b["pos"]["y"]).all()
    assert mean_dpos(a, b) <= 2.0 * spread + 1e-6
