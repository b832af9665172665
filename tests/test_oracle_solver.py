"""Behaviour of the CPU oracle's solver + world (no GPU): internal consistency, the reference's documented
mode semantics, ordering invariants, and regression fixtures.  These functions restate
the reference's .cpp files and are NOT pinned against it (see oracle/phx_oracle.h) — the checks here are the
ones that can be made without the reference: determinism, equivalences the algorithm implies, invariants."""
import os

import numpy as np
import pytest

from helpers import SMALL_SCENES, presolve_state, oracle_world, is_static
from phyx_amd import scenes

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", params=list(SMALL_SCENES))
def state(request):
    make, warm = SMALL_SCENES[request.param]
    return request.param, presolve_state(make(), warm)


def _solve(oracle, state, solve_mode, island_mode, ci=15, pi=15):
    b, cp, j = (a.copy() for a in state)
    order, st = oracle.solver_solve(b, cp, j, solve_mode, island_mode, ci, pi)
    return b, j, order, st


def test_deterministic(oracle, state):
    _, s = state
    a = _solve(oracle, s, oracle.SOLVE_AVX2, oracle.ISLAND_MULTIPLE)
    b = _solve(oracle, s, oracle.SOLVE_AVX2, oracle.ISLAND_MULTIPLE)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_scalar_single_equals_identity_order(oracle, state):
    """Single + scalar = joints swept in array order (ref: Solver.cpp:102-103, PrepareIndices no-op :221-222)."""
    _, s = state
    b1, j1, order, _ = _solve(oracle, s, oracle.SOLVE_SCALAR, oracle.ISLAND_SINGLE)
    nj = len(s[2])
    assert np.array_equal(order[:nj], np.arange(nj))
    b2, cp, j2 = (a.copy() for a in s)
    oracle.solver_solve_ordered(b2, cp, j2, np.arange(nj), None, 15, 15)
    assert b1.tobytes() == b2.tobytes() and j1.tobytes() == j2.tobytes()


def test_sloppy_equals_strict_with_zero_workers(oracle, state):
    """With workers = 0 the 512-joint batches run back to back in order (ref: Solver.cpp:138-139)."""
    _, s = state
    for mode in (oracle.SOLVE_SCALAR, oracle.SOLVE_AVX2):
        a = _solve(oracle, s, mode, oracle.ISLAND_SINGLE)
        b = _solve(oracle, s, mode, oracle.ISLAND_SINGLE_SLOPPY)
        assert a[0].tobytes() == b[0].tobytes()


@pytest.mark.parametrize("mode,n", [(1, 4), (2, 8)])
def test_prepare_indices_groups_are_body_disjoint(oracle, state, mode, n):
    """ref: Solver.cpp:217-273 — every full group of N consecutive slots before groupOffset shares no body."""
    _, s = state
    joints = s[2]
    _, _, order, st = _solve(oracle, s, mode, oracle.ISLAND_SINGLE)
    nj = len(joints)
    order = order[:nj]
    assert sorted(order.tolist()) == list(range(nj))
    assert st.group_offset % n == 0 and 0 <= st.group_offset <= nj
    for g in range(0, st.group_offset, n):
        seen = set()
        for k in order[g:g + n]:
            for body in (int(joints["body1"][k]), int(joints["body2"][k])):
                assert body not in seen
                seen.add(body)


def test_gather_islands_partition(oracle, state):
    """ref: Solver.cpp:285-454 — islands are body-disjoint (static bodies aside), coalesced to >= 256 joints."""
    _, s = state
    bodies, _, joints = s
    L = oracle.lib()
    nb, nj = len(bodies), len(joints)
    cap = nj + (nj // 256 + 2) * 8 + 16
    ji = np.full(cap, -1, dtype=np.int32)
    off = np.zeros(nb + 1, dtype=np.int32)
    siz = np.zeros(nb + 1, dtype=np.int32)
    import ctypes as C
    cnt, mx = C.c_int32(), C.c_int32()
    total = L.phxo_gather_islands(bodies.ctypes.data, nb, joints.ctypes.data, nj, 8, ji.ctypes.data, cap,
                                  off.ctypes.data, siz.ctypes.data, C.byref(cnt), C.byref(mx))
    assert total >= 0 and cnt.value >= 1
    static = is_static(bodies)
    owner = {}
    placed = 0
    for i in range(cnt.value):
        assert off[i] % 8 == 0
        if i < cnt.value - 1:
            assert siz[i] >= 256
        for k in ji[off[i]:off[i] + siz[i]]:
            assert k >= 0
            placed += 1
            for body in (int(joints["body1"][k]), int(joints["body2"][k])):
                if not static[body]:
                    assert owner.setdefault(body, i) == i
    both_static = int(np.sum(static[joints["body1"]] & static[joints["body2"]]))
    assert placed == nj - both_static
    assert mx.value == max(siz[:cnt.value])


def test_islands_do_not_change_scalar_results(oracle, state):
    """Body-disjoint islands are independent Gauss-Seidel problems, so splitting them (Multiple) leaves every
    joint's relative order inside its island unchanged and the scalar result identical — unless a static
    body's tag couples two islands: solveBodiesImpulse[ground].lastIteration is reset once per SolveJoints
    (ref: Solver.cpp:474), so in Multiple mode a later island sees the tag the previous island left behind
    (the reference's own TODO at Solver.cpp:244 is about this sharing).  Piles resting on the ground hit
    that coupling; free-standing stacks do not."""
    name, s = state
    if not name.startswith("stack"):
        pytest.skip("ground-tag coupling between islands changes skip decisions in this scene")
    a = _solve(oracle, s, oracle.SOLVE_SCALAR, oracle.ISLAND_SINGLE)
    b = _solve(oracle, s, oracle.SOLVE_SCALAR, oracle.ISLAND_MULTIPLE)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_group_modes_differ_from_scalar_but_stay_close(oracle):
    """N-grouping changes the Gauss-Seidel order, so SSE2/AVX2 != scalar (SURVEY.md §0 finding 2) — but a
    resting stack must still come to rest under every mode."""
    s = presolve_state(scenes.stack(10, 100), 3)
    a = _solve(oracle, s, oracle.SOLVE_SCALAR, oracle.ISLAND_SINGLE, 20, 20)
    b = _solve(oracle, s, oracle.SOLVE_AVX2, oracle.ISLAND_SINGLE, 20, 20)
    assert a[0].tobytes() != b[0].tobytes()
    for res in (a, b):
        v = res[0]["velocity"]
        assert np.isfinite(v["x"]).all() and np.isfinite(v["y"]).all()
        assert np.abs(v["y"]).max() < 25.0


def test_colour_sync_equals_sequential_static_tags(oracle, state, built_lib):
    """The device sees a static body's lastIteration tag only across colour boundaries;
    the oracle can run both rules in the device's colour order and reports how often they disagree."""
    import phyx_amd
    _, s = state
    bodies, cps, joints = s
    order, offs = phyx_amd.schedule_colours(joints["body1"], joints["body2"], is_static(bodies))
    b1, j1 = bodies.copy(), joints.copy()
    st_seq = oracle.solver_solve_ordered(b1, cps, j1, order, offs, 15, 15, oracle.STAG_SEQUENTIAL)
    b2, j2 = bodies.copy(), joints.copy()
    oracle.solver_solve_ordered(b2, cps, j2, order, offs, 15, 15, oracle.STAG_COLOUR_SYNC)
    assert st_seq.stag_events == 0
    assert b1.tobytes() == b2.tobytes() and j1.tobytes() == j2.tobytes()


def test_early_exit_is_equivalent_to_running_on(oracle, state):
    """A sweep after an unproductive sweep skips every joint, so stopping early (ref: Solver.cpp:189) and
    running all configured sweeps give the same result — the device relies on this."""
    _, s = state
    b1, cp, j1 = (a.copy() for a in s)
    st = oracle.solver_solve_ordered(b1, cp, j1, np.arange(len(j1)), None, 200, 200)
    assert st.displacement_iterations < 200     # the penetration loop always settles in these scenes
    b2, _, j2 = (a.copy() for a in s)
    oracle.solver_solve_ordered(b2, cp, j2, np.arange(len(j2)), None, st.impulse_iterations, st.displacement_iterations)
    assert b1.tobytes() == b2.tobytes() and j1.tobytes() == j2.tobytes()


def test_refresh_joint_formulas(oracle, state):
    """Spot-check RefreshJoints against the formulas of SURVEY.md Appendix A.1 evaluated in float64."""
    _, s = state
    bodies, cps, joints = s
    for k in range(0, len(joints), max(1, len(joints) // 40)):
        j = joints[k]
        out = oracle.refresh_joint(bodies, cps, j)
        b1, b2, cp = bodies[j["body1"]], bodies[j["body2"]], cps[j["contact_point_index"]]
        n = np.array([cp["normal"]["x"], cp["normal"]["y"]], dtype=np.float64)
        d1 = np.array([cp["delta1"]["x"], cp["delta1"]["y"]], dtype=np.float64)
        d2 = np.array([cp["delta2"]["x"], cp["delta2"]["y"]], dtype=np.float64)
        p1 = d1 + [b1["pos"]["x"], b1["pos"]["y"]]
        p2 = d2 + [b2["pos"]["x"], b2["pos"]["y"]]
        w1, w2 = d1, p1 - [b2["pos"]["x"], b2["pos"]["y"]]
        a1 = n[0] * w1[1] - n[1] * w1[0]
        a2 = -n[0] * w2[1] + n[1] * w2[0]
        m = (n @ n) * b1["inv_mass"] + a1 * a1 * b1["inv_inertia"] + (n @ n) * b2["inv_mass"] + a2 * a2 * b2["inv_inertia"]
        assert out[0] == cp["normal"]["x"] and out[2] == -cp["normal"]["x"]
        np.testing.assert_allclose(out[4], a1, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out[5], a2, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out[12], 0.0 if m == 0 else 1.0 / m, rtol=1e-4)
        depth = (p2 - p1) @ n
        assert out[14] == (np.float32(-0.1) if np.float32(depth) < 1 else 0.0)
        np.testing.assert_allclose(out[15], 0.1 * max(0.0, depth - 2.0), rtol=1e-4, atol=1e-5)
        assert out[17] == -cp["normal"]["y"] and out[18] == cp["normal"]["x"]          # tangent = (-n.y, n.x)


def test_world_stack_rests_and_is_deterministic(oracle):
    runs = []
    for _ in range(2):
        w = oracle_world(scenes.stack(4, 30))
        for _ in range(40):
            w.update(contact_iters=20, penetration_iters=20)
        runs.append(w.bodies().copy())
    assert runs[0].tobytes() == runs[1].tobytes()
    b = runs[0]
    assert np.isfinite(b["pos"]["x"]).all() and np.isfinite(b["pos"]["y"]).all()
    y0 = 15.0 + 10.0 * np.tile(np.arange(30), 4)
    # 20 sweeps do not converge a 30-high column (SURVEY.md §0 finding 2): it compresses and sways by a few
    # units but stays a stack
    assert np.abs(b["pos"]["y"][1:] - y0).max() < 10.0
    assert b["pos"]["y"][0] == 0.0 and b["velocity"]["y"][0] == 0.0  # the ground never moves


def test_world_contact_cache_bijection(oracle):
    """ref: World.cpp:72-149 — after RefreshContactJoints joints <-> live contact points is a bijection."""
    w = oracle_world(scenes.falling(300, width=60.0, ymax=250.0))
    for step in range(60):
        w.pre_solve()
        m, cps, joints = w.manifolds(), w.contact_points(), w.joints()
        live = [(int(mm["point_index"]) + k) for mm in m for k in range(int(mm["point_count"]))]
        assert len(live) == len(joints)
        assert sorted(int(x) for x in joints["contact_point_index"]) == sorted(live)
        for k, j in enumerate(joints):
            assert cps["solver_index"][j["contact_point_index"]] == k
        pairs = set(zip(m["body1"].tolist(), m["body2"].tolist()))
        assert len(pairs) == len(m)                                   # the pair set keeps manifolds unique
        w.solve_and_integrate()
    assert w.L.phxo_world_point_overflows(w.h) == 0


def test_regression_fixture(oracle):
    """Oracle-generated regression vectors (tests/golden/oracle_solver_stack2x10.npz, written by
    tests/golden/make_oracle_fixture.py): guards the oracle itself against accidental drift.  These are
    NOT reference outputs."""
    g = np.load(os.path.join(GOLD_DIR, "oracle_solver_stack2x10.npz"))
    for name, mode in (("scalar", 0), ("sse2", 1), ("avx2", 2)):
        b = g["bodies"].copy().view(oracle.body_dtype)
        j = g["joints"].copy().view(oracle.joint_dtype)
        cp = g["cps"].copy().view(oracle.contact_point_dtype)
        oracle.solver_solve(b, cp, j, mode, oracle.ISLAND_SINGLE, 15, 15)
        assert b.tobytes() == g["bodies_out_" + name].tobytes()
        assert j.tobytes() == g["joints_out_" + name].tobytes()


def test_binary16_rounding_model(oracle):
    """The fp16 ablation's rounding (float -> binary16 nearest-even -> float) against numpy's float16."""
    import ctypes as C
    L = oracle.lib()
    L.phxo_round_f16.restype = C.c_float
    L.phxo_round_f16.argtypes = [C.c_float]
    rng = np.random.default_rng(11)
    vals = np.concatenate([
        (rng.standard_normal(20000) * 10.0 ** rng.integers(-9, 6, 20000)).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 6.1035156e-05, 6.0975552e-05, 5.9604645e-08, 2.9802322e-08,
                  2.9802326e-08, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, np.inf, -np.inf], dtype=np.float32)])
    got = np.array([L.phxo_round_f16(float(v)) for v in vals], dtype=np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).astype(np.float32)
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("name", ["stack2x10", "stack10x100", "tilted60", "falling600"])
def test_cpu_baseline_strict_build_equals_the_oracle_avx2_order(oracle, name):
    """oracle/cpu_baseline.c (the 8-wide AVX2-order CPU baseline bench.py times with the reference's flags) is the same
    algorithm as the oracle's AVX2 mode: built with the oracle's strict flags it reproduces phxo_solver_solve(AVX2, Single)
    bit for bit — greedy 8-grouping, group-granular skip, SIMD flipsign, early exits.  The fast build (-ffast-math -mfma) stays
    within SURVEY.md §8(c)'s T1 tolerance of it after one solve."""
    from helpers import SMALL_SCENES, presolve_state
    make, warm = SMALL_SCENES[name]
    bodies, cps, joints = presolve_state(make(), warm)
    for iters in ((20, 20), (15, 15), (7, 0), (50, 50)):
        bo, jo = bodies.copy(), joints.copy()
        _, st = oracle.solver_solve(bo, cps, jo, oracle.SOLVE_AVX2, oracle.ISLAND_SINGLE, *iters)
        bs, js = bodies.copy(), joints.copy()
        ph = oracle.baseline_solve(bs, cps, js, iters[0], iters[1], 1, False, "strict")
        assert bs.tobytes() == bo.tobytes() and js.tobytes() == jo.tobytes()
        assert (ph.impulse_iterations, ph.displacement_iterations, ph.group_offset) == (st.impulse_iterations, st.displacement_iterations, st.group_offset)
        b2, j2 = bodies.copy(), joints.copy()
        oracle.baseline_solve(b2, cps, j2, iters[0], iters[1], 1, True, "strict")          # Sloppy batches, one thread: the same sequence
        assert b2.tobytes() == bs.tobytes() and j2.tobytes() == js.tobytes()
        bf, jf = bodies.copy(), joints.copy()
        oracle.baseline_solve(bf, cps, jf, iters[0], iters[1], 1, False, "fast")
        if iters[0] <= 20:
            for f in ("x", "y"):
                assert np.abs(bf["velocity"][f] - bo["velocity"][f]).max() <= 1e-3           # T1: |dvel| <= 1e-3
        bt, jt = bodies.copy(), joints.copy()
        oracle.baseline_solve(bt, cps, jt, iters[0], iters[1], 3, True, "fast")              # racy like the reference: finite, clamped
        assert np.isfinite(bt["velocity"]["x"]).all() and (jt["normal_acc"] >= 0).all()


def test_cpu_baseline_broadphase_counts_match_the_oracle(oracle):
    from helpers import presolve_state
    from phyx_amd import scenes
    bodies, _, _ = presolve_state(scenes.stack(12, 30), 3)
    _, _, ent = oracle.broadphase_build(bodies)
    cand, cnt, tests = oracle.sweep_candidates(ent)
    for threads in (1, 3):
        ph = oracle.baseline_broadphase(bodies, threads, 2, "fast")
        assert ph.overlapping_pairs == len(cand) and ph.candidate_tests == tests
