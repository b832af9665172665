"""Parity of the HIP solver against the oracle (MI355X).  The device sweeps the joints in its own colour
order; Gauss-Seidel results depend on that order, so the oracle is run in the SAME order
(phx_solver_get_schedule) — then every velocity and every accumulated impulse must match bit for bit
(fp32, strict IEEE on both sides).  Tolerances appear only where two different orders are compared."""
import numpy as np
import pytest

import phyx_amd
from phyx_amd import scenes, Configuration
from helpers import SMALL_SCENES, is_static, presolve_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(built_lib):
    return phyx_amd.Solver(0)


class Sched:
    """the device's schedule: slot order, colour boundaries, group (island) boundaries"""

    def __init__(self, solver):
        self.order, self.colours = solver.schedule()
        self.groups, self.lds_groups = solver.groups()


def _device_solve(solver, state, cfg):
    b, cp, j = (a.copy() for a in state)
    st = solver.SolveJoints(b, cp, j, cfg)
    sched = Sched(solver)
    return b, j, sched, sched.colours, st


def _oracle_in_device_order(oracle, state, sched, offs, cfg, mode):
    """Replay the device's schedule with the oracle's sequential scalar loop (groups = independent islands)."""
    b, cp, j = (a.copy() for a in state)
    st = oracle.solver_solve_grouped(b, cp, j, sched.order, sched.colours, sched.groups,
                                     cfg.contactIterationsCount, cfg.penetrationIterationsCount, mode)
    return b, j, st


@pytest.mark.parametrize("name", list(SMALL_SCENES))
@pytest.mark.parametrize("iters", [(15, 15), (20, 20), (7, 0), (0, 5), (1, 1)])
@pytest.mark.parametrize("island_mode", [0, 1])
def test_bit_exact_vs_oracle_in_device_order(solver, oracle, name, iters, island_mode):
    make, warm = SMALL_SCENES[name]
    state = presolve_state(make(), warm)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, island_mode, iters[0], iters[1])
    gb, gj, order, offs, st = _device_solve(solver, state, cfg)
    ob_, oj, ost = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes(), "body velocities differ from the oracle"
    assert gj.tobytes() == oj.tobytes(), "accumulated impulses differ from the oracle"
    assert st.impulse_iterations == ost.impulse_iterations
    assert st.displacement_iterations == ost.displacement_iterations
    # the reference's sequential static-tag rule gives the same answer on these scenes
    sb, sj, sst = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_SEQUENTIAL)
    assert sb.tobytes() == gb.tobytes() and sj.tobytes() == gj.tobytes()


def test_the_library_states_its_arithmetic_form_and_the_other_form_is_within_tolerance(solver, oracle, built_lib):
    """phx_arith_mode (include/phyx_amd.h): the library reports the arithmetic form of its sweeps; the oracle's MATCHING form replays it
    bit for bit (every test of this file), the OTHER form — the same order, the other rounding — does not, and stays within SURVEY.md
    section 8(c)'s T1 of it (|dvel| <= 1e-3 after one SolveJoints).  A parity suite that could not tell the forms apart would pin neither."""
    form = built_lib.phx_arith_mode()
    assert form in (oracle.ARITH_SOURCE, oracle.ARITH_FUSED) and oracle.get_arith() == form      # (conftest set the oracle to the library's form)
    make, warm = SMALL_SCENES["stack10x100"]
    state = presolve_state(make(), warm)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_MULTIPLE, 15, 15)
    gb, gj, sched, offs, st = _device_solve(solver, state, cfg)
    same_b, same_j, _ = _oracle_in_device_order(oracle, state, sched, offs, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == same_b.tobytes() and gj.tobytes() == same_j.tobytes()
    prev = oracle.set_arith(oracle.ARITH_SOURCE if form == oracle.ARITH_FUSED else oracle.ARITH_FUSED)
    try:
        other_b, other_j, _ = _oracle_in_device_order(oracle, state, sched, offs, cfg, oracle.STAG_COLOUR_SYNC)
    finally:
        oracle.set_arith(prev)
    assert gb.tobytes() != other_b.tobytes()
    for f in ("velocity", "displacing_velocity"):
        for n in ("x", "y"):
            assert np.max(np.abs(gb[f][n].astype(np.float64) - other_b[f][n])) <= 1e-3
    assert np.max(np.abs(gb["angular_velocity"].astype(np.float64) - other_b["angular_velocity"])) <= 1e-3


def test_every_config_mode_is_accepted_and_deterministic(solver, oracle):
    state = presolve_state(scenes.stack(10, 100), 3)
    ref = None
    for solve_mode in (phyx_amd.SOLVE_SCALAR, phyx_amd.SOLVE_SSE2, phyx_amd.SOLVE_AVX2):
        for island_mode in (phyx_amd.ISLAND_SINGLE, phyx_amd.ISLAND_MULTIPLE, phyx_amd.ISLAND_SINGLE_SLOPPY, phyx_amd.ISLAND_MULTIPLE_SLOPPY):
            cfg = Configuration(solve_mode, island_mode, 20, 20)
            gb, gj, order, offs, st = _device_solve(solver, state, cfg)
            ob_, oj, _ = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
            assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
            split = island_mode in (phyx_amd.ISLAND_MULTIPLE, phyx_amd.ISLAND_MULTIPLE_SLOPPY)
            if split:
                _, ost = oracle.solver_solve(*(a.copy() for a in state), 0, oracle.ISLAND_MULTIPLE, 20, 20)
                assert (st.island_count, st.island_max_size) == (ost.island_count, ost.island_max_size)
            else:
                assert (st.island_count, st.island_max_size) == (1, len(state[2]))
            if ref is None:
                ref = gb.tobytes()
    with pytest.raises(phyx_amd.PhxError):
        solver.SolveJoints(*(a.copy() for a in state), Configuration(7, 0, 1, 1))
    with pytest.raises(phyx_amd.PhxError):
        solver.SolveJoints(*(a.copy() for a in state), Configuration(0, 9, 1, 1))


def test_refresh_stage_bit_exact(solver, oracle):
    """RefreshJoints output (ref: Solver.cpp:592-695), 30 floats per joint, device vs oracle."""
    state = presolve_state(scenes.tilted(60), 25)
    cfg = Configuration(0, 0, 0, 0)      # no sweeps: isolates Refresh (+ PreStep on the bodies)
    gb, gj, order, offs, _ = _device_solve(solver, state, cfg)
    bodies, cps, joints = state
    for k in range(len(joints)):
        want = oracle.refresh_joint(bodies, cps, joints[k])
        got = solver.refreshed(k)
        assert got.tobytes() == want.tobytes(), "joint %d" % k
    # PreStep alone (ref: Solver.cpp:697-758)
    ob_, oj, _ = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()


def test_per_iteration_prefix_parity(solver, oracle):
    """Rerun from identical state with k = 0..12 sweeps: every prefix of the iteration loop matches."""
    state = presolve_state(scenes.stack(4, 40), 3)
    for k in range(13):
        cfg = Configuration(0, 0, k, 0)
        gb, gj, order, offs, _ = _device_solve(solver, state, cfg)
        ob_, oj, _ = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
        assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes(), "after %d sweeps" % k


def test_edge_inputs(solver, oracle):
    # empty world, bodies without joints
    cfg = Configuration(0, 0, 5, 5)
    b = np.zeros(0, dtype=phyx_amd.rigid_body_dtype)
    st = solver.SolveJoints(b, np.zeros(0, dtype=phyx_amd.contact_point_dtype), np.zeros(0, dtype=phyx_amd.contact_joint_dtype), cfg)
    assert st.colour_count == 0
    bodies, cps, joints = presolve_state(scenes.stack(2, 10), 0)      # step 0: contacts exist, no warm start yet
    lonely = bodies.copy()
    before = lonely.copy()
    solver.SolveJoints(lonely, cps, joints[:0].copy(), cfg)
    assert lonely.tobytes() == before.tobytes()                         # no joints: velocities pass through
    # out-of-range indices are rejected, not dereferenced
    bad = joints.copy()
    bad["body2"][0] = len(bodies) + 5
    with pytest.raises(phyx_amd.PhxError):
        solver.SolveJoints(bodies.copy(), cps, bad, cfg)
    bad = joints.copy()
    bad["contact_point_index"][0] = len(cps)
    with pytest.raises(phyx_amd.PhxError):
        solver.SolveJoints(bodies.copy(), cps, bad, cfg)
    # a scene with two static bodies touching each other and a dynamic hub
    w = oracle.OracleWorld()
    w.add_body(0, 0, 0, 200, 10, static=True)
    w.add_body(0, 15, 0, 50, 5, static=True)          # static on static: joints with zero effective mass
    w.add_body(0, 40, 0, 40, 20)                       # big dynamic hub
    for i in range(12):
        w.add_body(-33 + 6 * i, 63, 0, 2.5, 3)         # 12 small boxes resting on the hub
    for _ in range(6):
        w.update()
    w.pre_solve()
    state = (w.bodies().copy(), w.contact_points().copy(), w.joints().copy())
    assert len(state[2]) > 20
    gb, gj, order, offs, st = _device_solve(solver, state, cfg)
    ob_, oj, _ = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    assert st.colour_count >= 10                                          # the hub serialises its contacts


def test_schedule_reuse_and_device_resident_path(oracle, built_lib):
    import os
    os.environ["PHX_GRAPHS"] = "1"                     # also exercise the optional hipGraph replay of the launch sequence
    try:
        solver = phyx_amd.Solver(0)
    finally:
        del os.environ["PHX_GRAPHS"]
    state = presolve_state(scenes.stack(10, 100), 3)
    cfg = Configuration(0, 0, 20, 20)
    gb, gj, order, offs, st1 = _device_solve(solver, state, cfg)
    d_b, d_cp, d_j = (phyx_amd.DeviceArray(a) for a in state)
    solver.SolveJointsDevice(d_b, d_cp, d_j, cfg)
    solver.synchronize()
    st2 = solver.stats()
    assert st2.recoloured == 0                                            # same topology: schedule reused
    assert d_b.to_host().tobytes() == gb.tobytes() and d_j.to_host().tobytes() == gj.tobytes()
    # third identical solve replays the captured hipGraphs and still matches
    import ctypes as C
    d_b2, d_j2 = phyx_amd.DeviceArray(state[0]), phyx_amd.DeviceArray(state[2])
    for rep in range(3):
        solver.L.phx_memcpy_h2d(0, d_b2.ptr, state[0].ctypes.data_as(C.c_void_p), state[0].nbytes)
        solver.L.phx_memcpy_h2d(0, d_j2.ptr, state[2].ctypes.data_as(C.c_void_p), state[2].nbytes)
        solver.SolveJointsDevice(d_b2, d_cp, d_j2, cfg)
        solver.synchronize()
        assert d_b2.to_host().tobytes() == gb.tobytes() and d_j2.to_host().tobytes() == gj.tobytes()
    assert solver.stats().graph_replay == 1
    # a different joint list invalidates the schedule
    state2 = (state[0], state[1], state[2][::-1].copy())
    _, _, order2, _, st3 = _device_solve(solver, state2, cfg)
    assert st3.recoloured == 1 and not np.array_equal(order.order, order2.order)


def test_island_groups_structure(solver, oracle):
    """Island-aware modes: groups are body-disjoint (static bodies aside), LDS groups respect the workgroup caps,
    and a scene that is one big island falls back to a single HBM group."""
    state = presolve_state(scenes.stack(24, 30), 3)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 15, 15)
    gb, gj, sched, _, st = _device_solve(solver, state, cfg)
    bodies, _, joints = state
    static = (bodies["inv_mass"] == 0) & (bodies["inv_inertia"] == 0)
    assert sched.lds_groups == len(sched.groups) - 1 >= 3 and st.lds_islands == sched.lds_groups
    owner = {}
    for g in range(len(sched.groups) - 1):
        sl = sched.order[sched.groups[g]:sched.groups[g + 1]]
        assert 0 < len(sl) <= 512
        for j in sl:
            for body in (int(joints["body1"][j]), int(joints["body2"][j])):
                if not static[body]:
                    assert owner.setdefault(body, g) == g
    ob_, oj, ost = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    assert st.joint_visits == ost.joint_visits
    # same scene in Single mode: one HBM group, and (islands being independent) the very same velocities
    sb, sj, ssched, _, sst = _device_solve(solver, state, Configuration(0, phyx_amd.ISLAND_SINGLE, 15, 15))
    assert ssched.lds_groups == 0 and len(ssched.groups) == 2
    assert sb.tobytes() == gb.tobytes()
    # a pile is one island of thousands of joints: too big for a workgroup -> HBM group
    pile = presolve_state(scenes.falling(2500, width=100.0, ymax=400.0), 50)
    pb, pj, psched, _, pst = _device_solve(solver, pile, cfg)
    assert len(pile[2]) > 3000 and psched.groups[-1] - psched.groups[-2] > 512
    ob_, oj, _ = _oracle_in_device_order(oracle, pile, psched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert pb.tobytes() == ob_.tobytes() and pj.tobytes() == oj.tobytes()


def test_full_size_200k_boxes(solver, oracle):
    """BASELINE config 2 size: 200 001 bodies, ~4e5 joints — bit-exact against the oracle in device order,
    plus the order-independent facts: the ground is untouched, impulses respect their clamps."""
    state = presolve_state(scenes.stack(1000, 200), 3, iters=20)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 20, 20)
    gb, gj, order, offs, st = _device_solve(solver, state, cfg)
    ob_, oj, ost = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    assert st.impulse_iterations == ost.impulse_iterations
    assert gb[0].tobytes() == state[0][0].tobytes()
    assert (gj["normal_acc"] >= 0).all()                                  # ref: Solver.cpp:847 clamp
    assert (np.abs(gj["friction_acc"]) <= 0.3 * gj["normal_acc"] * (1 + 1e-6) + 1e-12).all()   # Coulomb cone, ref: :872-883
    # different order (the reference's own AVX2 grouping) => different but statistically close result
    rb, cp, rj = (a.copy() for a in state)
    oracle.solver_solve(rb, cp, rj, oracle.SOLVE_AVX2, oracle.ISLAND_SINGLE, 20, 20)
    dv = np.abs(gb["velocity"]["y"] - rb["velocity"]["y"])
    assert np.isfinite(dv).all() and np.median(dv) < 1.0


def test_speculative_schedule_is_verified(solver, oracle):
    """Solves on the same array sizes reuse the cached schedule without waiting for the topology fingerprint; the
    device refuses to commit if it differs and the host then rebuilds and repeats.  Same joint count, different
    wiring: the result must still be the right one for the NEW joint list."""
    a = presolve_state(scenes.stack(6, 40), 3)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 15, 15)
    _device_solve(solver, a, cfg)                                   # builds the schedule for `a`
    perm = np.random.default_rng(3).permutation(len(a[2]))
    b = (a[0], a[1], a[2][perm].copy())                              # same sizes, joints re-ordered => other topology fingerprint
    gb, gj, sched, _, st = _device_solve(solver, b, cfg)
    assert st.recoloured == 1
    ob_, oj, _ = _oracle_in_device_order(oracle, b, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    # and a corrupted body index is reported, not dereferenced, even on the speculative path
    bad = b[2].copy()
    bad["body1"][5] = 10 ** 6
    with pytest.raises(phyx_amd.PhxError):
        solver.SolveJoints(b[0].copy(), b[1], bad, cfg)
    gb2, gj2, sched2, _, _ = _device_solve(solver, a, cfg)          # the solver still works afterwards
    ob2, oj2, _ = _oracle_in_device_order(oracle, a, sched2, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb2.tobytes() == ob2.tobytes() and gj2.tobytes() == oj2.tobytes()


def test_island_kernel_verifies_the_cached_schedule_itself(solver, oracle):
    """On a cached schedule with nothing but workgroup-sized islands no hash pass runs: the island kernel compares the joints and the
    bodies' static-ness it loads anyway with what the schedule recorded (ISL_VERIFY, csrc/island_view.h) and commits only if every
    workgroup of the launch agrees.  Each kind of difference must be caught — one joint re-wired, one contact-point index changed,
    one body turned static, one static body turned dynamic — and answered with a rebuild, never with a wrong result."""
    a = presolve_state(scenes.stack(8, 30), 3)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 12, 12)

    def check(state, must_rebuild):
        gb, gj, sched, _, st = _device_solve(solver, state, cfg)
        assert (st.recoloured != 0) == must_rebuild                   # (1: rebuilt, 2: rebuilt without a host round trip)
        ob_, oj, _ = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
        assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()

    check(a, True)                                                   # builds the schedule
    check(a, False)                                                  # verified by the island kernel: reused
    assert solver.stats().lds_islands > 0 and solver.stats().colour_count > 0
    j = a[2].copy()
    k = len(j) // 2
    j["body2"][k], j["body1"][k] = j["body1"][k], j["body2"][k]      # one joint's bodies swapped: same sizes, other topology
    check((a[0], a[1], j), True)
    check((a[0], a[1], j), False)
    j2 = j.copy()
    free = sorted(set(range(len(a[1]))) - set(j2["contact_point_index"].tolist()))
    if free:                                                         # a joint moved to an unused contact point slot (another colouring priority)
        j2["contact_point_index"][3] = free[0]
        check((a[0], a[1], j2), True)
    b = a[0].copy()
    dyn = int(j["body1"][7])
    b["inv_mass"][dyn] = 0.0; b["inv_inertia"][dyn] = 0.0             # a dynamic body turned static
    check((b, a[1], j), True)
    check((b, a[1], j), False)
    b2 = b.copy()
    b2["inv_mass"][0] = 1e-3; b2["inv_inertia"][0] = 1e-5             # the ground turned dynamic: every column joins one island
    check((b2, a[1], j), True)
    check(a, True)                                                   # and back


def test_island_groups_left_uncommitted_are_completed(oracle, built_lib, monkeypatch):
    """A workgroup of a verified launch commits only after every workgroup has arrived; its wait is bounded (a GPU shared with
    somebody else's kernels may not hold them all at once).  With the bound forced to zero every workgroup gives up: the solve must
    be completed by the second launch (ISL_COMPLETE) and still be the oracle's, bit for bit."""
    monkeypatch.setenv("PHX_ISL_WAIT_POLLS", "0")
    solver = phyx_amd.Solver(0)
    monkeypatch.delenv("PHX_ISL_WAIT_POLLS")
    state = presolve_state(scenes.stack(12, 25), 3)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 15, 15)
    first = _device_solve(solver, state, cfg)                       # builds
    for _ in range(2):                                               # verified launches that all time out
        gb, gj, sched, _, st = _device_solve(solver, state, cfg)
        assert st.recoloured == 0
        assert gb.tobytes() == first[0].tobytes() and gj.tobytes() == first[1].tobytes()
        ob_, oj, ost = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
        assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
        assert st.impulse_iterations == ost.impulse_iterations and st.joint_visits == ost.joint_visits


@pytest.mark.parametrize("gate", ["in_kernel", "hash"])
def test_two_queued_solves_on_a_stale_schedule_are_both_replayed(built_lib, oracle, monkeypatch, gate):
    """Solver-only sub-stepping: two SolveJoints queued back to back on the same device arrays, no synchronisation asked for in
    between, while the cached schedule is stale.  The arrays must end up holding two successive solves of the new joint list, not
    one.  Hash-gated solves (PHX_NO_FUSED_VERIFY=1) chain: neither commits, synchronize() rebuilds and replays BOTH.  Solves whose
    island launch checks the schedule itself do not chain — the next solve's first kernel clears the control word a timed-out
    workgroup would have marked — so the second call settles the first (rebuild + replay) and then runs on the fresh schedule."""
    if gate == "hash":
        monkeypatch.setenv("PHX_NO_FUSED_VERIFY", "1")
    solver = phyx_amd.Solver(0)
    a = presolve_state(scenes.stack(6, 40), 3)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 10, 10)
    db, dc, dj = (phyx_amd.DeviceArray(x) for x in a)
    solver.SolveJointsDevice(db, dc, dj, cfg)
    solver.synchronize()                                             # schedule cached for `a`
    perm = np.random.default_rng(11).permutation(len(a[2]))
    b = (a[0].copy(), a[1], a[2][perm].copy())
    import ctypes as C
    from phyx_amd import _lib
    L = _lib.load()
    _lib.check(L.phx_memcpy_h2d(0, db.ptr, b[0].ctypes.data_as(C.c_void_p), b[0].nbytes))
    _lib.check(L.phx_memcpy_h2d(0, dj.ptr, b[2].ctypes.data_as(C.c_void_p), b[2].nbytes))
    solver.SolveJointsDevice(db, dc, dj, cfg)                        # speculative on the stale schedule
    solver.SolveJointsDevice(db, dc, dj, cfg)                        # and again, before the first was verified
    solver.synchronize()
    assert solver.stats().recoloured == (1 if gate == "hash" else 0)
    sched = Sched(solver)
    ob_, cp, oj = (x.copy() for x in b)
    for _ in range(2):
        oracle.solver_solve_grouped(ob_, cp, oj, sched.order, sched.colours, sched.groups, 10, 10, oracle.STAG_COLOUR_SYNC)
    assert db.to_host().tobytes() == ob_.tobytes() and dj.to_host().tobytes() == oj.tobytes()


def test_tall_columns_use_the_1024_lane_island_shape(solver, oracle):
    """Columns of 500 boxes are ~1020 joints each: too big for the 512-lane workgroup, they take the 1024-lane
    shape instead of falling back to HBM (BASELINE config 5 geometry, 50 iterations)."""
    state = presolve_state(scenes.stack(6, 500), 3, iters=50)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 50, 50)
    gb, gj, sched, _, st = _device_solve(solver, state, cfg)
    sizes = np.diff(sched.groups)
    assert sched.lds_groups == len(sizes) and sizes.max() > 512 and sizes.max() <= 1024
    ob_, oj, ost = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    assert st.impulse_iterations == ost.impulse_iterations and st.joint_visits == ost.joint_visits


def test_full_size_500k_boxes_50_iterations(solver, oracle):
    """BASELINE config 5 size: stack(1000,500) = 500 001 bodies, ~1e6 joints, 50+50 iterations (fp32 body state)."""
    state = presolve_state(scenes.stack(1000, 500), 2, iters=50)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 50, 50)
    gb, gj, sched, _, st = _device_solve(solver, state, cfg)
    assert len(state[2]) > 900000 and st.lds_islands >= 900
    ob_, oj, ost = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    assert (gj["normal_acc"] >= 0).all()


def test_full_size_500k_boxes_fp16_body_state(oracle, built_lib):
    """BASELINE config 5's ablation AT ITS SIZE: stack(1000,500), 50+50 iterations, body velocities kept in IEEE half between joint
    updates — bit for bit the oracle's model of that rounding, and a bounded distance from the fp32 solve of the same input."""
    state = presolve_state(scenes.stack(1000, 500), 2, iters=50)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 50, 50)
    s16 = phyx_amd.Solver(0)
    s16.set_body_state_bits(16)
    hb, hj, sched, _, st = _device_solve(s16, state, cfg)
    assert len(state[2]) > 900000 and st.lds_islands >= 900 and sched.lds_groups == st.lds_islands
    b, cp, j = (a.copy() for a in state)
    oracle.solver_solve_grouped(b, cp, j, sched.order, sched.colours, sched.groups, 50, 50, oracle.STAG_COLOUR_SYNC, fp16_groups=sched.lds_groups)
    assert hb.tobytes() == b.tobytes() and hj.tobytes() == j.tobytes()
    fb, fj, _, _, _ = _device_solve(phyx_amd.Solver(0), state, cfg)
    dv = np.abs(hb["velocity"]["y"] - fb["velocity"]["y"])
    assert np.isfinite(hb["velocity"]["y"]).all() and hb.tobytes() != fb.tobytes()
    assert dv.max() < 8.0 and dv.mean() < 0.05, (dv.max(), dv.mean())       # half: ~3 decimal digits on velocities of O(1..100)


def test_device_schedule_builder_equals_host_builder(oracle, built_lib):
    """Schedules are built on the device (connected components by atomicMin linking, per-bin colouring in LDS, the HBM
    group coloured by Jones-Plassmann rounds in HBM); the host builder is the specification.  Both must produce the
    very same schedule, in the island modes and in Single mode."""
    import os
    os.environ["PHX_SCHEDULE_BUILDER"] = "host"
    try:
        host_solver = phyx_amd.Solver(0)
    finally:
        del os.environ["PHX_SCHEDULE_BUILDER"]
    dev_solver = phyx_amd.Solver(0)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 15, 15)
    cases = [presolve_state(SMALL_SCENES[n][0](), SMALL_SCENES[n][1]) for n in SMALL_SCENES]
    cases.append(presolve_state(scenes.stack(40, 60), 4))                       # many bins of several components each
    cases.append(presolve_state(scenes.stack(5, 500), 3, iters=30))             # 1024-lane shape
    cases.append(presolve_state(scenes.falling(2500, width=100.0, ymax=400.0), 50))   # one huge island + loose boxes -> HBM group
    single = Configuration(0, phyx_amd.ISLAND_SINGLE, 15, 15)
    for state, cfg in [(st, cfg) for st in cases] + [(st, single) for st in cases]:
        hb, hj, hs, _, hst = _device_solve(host_solver, state, cfg)
        db, dj, ds, _, dst = _device_solve(dev_solver, state, cfg)
        assert np.array_equal(hs.order, ds.order)
        assert np.array_equal(hs.colours, ds.colours)
        assert np.array_equal(hs.groups, ds.groups) and hs.lds_groups == ds.lds_groups
        assert (hst.island_count, hst.island_max_size, hst.colour_count) == (dst.island_count, dst.island_max_size, dst.colour_count)
        assert hb.tobytes() == db.tobytes() and hj.tobytes() == dj.tobytes()


def test_device_lanes_are_the_host_builders(built_lib):
    """The lanes the device builder (k_build_bin: the classes' ranges laid out on the wave's scalar unit) gives the units of the LDS groups
    are the ones the host builder's layout_classes gives them (phx_schedule_groups, restated in tests/test_host_logic.py): one lane per
    unit, classes on wave boundaries where the lanes allow it — for the small shape (stack columns of 100), the 512-lane shape (columns
    of 500) and a scene of several components per group."""
    solver = phyx_amd.Solver(0)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 15, 15)
    for scene, warm, iters, lanes, cap in ((scenes.stack(10, 100), 3, 15, 256, 768), (scenes.stack(3, 500), 3, 30, 512, 1024), (scenes.stack(40, 60), 4, 15, 256, 768)):
        state = presolve_state(scene, warm, iters=iters)
        bodies, _, joints = state
        _, _, sched, _, st = _device_solve(solver, state, cfg)
        slot, lane = solver.lanes()
        hs = phyx_amd.schedule_groups(joints["body1"], joints["body2"], is_static(bodies), joints["contact_point_index"], lanes=lanes, body_cap=cap)
        assert hs["lds_groups"] == sched.lds_groups and np.array_equal(hs["order"], sched.order)
        want = dict(zip(hs["unit_leader_slot"].tolist(), hs["unit_lane"].tolist()))
        got = dict(zip(slot.tolist(), lane.tolist()))
        assert len(got) == len(slot) and got == want
        # ... and they do what they are for: no class straddles a wave more than back-to-back ranges would make it
        grp = np.searchsorted(np.asarray(sched.groups), slot, side="right") - 1
        cls = np.searchsorted(np.asarray(sched.colours), slot, side="right") - 1
        passes = len(np.unique(np.stack([grp, cls, lane // 64], axis=1), axis=0))
        plain = 0
        for g in range(sched.lds_groups):
            at = 0
            for c in np.unique(cls[grp == g]):
                n = int(((grp == g) & (cls == c)).sum())
                plain += (at + n - 1) // 64 - at // 64 + 1
                at += n
        assert passes <= plain


@pytest.mark.parametrize("island_mode", [phyx_amd.ISLAND_MULTIPLE, phyx_amd.ISLAND_SINGLE])
def test_partitioned_component_is_swept_by_parts(oracle, built_lib, monkeypatch, island_mode):
    """A connected component of more than 1024 joints (here a brick wall: one island of 1.1e4 joints — the shape of a settled pile)
    is partitioned: units inside one block of 512 bodies get the leading classes of the HBM group and ONE launch per sweep
    sweeps them all (k_solve_parts), the boundary classes stay one launch each.  Device builder == host builder, the fused
    launch == one launch per class (PHX_NO_PARTS=1, with and without k_solve_tail's one launch for the trailing tiny classes), and all of
    them == the oracle's replay of the exported schedule."""
    import os
    state = presolve_state(scenes.wall(48, 60), 10)
    assert len(state[2]) > 1024
    ci, pi = 12, 6
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, island_mode, ci, pi)
    dev = phyx_amd.Solver(0)
    monkeypatch.setenv("PHX_SCHEDULE_BUILDER", "host")
    host = phyx_amd.Solver(0)
    monkeypatch.delenv("PHX_SCHEDULE_BUILDER")
    monkeypatch.setenv("PHX_NO_PARTS", "1")
    tailed = phyx_amd.Solver(0)                 # no parts: every class a launch of its own — but for the trailing tiny ones, which k_solve_tail sweeps in one
    monkeypatch.setenv("PHX_NO_TAIL", "1")
    plain = phyx_amd.Solver(0)
    monkeypatch.delenv("PHX_NO_PARTS")
    monkeypatch.delenv("PHX_NO_TAIL")
    gb, gj, sched, _, st = _device_solve(dev, state, cfg)
    ki, parts, launches = dev.partition()
    classes = len(sched.colours) - 1
    assert st.lds_islands == 0 and ki >= 4 and classes - ki >= 1 and parts == 2 * ((len(state[0]) + 511) // 512) + 1      # both levels
    sweeps = max(st.impulse_iterations, st.displacement_iterations)
    assert launches <= max(ci, pi) * (2 + classes - ki)            # one launch per level for the interior classes + one per rest class (or fewer: the tail)
    hb, hj, hsched, _, hst = _device_solve(host, state, cfg)
    assert np.array_equal(hsched.order, sched.order) and np.array_equal(hsched.colours, sched.colours) and np.array_equal(hsched.groups, sched.groups)
    assert host.partition()[:2] == (ki, parts)
    assert hb.tobytes() == gb.tobytes() and hj.tobytes() == gj.tobytes()
    pb, pj, psched, _, pst = _device_solve(plain, state, cfg)
    assert np.array_equal(psched.order, sched.order) and np.array_equal(psched.colours, sched.colours)
    assert plain.partition() == (ki, 0, max(ci, pi) * classes)
    assert pb.tobytes() == gb.tobytes() and pj.tobytes() == gj.tobytes()
    assert (pst.impulse_iterations, pst.displacement_iterations, pst.joint_visits) == (st.impulse_iterations, st.displacement_iterations, st.joint_visits)
    # ... and the trailing classes of at most 1024 units in ONE workgroup's launch (k_solve_tail): fewer launches, the same bytes
    tb, tj, tsched, _, tst = _device_solve(tailed, state, cfg)
    assert np.array_equal(tsched.order, sched.order) and np.array_equal(tsched.colours, sched.colours)
    tki, tparts, tlaunches = tailed.partition()
    assert (tki, tparts) == (ki, 0) and tlaunches < max(ci, pi) * classes and tlaunches % max(ci, pi) == 0
    assert tb.tobytes() == gb.tobytes() and tj.tobytes() == gj.tobytes()
    assert (tst.impulse_iterations, tst.displacement_iterations, tst.joint_visits) == (st.impulse_iterations, st.displacement_iterations, st.joint_visits)
    ob_, oj, ost = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    assert (st.impulse_iterations, st.displacement_iterations) == (ost.impulse_iterations, ost.displacement_iterations) and sweeps > 0
    # interior units: both bodies dynamic, one block of 512 indices of the plain or of the shifted grid; nothing else in the first ki classes
    b1, b2 = state[2]["body1"], state[2]["body2"]
    lead = sched.order[:sched.colours[ki]]
    assert (((b1[lead] // 512) == (b2[lead] // 512)) | (((b1[lead] + 256) // 512) == ((b2[lead] + 256) // 512))).all()
    assert (state[0]["inv_mass"][b1[lead]] > 0).all() and (state[0]["inv_mass"][b2[lead]] > 0).all()
    # a second solve on the cached schedule (fingerprint-gated) gives the same bytes
    gb2, gj2, _, _, st2 = _device_solve(dev, state, cfg)
    assert st2.recoloured == 0 and gb2.tobytes() == gb.tobytes() and gj2.tobytes() == gj.tobytes()


def test_speculative_binning_equals_the_builders_long_way(oracle, built_lib):
    """A rebuild of a world that was nothing but workgroup-sized islands last time makes its bins on the device, with last build's
    bin count as the launch grid and no host round trip (stats.recoloured == 2).  Whatever does not hold any more — more bins than
    the grid, the other workgroup shape, an island too big for a workgroup — spoils the solve, which is then rebuilt the long
    way and repeated.  Schedule, statistics and results must equal those of a solver that never speculates."""
    import os
    os.environ["PHX_NO_SPEC_BINS"] = "1"
    try:
        plain = phyx_amd.Solver(0)
    finally:
        del os.environ["PHX_NO_SPEC_BINS"]
    spec = phyx_amd.Solver(0)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 12, 12)
    seq = [("stack40x60", presolve_state(scenes.stack(40, 60), 4), 1),            # first build: the long way
           ("again", presolve_state(scenes.stack(40, 60), 5), 2),                  # same world one step on: speculative
           ("fewer bins", presolve_state(scenes.stack(25, 60), 4), 2),
           ("more bins than the grid", presolve_state(scenes.stack(300, 60), 4), 1),
           ("again", presolve_state(scenes.stack(300, 60), 5), 2),
           ("the roomier shape", presolve_state(scenes.stack(5, 500), 3, iters=30), 1),
           ("again", presolve_state(scenes.stack(5, 500), 4, iters=30), 2),
           ("back to the small shape", presolve_state(scenes.stack(40, 60), 4), 1),
           ("again", presolve_state(scenes.stack(40, 60), 5), 2),
           ("an island for the HBM group", presolve_state(scenes.falling(2500, width=100.0, ymax=400.0), 50), 1),
           ("after it", presolve_state(scenes.stack(40, 60), 4), 1),               # (the last build had an HBM group: no speculation)
           ("again", presolve_state(scenes.stack(40, 60), 5), 2),
           ("tiny", presolve_state(scenes.stack(2, 10), 2), 2),
           ("more components than one window of the binning kernel holds", presolve_state(scenes.stack(9000, 3), 3), 1),      # (more bins than the grid)
           ("again: two windows", presolve_state(scenes.stack(9000, 3), 4), 2),
           ("and three", presolve_state(scenes.stack(17000, 2), 3), 2)]
    for what, state, want in seq:
        pb, pj, ps, _, pst = _device_solve(plain, state, cfg)
        sb, sj, ss, _, sst = _device_solve(spec, state, cfg)
        assert pst.recoloured == 1 and sst.recoloured == want, (what, sst.recoloured)
        assert np.array_equal(ps.order, ss.order) and np.array_equal(ps.colours, ss.colours), what
        assert np.array_equal(ps.groups, ss.groups) and ps.lds_groups == ss.lds_groups, what
        assert (pst.island_count, pst.island_max_size, pst.colour_count, pst.lds_islands) == (sst.island_count, sst.island_max_size, sst.colour_count, sst.lds_islands), what
        assert (pst.impulse_iterations, pst.displacement_iterations, pst.joint_visits) == (sst.impulse_iterations, sst.displacement_iterations, sst.joint_visits), what
        assert pb.tobytes() == sb.tobytes() and pj.tobytes() == sj.tobytes(), what


def test_bench_on_staged_copies(solver):
    """bench() on copies of the input staged before the clock starts (phx_solver_bench_stage) does the same work as bench() with
    the restore copies inside the timed region: same sweeps, same visits; the copies are consumed by one call."""
    state = presolve_state(scenes.stack(8, 30), 3)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 10, 10)
    d = [phyx_amd.DeviceArray(a) for a in state]
    plain = solver.bench(*d, cfg, 1, 6)
    solver.bench_stage(d[0], d[2], 6)
    staged = solver.bench(*d, cfg, 0, 6)
    again = solver.bench(*d, cfg, 0, 6)                # nothing staged any more: restores in front of every step
    for r in (staged, again):
        assert (r.joint_visits, r.impulse_iterations, r.impulse_launches) == (plain.joint_visits, plain.impulse_iterations, plain.impulse_launches)
    assert d[0].to_host().tobytes() == state[0].tobytes() and d[2].to_host().tobytes() == state[2].tobytes()      # the caller's arrays stay untouched
    # the live-topology mode (a rebuild inside every step) works on staged copies too
    solver.set_schedule_reuse(False)
    try:
        solver.bench_stage(d[0], d[2], 4)
        live = solver.bench(*d, cfg, 0, 4)
    finally:
        solver.set_schedule_reuse(True)
    assert live.impulse_iterations * 6 == plain.impulse_iterations * 4


def test_bench_step_hook(solver):
    """phx_solver_bench_hooked calls back once per queued step (bench.py enqueues the per-step all-reduce there); an
    exception raised in the hook aborts the run and reaches the caller, and the handle stays usable."""
    state = presolve_state(scenes.stack(8, 30), 3)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 10, 10)
    d = [phyx_amd.DeviceArray(a) for a in state]
    assert solver.stream_ptr() != 0
    seen = []
    plain = solver.bench(*d, cfg, 1, 5)
    hooked = solver.bench(*d, cfg, 1, 5, hook=lambda step, phase: seen.append((step, phase)))
    # every step: phase 1 before its sweeps, phase 0 once it is queued; one warm-up step (-1), five timed ones, a final drain
    assert seen == [(-1, 1), (-1, 0)] + [(k, p) for k in range(5) for p in (1, 0)] + [(5, 1)]
    assert hooked.joint_visits == plain.joint_visits and hooked.impulse_iterations == plain.impulse_iterations

    def boom(step, phase):
        if step == 2 and phase == 0:
            raise RuntimeError("hook failed on purpose")
    with pytest.raises(RuntimeError, match="on purpose"):
        solver.bench(*d, cfg, 0, 5, hook=boom)
    again = solver.bench(*d, cfg, 0, 3)
    assert again.joint_visits * 5 == plain.joint_visits * 3


def _random_state(rng, nb, nj, static_frac, dup_ids=False, hub=0, units=False):
    """A synthetic solver input with an arbitrary contact graph: random pairs over nb bodies (a few static), plausible
    small offsets and unit normals, so that the arithmetic stays finite while the topology is nothing like a stack."""
    bodies = np.zeros(nb, dtype=phyx_amd.rigid_body_dtype)
    bodies["index"] = np.arange(nb)
    bodies["inv_mass"] = rng.uniform(0.5, 2.0, nb)
    bodies["inv_inertia"] = rng.uniform(0.01, 0.1, nb)
    static = rng.random(nb) < static_frac
    bodies["inv_mass"][static] = 0
    bodies["inv_inertia"][static] = 0
    bodies["pos"]["x"] = rng.uniform(-100, 100, nb)
    bodies["pos"]["y"] = rng.uniform(-100, 100, nb)
    bodies["velocity"]["x"] = np.where(static, 0.0, rng.uniform(-1, 1, nb))      # (+0.0 exactly: see DESIGN §4.3 on -0.0 statics)
    bodies["velocity"]["y"] = np.where(static, 0.0, rng.uniform(-1, 1, nb))
    b1 = rng.integers(0, nb, nj)
    b2 = (b1 + rng.integers(1, nb, nj)) % nb                           # never the same body twice
    if hub:
        b1[:hub] = 0                                                   # body 0 made a hub touched by `hub` joints
        bodies["inv_mass"][0], bodies["inv_inertia"][0] = 1.0, 0.05
    cps = np.zeros(nj, dtype=phyx_amd.contact_point_dtype)
    ang = rng.uniform(0, 2 * np.pi, nj)
    cps["normal"]["x"], cps["normal"]["y"] = np.cos(ang), np.sin(ang)
    for d in ("delta1", "delta2"):
        cps[d]["x"] = rng.uniform(-5, 5, nj)
        cps[d]["y"] = rng.uniform(-5, 5, nj)
    joints = np.zeros(nj, dtype=phyx_amd.contact_joint_dtype)
    joints["body1"], joints["body2"] = b1, b2
    joints["contact_point_index"] = rng.permutation(nj) if not dup_ids else rng.integers(0, max(nj // 4, 1), nj)
    joints["normal_acc"] = rng.uniform(0, 0.1, nj)
    if units:
        # couples of joints on one body pair with contact points 2m / 2m+1 (units, csrc/schedule.h), scattered over the joint
        # array — and the near misses that must NOT pair: the follower's bodies swapped, a third joint claiming the odd id, a
        # lone even id, an odd id whose even partner sits on other bodies
        half = nj // 2
        perm = rng.permutation(nj)
        lead, foll = perm[:half], perm[half:2 * half]
        joints["body1"][foll], joints["body2"][foll] = joints["body1"][lead], joints["body2"][lead]
        joints["contact_point_index"][lead] = 2 * np.arange(half)
        joints["contact_point_index"][foll] = 2 * np.arange(half) + 1
        k = half // 10
        swap = foll[:k]
        joints["body1"][swap], joints["body2"][swap] = joints["body2"][swap].copy(), joints["body1"][swap].copy()
        dup = foll[k:2 * k]
        joints["contact_point_index"][dup] = joints["contact_point_index"][foll[2 * k:3 * k]]          # two joints carry one odd id
        other = foll[3 * k:4 * k]
        joints["body2"][other] = (joints["body2"][other] + 1) % nb
        joints["body2"][other] = np.where(joints["body2"][other] == joints["body1"][other], (joints["body2"][other] + 1) % nb, joints["body2"][other])
    return bodies, cps, joints


def _priority_chain_state(n):
    """A chain of n joints whose colouring priorities decrease monotonically along the chain: every Jones-Plassmann round
    can colour exactly one joint, so the device needs n rounds — more than it allows itself before handing the HBM group
    to the host builder."""
    rng = np.random.default_rng(11)
    bodies, cps, joints = _random_state(rng, n + 1, n, 0.0)
    joints["body1"] = np.arange(n)
    joints["body2"] = np.arange(1, n + 1)
    hi = np.array([phyx_amd.schedule_priority(i, 0) >> 32 for i in range(n)])
    joints["contact_point_index"] = np.argsort(-hi, kind="stable")                  # a permutation of the n contact points
    return bodies, cps, joints


def test_pathological_priority_chain_falls_back_to_the_host_builder(oracle, built_lib):
    state = _priority_chain_state(700)
    solver = phyx_amd.Solver(0)
    cfg = Configuration(0, phyx_amd.ISLAND_SINGLE, 3, 3)
    gb, gj, sched, _, st = _device_solve(solver, state, cfg)
    assert sorted(sched.order.tolist()) == list(range(700)) and st.colour_count == 2      # a path needs two colours
    ob_, oj, _ = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
    # the island-aware mode colours such a chain inside one workgroup (no round limit there); 500 single-joint units fit its 512 lanes
    state = _priority_chain_state(500)
    cfg = Configuration(0, phyx_amd.ISLAND_MULTIPLE, 3, 3)
    gb, gj, sched, _, st = _device_solve(solver, state, cfg)
    assert st.lds_islands == 1 and st.colour_count == 2
    ob_, oj, _ = _oracle_in_device_order(oracle, state, sched, None, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()


@pytest.mark.parametrize("case", ["sparse", "dense", "dup_ids", "mostly_static", "hub_over_64_colours", "units_and_near_misses"])
def test_random_contact_graphs_both_builders_and_oracle(oracle, built_lib, case):
    """Arbitrary contact graphs (not stacks): the device builder must reproduce the host builder's schedule — including
    duplicated priority ids (ties broken by joint index) and a hub that needs more than 64 colours (device falls back to
    the host builder) — and the solve must match the oracle's replay of that schedule bit for bit."""
    import os
    rng = np.random.default_rng({"sparse": 1, "dense": 2, "dup_ids": 3, "mostly_static": 4, "hub_over_64_colours": 5, "units_and_near_misses": 6}[case])
    nb, nj, sf, dup, hub = {"sparse": (4000, 3000, 0.05, False, 0), "dense": (600, 6000, 0.05, False, 0), "dup_ids": (2000, 5000, 0.1, True, 0),
                            "mostly_static": (3000, 4000, 0.7, False, 0), "hub_over_64_colours": (1500, 2500, 0.05, False, 90),
                            "units_and_near_misses": (3000, 6000, 0.1, False, 0)}[case]
    state = _random_state(rng, nb, nj, sf, dup, hub, units=case == "units_and_near_misses")
    os.environ["PHX_SCHEDULE_BUILDER"] = "host"
    try:
        host_solver = phyx_amd.Solver(0)
    finally:
        del os.environ["PHX_SCHEDULE_BUILDER"]
    dev_solver = phyx_amd.Solver(0)
    for island_mode in (phyx_amd.ISLAND_SINGLE, phyx_amd.ISLAND_MULTIPLE):
        cfg = Configuration(0, island_mode, 4, 3)
        hb, hj, hs, _, hst = _device_solve(host_solver, state, cfg)
        db, dj, ds, _, dst = _device_solve(dev_solver, state, cfg)
        assert np.array_equal(hs.order, ds.order) and np.array_equal(hs.colours, ds.colours) and np.array_equal(hs.groups, ds.groups)
        assert sorted(ds.order.tolist()) == list(range(nj))
        if case == "hub_over_64_colours":
            assert dst.colour_count >= 90
        ob_, oj, _ = _oracle_in_device_order(oracle, state, ds, None, cfg, oracle.STAG_COLOUR_SYNC)
        assert np.isfinite(db["velocity"]["x"]).all()
        assert db.tobytes() == ob_.tobytes() and dj.tobytes() == oj.tobytes()
        assert hb.tobytes() == db.tobytes() and hj.tobytes() == dj.tobytes()


def test_fp16_body_state_ablation(oracle, built_lib):
    """BASELINE config 5's ablation: body velocities kept in IEEE half between joint updates (fp32 arithmetic).  The device
    must agree bit for bit with the oracle's model of that rounding, and stay close to the fp32 result."""
    s16 = phyx_amd.Solver(0)
    s16.set_body_state_bits(16)
    s32 = phyx_amd.Solver(0)
    for state, iters in ((presolve_state(scenes.stack(6, 60), 3), 20), (presolve_state(scenes.stack(3, 500), 3, iters=50), 50),
                         (presolve_state(scenes.tilted(60), 25), 15)):
        cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, iters, iters)
        hb, hj, sched, _, st = _device_solve(s16, state, cfg)
        b, cp, j = (a.copy() for a in state)
        oracle.solver_solve_grouped(b, cp, j, sched.order, sched.colours, sched.groups, iters, iters, oracle.STAG_COLOUR_SYNC,
                                    fp16_groups=sched.lds_groups)
        assert hb.tobytes() == b.tobytes() and hj.tobytes() == j.tobytes()
        fb, fj, _, _, _ = _device_solve(s32, state, cfg)
        dv = np.abs(hb["velocity"]["y"] - fb["velocity"]["y"])
        assert np.isfinite(hb["velocity"]["y"]).all() and hb.tobytes() != fb.tobytes()
        assert dv.max() < 2.0, dv.max()          # half has ~3 decimal digits; velocities here are O(1..10)
    with pytest.raises(phyx_amd.PhxError):
        s16.set_body_state_bits(8)


@pytest.mark.parametrize("case", ["cfg2", "cfg5"])
def test_static_tag_rule_deviation_at_full_size(solver, oracle, case):
    """DESIGN §9.4, measured: the device gives every group a private, class-synchronous copy of a static body's lastIteration tag and
    lets every group leave its sweeps on its own; the reference's Single modes keep ONE word per static body (ref: Solver.cpp:474-478,
    790-798, 900-910) and ONE early exit for the whole joint list (:189) — the ground couples the skip decisions of all columns.
    The device's order is replayed under the reference's rule (phxo_solver_solve_ordered, PHXO_STAG_SEQUENTIAL: one island, one shared
    tag, sequential visibility, global early exit).  The device equals the oracle bit for bit in ITS rule; against the reference's rule
    it differs in a few hundred to a few thousand bodies (0.3 %) by impulses of a few convergence thresholds (1e-4, ref: :895) — ground
    contacts evaluated once more or once less.  SURVEY §8(c)'s T1 (1e-3 in a velocity) is NOT met and cannot be by independent groups
    (measured: cfg 2 max |d impulse| 2.9e-4, |d velocity| 1.14, |d position| after the step 0.019; cfg 5 1.4e-3 / 2.03 / 0.034).  The
    stated tolerance of this backend against the reference's Single-mode rule: 5e-3 in an accumulated impulse (50 thresholds), 5 in a
    velocity, 0.1 in a position after the step's integration, at most 1 % of the bodies touched.  (tools/static_tag_deviation.py prints
    the numbers; profiles/r06_static_tag_deviation.json keeps them.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from static_tag_deviation import CASES, deviation
    c, r, it = CASES[case]
    d = deviation(c, r, it, solver=solver)
    print(d)
    assert d["device_equals_oracle_in_device_rule"]
    assert d["max_abs_dimpulse"] <= 5e-3, d
    assert d["max_abs_dvel"] <= 5.0 and d["max_abs_dpos_after_integrate"] <= 0.1, d
    assert d["max_abs_ddisplacing"] <= 1e-3, d
    assert d["bodies_differing"] <= 0.01 * d["bodies"], d
    assert abs(d["device_impulse_sweeps_max"] - d["reference_rule_impulse_sweeps"]) <= 2, d
