"""Parity of the HIP solver against the oracle (MI355X).  The device sweeps the joints in its own colour
order; Gauss-Seidel results depend on that order, so the oracle is run in the SAME order
(phx_solver_get_schedule) — then every velocity and every accumulated impulse must match bit for bit
(fp32, strict IEEE on both sides).  Tolerances appear only where two different orders are compared."""
import numpy as np
import pytest

import phyx_amd
from phyx_amd import scenes, Configuration
from helpers import SMALL_SCENES, presolve_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(built_lib):
    return phyx_amd.Solver(0)


def _device_solve(solver, state, cfg):
    b, cp, j = (a.copy() for a in state)
    st = solver.SolveJoints(b, cp, j, cfg)
    order, offs = solver.schedule()
    return b, j, order, offs, st


def _oracle_in_device_order(oracle, state, order, offs, cfg, mode):
    b, cp, j = (a.copy() for a in state)
    st = oracle.solver_solve_ordered(b, cp, j, order, offs, cfg.contactIterationsCount, cfg.penetrationIterationsCount, mode)
    return b, j, st


@pytest.mark.parametrize("name", list(SMALL_SCENES))
@pytest.mark.parametrize("iters", [(15, 15), (20, 20), (7, 0), (0, 5), (1, 1)])
def test_bit_exact_vs_oracle_in_device_order(solver, oracle, name, iters):
    make, warm = SMALL_SCENES[name]
    state = presolve_state(make(), warm)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_SINGLE, iters[0], iters[1])
    gb, gj, order, offs, st = _device_solve(solver, state, cfg)
    ob_, oj, ost = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_COLOUR_SYNC)
    assert gb.tobytes() == ob_.tobytes(), "body velocities differ from the oracle"
    assert gj.tobytes() == oj.tobytes(), "accumulated impulses differ from the oracle"
    assert st.impulse_iterations == ost.impulse_iterations
    assert st.displacement_iterations == ost.displacement_iterations
    # the reference's sequential static-tag rule gives the same answer on these scenes (0 divergent decisions)
    sb, sj, sst = _oracle_in_device_order(oracle, state, order, offs, cfg, oracle.STAG_SEQUENTIAL)
    assert sst.stag_events == 0 and sb.tobytes() == gb.tobytes() and sj.tobytes() == gj.tobytes()


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
    assert st.colour_count >= 20                                          # the hub serialises its contacts


def test_schedule_reuse_and_device_resident_path(solver, oracle):
    state = presolve_state(scenes.stack(10, 100), 3)
    cfg = Configuration(0, 0, 20, 20)
    gb, gj, order, offs, st1 = _device_solve(solver, state, cfg)
    d_b, d_cp, d_j = (phyx_amd.DeviceArray(a) for a in state)
    solver.SolveJointsDevice(d_b, d_cp, d_j, cfg)
    solver.synchronize()
    st2 = solver.stats()
    assert st2.recoloured == 0                                            # same topology: schedule reused
    assert d_b.to_host().tobytes() == gb.tobytes() and d_j.to_host().tobytes() == gj.tobytes()
    # a different joint list invalidates it
    state2 = (state[0], state[1], state[2][::-1].copy())
    _, _, order2, _, st3 = _device_solve(solver, state2, cfg)
    assert st3.recoloured == 1 and not np.array_equal(order, order2)


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
