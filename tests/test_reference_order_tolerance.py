"""The north star's last parity clause: "results match the reference CPU AVX2 path on the same scene to a stated float
tolerance on body positions/velocities after K steps".

Sequential impulse is order dependent, and the reference's own solve modes do not agree with each other: scalar (N=1)
and AVX2 (N=8) sweep the joints in different orders (PrepareIndices regroups them, ref: Solver.cpp:217-273) and drift
apart chaotically.  The device's colour order is a third order.  The tolerance is therefore stated
relative to the reference's own cross-mode band, on BASELINE config 1's scene (1k boxes, 20 iterations), with the
reference orders taken from the oracle's restatement of the N=1 and N=8 paths:

    K = 1 step :  mean |dpos| <= 1e-3,  max |dpos| <= 0.1   (boxes are 10 units wide, fall at up to ~300 units/s)
                  mean |dvel| <= 0.05
    K <= 10    :  mean |dpos|(device, AVX2) <= 2 x mean |dpos|(scalar, AVX2)      and the same for velocities

The CPU test drives the host schedule builder + the oracle's replay (which the GPU tests prove bit-identical to the
device); the GPU test drives the device-resident World itself.
"""
import numpy as np
import pytest

import phyx_amd
from phyx_amd import scenes, Configuration

K_STEPS = (1, 3, 5, 10)
DT = 1.0 / 60.0
ITERS = 20


def _reference(oracle, scene, solve_mode):
    w = oracle.OracleWorld(-200.0)
    w.add_scene(scene)
    out = {}
    for k in range(1, max(K_STEPS) + 1):
        w.update(DT, solve_mode, oracle.ISLAND_SINGLE, ITERS, ITERS)
        if k in K_STEPS:
            b = w.bodies()
            out[k] = (b["pos"].copy(), b["velocity"].copy())
    return out


def _diff(a, b):
    dp = np.hypot(a[0]["x"] - b[0]["x"], a[0]["y"] - b[0]["y"])
    dv = np.hypot(a[1]["x"] - b[1]["x"], a[1]["y"] - b[1]["y"])
    return float(dp.mean()), float(dp.max()), float(dv.mean())


def _check(dev, scalar, avx2):
    mean_dp, max_dp, mean_dv = _diff(dev[1], avx2[1])
    assert mean_dp <= 1e-3 and max_dp <= 0.1 and mean_dv <= 0.05, (mean_dp, max_dp, mean_dv)
    for k in K_STEPS:
        band = _diff(scalar[k], avx2[k])
        got = _diff(dev[k], avx2[k])
        assert np.isfinite(dev[k][0]["x"]).all() and np.isfinite(dev[k][1]["x"]).all()
        assert got[0] <= 2.0 * band[0] and got[2] <= 2.0 * band[2], (k, got, band)


def test_host_schedule_order_is_within_the_reference_cross_mode_band(oracle, built_lib):
    scene = scenes.stack(10, 100)
    scalar = _reference(oracle, scene, oracle.SOLVE_SCALAR)
    avx2 = _reference(oracle, scene, oracle.SOLVE_AVX2)
    w = oracle.OracleWorld(-200.0)
    w.add_scene(scene)
    dev = {}
    for k in range(1, max(K_STEPS) + 1):
        w.pre_solve(DT)
        b, cp, j = w.bodies(), w.contact_points(), w.joints()
        if len(j):
            static = ((b["inv_mass"] == 0) & (b["inv_inertia"] == 0)).astype(np.uint8)
            order, offs = phyx_amd.schedule_colours(j["body1"], j["body2"], static, j["contact_point_index"])
            oracle.solver_solve_ordered(b, cp, j, order, offs, ITERS, ITERS, oracle.STAG_COLOUR_SYNC)
        w.integrate_position(DT)
        if k in K_STEPS:
            b = w.bodies()
            dev[k] = (b["pos"].copy(), b["velocity"].copy())
    _check(dev, scalar, avx2)


@pytest.mark.gpu
@pytest.mark.parametrize("island_mode", [phyx_amd.ISLAND_SINGLE, phyx_amd.ISLAND_SINGLE_SLOPPY])
def test_device_world_is_within_the_reference_cross_mode_band(oracle, built_lib, island_mode):
    scene = scenes.stack(10, 100)
    scalar = _reference(oracle, scene, oracle.SOLVE_SCALAR)
    avx2 = _reference(oracle, scene, oracle.SOLVE_AVX2)
    w = phyx_amd.World(0, gravity=-200.0)
    w.add_scene(scene)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, island_mode, ITERS, ITERS)
    dev = {}
    for k in range(1, max(K_STEPS) + 1):
        w.Update(DT, cfg)
        if k in K_STEPS:
            b = w.bodies
            dev[k] = (b["pos"].copy(), b["velocity"].copy())
    _check(dev, scalar, avx2)


@pytest.mark.gpu
def test_full_size_200k_boxes_within_the_cross_mode_band(oracle, built_lib):
    """The same statement at BASELINE config 2's size (200 001 bodies, Single Sloppy, 20 iterations), K = 1..3 steps."""
    scene = scenes.stack(1000, 200)
    steps = (1, 2, 3)

    def reference(mode):
        w = oracle.OracleWorld(-200.0)
        w.add_scene(scene)
        out = {}
        for k in range(1, max(steps) + 1):
            w.update(DT, mode, oracle.ISLAND_SINGLE, ITERS, ITERS)
            b = w.bodies()
            out[k] = (b["pos"].copy(), b["velocity"].copy())
        return out
    scalar, avx2 = reference(oracle.SOLVE_SCALAR), reference(oracle.SOLVE_AVX2)
    w = phyx_amd.World(0, gravity=-200.0)
    w.add_scene(scene)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, ITERS, ITERS)
    for k in range(1, max(steps) + 1):
        w.Update(DT, cfg)
        b = w.bodies
        dev = (b["pos"], b["velocity"])
        band, got = _diff(scalar[k], avx2[k]), _diff(dev, avx2[k])
        assert np.isfinite(dev[0]["x"]).all() and np.isfinite(dev[1]["y"]).all()
        assert got[0] <= 2.0 * band[0] + 1e-6 and got[2] <= 2.0 * band[2] + 1e-6, (k, got, band)
        if k == 1:
            assert got[0] <= 1e-3 and got[1] <= 0.1 and got[2] <= 0.05, got
