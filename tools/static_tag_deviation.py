"""The static-tag deviation at full size: the device gives every group a private, class-synchronous copy of a static
body's lastIteration tag and lets every group leave its sweeps on its own; the reference's Single modes keep ONE word per static body
(ref: Solver.cpp:474-478 reset once per call, :790-798 read, :900-910 written) and ONE early exit for the whole joint list (:189).
This replays the device's order through the oracle under the reference's rule (one island, one shared tag, sequential visibility,
global early exit) and reports how far the device's result is from it.

usage: static_tag_deviation.py [cfg2|cfg5|COLUMNSxROWS[:ITERS]] ... [--out file.json]      (test infrastructure: uses oracle/)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

CASES = {"cfg2": (1000, 200, 20), "cfg5": (1000, 500, 50)}


def deviation(columns, rows, iters, scene_steps=3, solver=None):
    import phyx_amd
    from phyx_amd import scenes, Configuration, _lib
    from oracle import binding as ob
    from helpers import presolve_state
    prev = ob.set_arith(_lib.load().phx_arith_mode())
    try:
        state = presolve_state(scenes.stack(columns, rows), scene_steps, iters=iters)
        cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, iters, iters)
        solver = solver or phyx_amd.Solver(0)
        gb, cp, gj = (a.copy() for a in state)
        st = solver.SolveJoints(gb, cp, gj, cfg)
        order, colours = solver.schedule()
        groups, lds_groups = solver.groups()
        # the device's own rule (what every parity test replays): bit-exact
        ob_, _, oj = (a.copy() for a in state)
        ob.solver_solve_grouped(ob_, cp, oj, order, colours, groups, iters, iters, ob.STAG_COLOUR_SYNC)
        exact = gb.tobytes() == ob_.tobytes() and gj.tobytes() == oj.tobytes()
        # the reference's Single-mode rule on the same order
        rb, _, rj = (a.copy() for a in state)
        rst = ob.solver_solve_ordered(rb, cp, rj, order, colours, iters, iters, ob.STAG_SEQUENTIAL)

        def amax(a, b):
            return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if len(a) else 0.0
        dvel = max(amax(gb["velocity"]["x"], rb["velocity"]["x"]), amax(gb["velocity"]["y"], rb["velocity"]["y"]))
        dang = amax(gb["angular_velocity"], rb["angular_velocity"])
        ddis = max(amax(gb["displacing_velocity"]["x"], rb["displacing_velocity"]["x"]), amax(gb["displacing_velocity"]["y"], rb["displacing_velocity"]["y"]))
        dimp = max(amax(gj["normal_acc"], rj["normal_acc"]), amax(gj["friction_acc"], rj["friction_acc"]))
        differing = int(np.count_nonzero((gb["velocity"]["x"] != rb["velocity"]["x"]) | (gb["velocity"]["y"] != rb["velocity"]["y"]) |
                                         (gb["angular_velocity"] != rb["angular_velocity"])))
        # position after IntegratePosition moves by dt * dvel + ddis
        dt = 1.0 / 60.0
        return {"scene": "stack(%d,%d)" % (columns, rows), "bodies": int(len(gb)), "joints": int(len(gj)), "iterations": iters, "groups": int(len(groups) - 1),
                "device_equals_oracle_in_device_rule": bool(exact),
                "device_impulse_sweeps_max": int(st.impulse_iterations), "reference_rule_impulse_sweeps": int(rst.impulse_iterations),
                "reference_rule_joints_computed": int(rst.joints_computed), "stag_events": int(rst.stag_events),
                "bodies_differing": differing,
                "max_abs_dvel": dvel, "max_abs_dangvel": dang, "max_abs_ddisplacing": ddis, "max_abs_dimpulse": dimp,
                "max_abs_dpos_after_integrate": dt * dvel + ddis,
                "T1_vel": 1e-3, "T1_pos": 1e-4, "inside_T1": bool(dvel <= 1e-3 and dt * dvel + ddis <= 1e-4)}
    finally:
        ob.set_arith(prev)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = None
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
        args = [a for a in args if a != out]
    res = {}
    for a in args or ["cfg2"]:
        if a in CASES:
            c, r, it = CASES[a]
        else:
            dims, _, its = a.partition(":")
            c, r = (int(x) for x in dims.split("x"))
            it = int(its or 20)
        res[a] = deviation(c, r, it)
        print(a, json.dumps(res[a]), flush=True)
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
