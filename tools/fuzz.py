"""Differential fuzz: random worlds (random sizes, angles, overlaps, several static boxes) stepped in lockstep on the device
and in the oracle; every byte of bodies / manifolds / joints compared after every step.   usage: fuzz.py [first_seed [count]] [--big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import Configuration
from oracle import binding as ob
from phyx_amd import _lib as _phx_lib
ob.set_arith(_phx_lib.load().phx_arith_mode())      # the oracle sweeps in the library's arithmetic form


BIG = "--big" in sys.argv        # thousands of bodies: many bins, 1024-lane groups, an HBM group once piles form


def scene(rng):
    n = int(rng.integers(3000, 20000)) if BIG else int(rng.integers(20, 700))
    width = float(rng.uniform(400, 3000)) if BIG else float(rng.uniform(40, 400))
    px, py, ang, sx, sy, st = [0.0], [0.0], [0.0], [width * 1.5], [10.0], [True]
    for _ in range(int(rng.integers(0, 3))):                  # walls / shelves
        px.append(float(rng.uniform(-width, width))); py.append(float(rng.uniform(20, 200))); ang.append(float(rng.uniform(-0.5, 0.5)))
        sx.append(float(rng.uniform(5, 60))); sy.append(float(rng.uniform(2, 10))); st.append(True)
    for _ in range(n):
        px.append(float(rng.uniform(-width, width))); py.append(float(rng.uniform(12, 400))); ang.append(float(rng.uniform(-3.2, 3.2)) if rng.random() < 0.7 else 0.0)
        sx.append(float(rng.uniform(1.5, 12))); sy.append(float(rng.uniform(1.5, 12))); st.append(False)
    f = lambda a: np.asarray(a, dtype=np.float32)
    return {"px": f(px), "py": f(py), "angle": f(ang), "sx": f(sx), "sy": f(sy), "static": np.asarray(st, dtype=bool)}


def run(seed):
    rng = np.random.default_rng(seed)
    sc = scene(rng)
    mode = int(rng.integers(0, 4)); iters = int(rng.integers(1, 25)); steps = int(rng.integers(10, 40 if BIG else 70))
    cfg = Configuration(int(rng.integers(0, 3)), mode, iters, int(rng.integers(0, 25)))
    pw = phyx_amd.World(0, gravity=-200.0); pw.add_scene(sc)
    ow = ob.OracleWorld(); ow.add_scene(sc)
    for step in range(steps):
        pw.Update(1 / 60, cfg)
        ow.pre_solve(1 / 60)
        order, offs = pw.solver.schedule(); groups, _ = pw.solver.groups()
        ob.solver_solve_grouped(ow.bodies(), ow.contact_points(), ow.joints(), order, offs, groups, cfg.contactIterationsCount, cfg.penetrationIterationsCount, ob.STAG_COLOUR_SYNC)
        ow.integrate_position(1 / 60)
        PARTS[0] += 1 if pw.solver.partition()[1] > 0 else 0; PARTS[1] += 1
        if not (pw.bodies.tobytes() == ow.bodies().tobytes() and pw.contactJoints.tobytes() == ow.joints().tobytes() and pw.manifolds.tobytes() == ow.manifolds().tobytes()):
            print("seed %d DIVERGED at step %d (bodies %d, mode %d, iters %d)" % (seed, step, len(sc["px"]), mode, iters)); return False
    return True


PARTS = [0, 0]          # steps solved with the partitioned-component path / all steps
args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(args[0]) if len(args) > 0 else 0
count = int(args[1]) if len(args) > 1 else 40
t0 = time.time(); bad = [s for s in range(first, first + count) if not run(s)]
print("fuzz: %d seeds, %d diverged %s, %.0f s; %d of %d steps took the partitioned-component path" % (count, len(bad), bad, time.time() - t0, PARTS[0], PARTS[1]))
sys.exit(1 if bad else 0)
