import sys, os; R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, phyx_amd
from phyx_amd import scenes, Configuration
from oracle import binding as oracle
from phyx_amd import _lib as _phx_lib
oracle.set_arith(_phx_lib.load().phx_arith_mode())      # the oracle sweeps in the library's arithmetic form
from helpers import oracle_world
scene = scenes.stack(10, 100)
for iters in (15, 20):
  for mode in (0, 2):
    cfg = Configuration(phyx_amd.SOLVE_AVX2, mode, iters, iters)
    pw = phyx_amd.World(0, gravity=-200.0); pw.add_scene(scene); ow = oracle_world(scene)
    for step in range(3):
        pw.Update(1/60, cfg); ow.pre_solve(1/60)
        order, offs = pw.solver.schedule(); groups, _ = pw.solver.groups()
        b, cp, j = ow.bodies(), ow.contact_points(), ow.joints()
        st = oracle.solver_solve_grouped(b, cp, j, order, offs, groups, iters, iters, oracle.STAG_COLOUR_SYNC)
        ow.integrate_position(1/60)
        same = pw.bodies.tobytes() == ow.bodies().tobytes()
        ds = pw.solver.stats()
        print("iters", iters, "mode", mode, "step", step, "bodies equal", same, "groups", len(groups)-1, "colours", len(offs)-1,
              "device iterations", ds.impulse_iterations, ds.displacement_iterations, "oracle", st.impulse_iterations, st.displacement_iterations)
        if not same:
            d = pw.bodies; o = ow.bodies()
            bad = np.nonzero((d["velocity"]["x"] != o["velocity"]["x"]) | (d["velocity"]["y"] != o["velocity"]["y"]))[0]
            print("  differing bodies", len(bad), bad[:10])
            break
