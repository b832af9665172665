"""Long lockstep run: device-resident World vs oracle World, every byte compared every `every` steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
from oracle import binding as ob
from phyx_amd import _lib as _phx_lib
ob.set_arith(_phx_lib.load().phx_arith_mode())      # the oracle sweeps in the library's arithmetic form

def run(name, scene, steps, mode, every=10):
    cfg = Configuration(0, mode, 15, 15)
    pw = phyx_amd.World(0, gravity=-200.0); pw.add_scene(scene)
    ow = ob.OracleWorld(); ow.add_scene(scene)
    t0 = time.time(); worst = None
    for step in range(steps):
        pw.Update(1/60, cfg)
        ow.pre_solve(1/60)
        order, offs = pw.solver.schedule(); groups, _ = pw.solver.groups()
        ob.solver_solve_grouped(ow.bodies(), ow.contact_points(), ow.joints(), order, offs, groups, 15, 15, ob.STAG_COLOUR_SYNC)
        ow.integrate_position(1/60)
        if step % every == 0 or step == steps - 1:
            same = pw.bodies.tobytes() == ow.bodies().tobytes() and pw.contactJoints.tobytes() == ow.joints().tobytes() and pw.manifolds.tobytes() == ow.manifolds().tobytes()
            if not same:
                d = np.abs(pw.bodies["pos"]["y"] - ow.bodies()["pos"]["y"]).max()
                print(f"[{name}] DIVERGED at step {step}: max|dpos.y|={d}"); return False
    st = pw.solver.stats()
    print(f"[{name}] {steps} steps bit-exact; joints={pw.counts()[3]} groups(lds)={st.lds_islands} colours={st.colour_count} {time.time()-t0:.1f}s")
    return True

ok = True
ok &= run("falling3000/multiple", scenes.falling(3000, width=120.0, ymax=500.0), 240, 1)
ok &= run("tilted300/single", scenes.tilted(300), 300, 0)
ok &= run("stack12x80/sloppy", scenes.stack(12, 80), 150, 3)
if "--long" in sys.argv:      # tall columns that lean into each other: islands merge past the LDS caps, the HBM group is coloured on the device
    ok &= run("stack30x200/multiple", scenes.stack(30, 200), 60, 1, every=5)
    ok &= run("stack30x200/single", scenes.stack(30, 200), 40, 0, every=5)
print("SOAK", "OK" if ok else "FAILED")
