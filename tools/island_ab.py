import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(int(sys.argv[1]) if len(sys.argv)>1 else 1000, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(3): w.Update(1/60, cfg)
w.PreSolve(1/60)
b, cp, j = w.bodies, w.contactPoints, w.contactJoints
s = phyx_amd.Solver(0)
db, dc, dj = (phyx_amd.DeviceArray(a) for a in (b, cp, j))
for iters in (0, 1, 2, 5, 20):
    c = Configuration(2, 2, iters, iters)
    r = s.bench(db, dc, dj, c, 3, 10)
    st = s.stats()
    print("iters", iters, "sweep_ms/step %.4f"%(r.impulse_kernel_ms/10), "total %.4f"%(r.total_ms/10), "groups", st.lds_islands, "colours", st.colour_count, "imp sweeps", st.impulse_iterations)
