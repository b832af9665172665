"""Island kernel time by sweep mix: (ci, pi) = both, impulses only, displacement only, none — what a role split could reach.
usage: ab_iters.py [columns=1000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(3): w.Update(1 / 60, cfg)
w.PreSolve(1 / 60)
b, cp, j = w.bodies, w.contactPoints, w.contactJoints
s = phyx_amd.Solver(0)
db, dc, dj = (phyx_amd.DeviceArray(a) for a in (b, cp, j))
for ci, pi in ((20, 20), (20, 0), (0, 20), (0, 0), (1, 1), (20, 20)):
    c = Configuration(2, 2, ci, pi)
    r = s.bench(db, dc, dj, c, 3, 10)
    st = s.stats()
    print("ci %2d pi %2d  island launch %.2f us  total %.2f us/step  imp sweeps %d disp sweeps %d" % (ci, pi, 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1), 1e3 * r.total_ms / 10, st.impulse_iterations, st.displacement_iterations))
