"""PCIe-inclusive cost of the host-pointer drop-in calls at cfg 2 size (phx_solver_solve, phx_broadphase_update)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(3): w.Update(1/60, cfg)
w.PreSolve(1/60)
b, cp, j = w.bodies, w.contactPoints, w.contactJoints
s = phyx_amd.Solver(0)
for _ in range(3): s.SolveJoints(b.copy(), cp, j.copy(), cfg)
t = []
for _ in range(10):
    bb, jj = b.copy(), j.copy()
    t0 = time.perf_counter(); s.SolveJoints(bb, cp, jj, cfg); t.append(time.perf_counter() - t0)
mb_up = (b.nbytes + cp.nbytes + j.nbytes) / 1e6; mb_down = (b.nbytes + j.nbytes) / 1e6
print("phx_solver_solve: %.2f ms per call (median of 10); %.1f MB up, %.1f MB down; %.2f G joint-visits/s PCIe-inclusive" % (1e3 * np.median(t), mb_up, mb_down, s.stats().joint_visits / np.median(t) / 1e9))
c = phyx_amd.Collider(0)
c.UpdateBroadphaseAndPairs(b)
t = []
for _ in range(5):
    t0 = time.perf_counter(); c.UpdateBroadphaseAndPairs(b); t.append(time.perf_counter() - t0)
print("phx_broadphase_update: %.2f ms per call; %.1f MB up" % (1e3 * np.median(t), b.nbytes / 1e6))
