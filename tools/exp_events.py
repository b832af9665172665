"""experiment: cost of the HIP events that bracket the sweep launches inside bench(): run with PHX_BENCH_BRACKET_STRIDE=1 (every
step), unset (every 4th step, the default) and =0 (none)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
cfg = Configuration(2, 2, 20, 20)
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
for _ in range(3): w.Update(1/60, cfg)
w.PreSolve(1/60)
d = [phyx_amd.DeviceArray(a, 0) for a in (w.bodies, w.contactPoints, w.contactJoints)]
s = phyx_amd.Solver(0)
s.bench(d[0], d[1], d[2], cfg, 0, 1)
t = []
for _ in range(15):
    s.bench_stage(d[0], d[2], 20); s.synchronize()
    t0 = time.perf_counter(); r = s.bench(d[0], d[1], d[2], cfg, 0, 20); s.synchronize(); t.append((time.perf_counter() - t0) / 20)
print("ms per step: median %.4f min %.4f; events total %.4f" % (1e3 * np.median(t), 1e3 * min(t), r.total_ms / 20))
