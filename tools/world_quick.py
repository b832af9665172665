"""cfg-2 world step time (median of 9 steps after 4) and broadphase device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(4): w.Update(1/60, cfg)
w.sync()
t = []
for _ in range(9):
    t0 = time.perf_counter(); w.Update(1/60, cfg); w.sync(); t.append(time.perf_counter() - t0)
print("world step median %.3f ms, broadphase device %.3f ms" % (1e3 * float(np.median(t)), w.collider.stats().device_ms))
