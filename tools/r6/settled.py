"""The merged-island regime for the profiler: the 200k-box world stepped until its columns have merged into one island (step ~57), a few
steps more, nothing else.  usage: settled.py [steps=62]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 62
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 20, 20)
t = []
for s in range(steps):
    t0 = time.perf_counter(); w.Update(1 / 60, cfg); w.sync(); t.append(time.perf_counter() - t0)
st = w.solver.stats()
print("settled: last 5 steps %s ms, islands %d (max %d joints), counts %s" % (" ".join("%.3f" % (1e3 * x) for x in t[-5:]), st.island_count, st.island_max_size, w.counts()))
