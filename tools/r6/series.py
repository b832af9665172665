import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/phyx_amd") else ".")
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
"""usage: series.py [steps=120]"""
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 20, 20)
t = []; info = []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for s in range(N):
    t0 = time.perf_counter(); w.Update(1 / 60, cfg); w.sync(); t.append(1e3 * (time.perf_counter() - t0))
    st = w.solver.stats()
    ki, parts, launches = w.solver.partition()
    info.append((st.lds_islands, st.colour_count, st.recoloured, w.counts()[3], ki, parts, launches, st.island_count, st.island_max_size))
for s in range(N):
    print("%2d %.3f ms  lds %4d colours %4d rec %d joints %d  ki %d parts %d sweep-launches %d islands %d max %d" % (s, t[s], *info[s]))
print("builds", w.build_counts(), "diag", getattr(w, "diagnostics", lambda: None)())
