import sys; sys.path.insert(0, "/root/repo")
import numpy as np, phyx_amd
from phyx_amd import Configuration
cfg = Configuration(0, 1, 5, 5)
w = phyx_amd.World(0); w.Update(1/60, cfg); w.sync(); print("empty world ok", w.counts())
w = phyx_amd.World(0); w.AddBody((0, 0), 0.0, (5, 5)); 
for _ in range(3): w.Update(1/60, cfg)
print("one body ok", w.counts(), w.bodies["pos"])
w = phyx_amd.World(0)
g = w.AddBody((0, 0), 0.0, (100, 10)); w.set_static(g) if hasattr(w, "set_static") else None
for i in range(150): w.AddBody((0.01 * i, 14.0), 0.0, (5, 5))       # 150 boxes in the same place: a dense clique
for s in range(4):
    w.Update(1/60, cfg)
st = w.solver.stats(); print("clique ok", w.counts(), "colours", st.colour_count, "finite", bool(np.isfinite(w.bodies["velocity"]["x"]).all()))
w = phyx_amd.World(0)
w.AddBody((0, 0), 0.0, (5, 5)); w.AddBody((float("nan"), 3), 0.0, (5, 5)); w.AddBody((1e30, 0), 0.0, (5, 5)); w.AddBody((2, 8), 0.3, (5, 5))
for s in range(3): w.Update(1/60, cfg)
print("nan/huge ok", w.counts())
for mode in (0, 1, 2, 3):
    w = phyx_amd.World(0); 
    for i in range(40): w.AddBody((12.0 * (i % 8), 6 + 11.0 * (i // 8)), 0.1 * i, (5, 5))
    for s in range(20): w.Update(1/60, Configuration(2, mode, 8, 8))
    print("mode", mode, "ok", w.counts(), bool(np.isfinite(w.bodies["pos"]["y"]).all()))
