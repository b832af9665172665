"""Class sizes (units) of the LDS groups of the device-built schedule of a stack scene: rows of 64 lanes per sweep.
usage: class_sizes.py [columns] [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(nx, ny))
cfg = Configuration(2, 2, 20, 20)
for _ in range(3): w.Update(1 / 60, cfg)
w.PreSolve(1 / 60)
b, cp, j = w.bodies.copy(), w.contactPoints.copy(), w.contactJoints.copy()
s = phyx_amd.Solver(0)
st = s.SolveJoints(b, cp, j, cfg)
order, offs = s.schedule()
groups = s.groups()
print("stats", st.colour_count, st.lds_islands, "groups", len(groups[0]) - 1 if isinstance(groups, tuple) else groups)
goffs = groups[0] if isinstance(groups, tuple) else groups
sizes = np.diff(offs)
rows = []
ci = 0
for g in range(len(goffs) - 1):
    cls = []
    while ci < len(sizes) and offs[ci + 1] <= goffs[g + 1]:
        cls.append(int(sizes[ci])); ci += 1
    rows.append(cls)
for g, cls in enumerate(rows[:12]):
    print(g, cls)
