"""The schedule rebuild whose components, joint counts and bins come from the MANIFOLDS (side stream: csrc/schedule_kernels.h
k_cc_link_manifolds, k_manifold_components, k_bin_components; the joints dealt by k_joint_scatter<true>) against the one that takes them from
the joints: a world is stepped and after every step the schedule the solver used (slot order, class offsets, groups) and every array are
hashed; the digest must be the same with PHX_NO_PRELABEL=1 (every rebuild from the joints), and without it some rebuilds must have come
from the manifolds.
usage: build_twin.py [scene=stack|falling|tilted|merge] [steps=40]   -> prints '<digest> <builds from the manifolds> <builds from the joints>'"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration

which = sys.argv[1] if len(sys.argv) > 1 else "stack"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene = {"stack": lambda: scenes.stack(24, 60), "falling": lambda: scenes.falling(700, width=90.0, ymax=300.0), "tilted": lambda: scenes.tilted(60),
         "merge": lambda: scenes.stack(16, 120)}[which]()
w = phyx_amd.World(0, gravity=-200.0)
w.add_scene(scene)
cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE_SLOPPY, 12, 8)
h = hashlib.sha256()
for step in range(steps):
    w.Update(1 / 60, cfg)
    order, offs = w.solver.schedule()
    groups = w.solver.groups()
    for a in (order, offs) + tuple(np.asarray(g) for g in (groups if isinstance(groups, tuple) else (groups,))):
        h.update(np.ascontiguousarray(a).tobytes())
    for a in (w.bodies, w.manifolds, w.contactJoints):
        h.update(a.tobytes())
lite, full = w.build_counts()
print(h.hexdigest(), lite, full)
