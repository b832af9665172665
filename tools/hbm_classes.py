"""Class sizes of the HBM group in the merged-island regime of the stack scene (step 47)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 48):
    w.Update(1/60, cfg)
w.sync()
order, offs = w.solver.schedule(); groups, lds = w.solver.groups()
offs = np.asarray(offs); groups = np.asarray(groups)
hb = groups[lds] if lds < len(groups) - 1 else None
print("groups", len(groups) - 1, "lds", lds, "joints", len(order))
if hb is not None:
    sizes = np.diff(offs[np.searchsorted(offs, hb):])
    print("HBM group: %d joints in %d classes:" % (groups[-1] - hb, len(sizes)), sizes.tolist())
