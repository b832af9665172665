"""How often does PackManifolds find dead manifolds? (dead = manifolds before + new pairs - manifolds after), cfg-2 scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
prev = 0; out = []
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 48):
    w.Update(1/60, cfg); w.sync()
    nm = w.counts()[1]; new = w.collider.stats().new_pairs
    out.append(prev + new - nm); prev = nm
print("dead manifolds per step:", out)
