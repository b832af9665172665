"""How often does the broadphase's sorted order change from one step to the next? (cfg-2 scene)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
prev = None; out = []
n = w.counts()[0]
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    w.Update(1/60, cfg); w.sync()
    srt, ent = w.collider.sorted(n)
    idx = np.asarray(srt["index"]).copy()
    if prev is not None:
        moved = int((idx != prev).sum())
        out.append(moved)
    prev = idx
print("bodies whose sorted position changed, per step:", out)
