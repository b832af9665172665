import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for step in range(46):
    if step == 45: os.environ["X"]="1"
    w.Update(1/60, cfg)
w.sync()
