"""Runs a small world for a few dozen steps and prints a digest of every array after every 5th step: the same digest must come
out under every debugging knob (PHX_NO_MAILBOX, PHX_NO_SPECULATION, PHX_SCHEDULE_BUILDER=host ...).  usage: knob_check.py [mode]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phyx_amd
from phyx_amd import scenes, Configuration

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = hashlib.sha256()
for scene in (scenes.stack(12, 60), scenes.falling(700, width=90.0, ymax=300.0), scenes.clique(90)):
    w = phyx_amd.World(0, gravity=-200.0)
    w.add_scene(scene)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, mode, 12, 8)
    for step in range(30):
        w.Update(1 / 60, cfg)
        if step % 5 == 4:
            for a in (w.bodies, w.manifolds, w.contactJoints):
                h.update(a.tobytes())
print(h.hexdigest())
